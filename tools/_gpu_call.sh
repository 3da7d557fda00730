timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_packed.py -q -x -p no:cacheprovider -k "graph or captured or replay" 2>&1 | tail -5
bash tools/gpu_ab_attr.sh "o.SPLIT_TAIL = False" "pass" 2>&1 | tee gpurun_out/r06_u_ab_split_tail.txt
bash tools/gpu_timeline.sh r06_u
