timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_packed.py -q -x -p no:cacheprovider 2>&1 | tail -5
bash tools/gpu_ab_attr.sh "o.BALANCED_ORDER = False" "pass" 2>&1 | tee gpurun_out/r06_t_ab_balanced_order.txt
bash tools/gpu_r6.sh r06_t model
