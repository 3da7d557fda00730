timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_packed.py -q -x -p no:cacheprovider -k "attention or attn" 2>&1 | tail -3
for i in 1 2; do timeout 300 python tools/probes/attn_bwd_forms_time.py --forms recompute --order snake 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_ac_fwd32_pv_order.txt
