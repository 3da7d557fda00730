timeout 600 python -m pytest tests/test_gpu_round6.py -q -x -p no:cacheprovider -k "shared_by_the_heads" 2>&1 | tail -15
