#!/bin/bash
# the parity numbers the GPU tests print (log-prob distance, gradient errors against the oracle), per arm of bmt_amd.ops module attributes
# usage: tools/probes/parity_arms.sh "o.ATTR = value" ...
for arm in "$@"; do
  echo "=== $arm"
  timeout 600 python -c "
import sys
import bmt_amd.ops as o
$arm
import pytest
sys.exit(pytest.main(['tests/test_gpu_model.py', 'tests/test_gpu_raw_memory.py', '-m', 'gpu', '-q', '-s', '-p', 'no:cacheprovider', '-k', 'mid or cfg0 or full_cap or raw_memory or ten_adam']))
" 2>&1 | grep "global relative\|dlogp\|gradient norm\|reassociated cross\|passed\|failed\|loss, "
done
