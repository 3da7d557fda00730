#!/bin/bash
# the ten-Adam-step trajectory test's loss gaps to the oracle, four runs per arm of bmt_amd.ops module attributes (the trajectory is chaotic in the
# gradients' low bits: tests/test_gpu_model.py::test_ten_adam_steps_against_the_oracle).  usage: tools/probes/adam_gap_arms.sh "o.ATTR = value" ...
for arm in "$@"; do
  for i in 1 2 3 4; do
    timeout 300 python -c "
import sys
import bmt_amd.ops as o
$arm
import pytest
sys.exit(pytest.main(['tests/test_gpu_model.py', '-m', 'gpu', '-q', '-x', '-s', '-p', 'no:cacheprovider', '-k', 'ten_adam_steps']))
" 2>&1 | grep "loss, " | python -c "
import sys,re
ls=[l for l in sys.stdin]
a=[float(x) for x in re.findall(r\"'([0-9.]+)'\", ls[0])]; b=[float(x) for x in re.findall(r\"'([0-9.]+)'\", ls[1])]
g=[abs(x-y) for x,y in zip(a,b)]
print('$arm', 'max1-5 %.2e  max %.2e' % (max(g[:5]), max(g)), ' '.join('%.1e'%x for x in g))
"
  done
done
