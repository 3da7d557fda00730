#!/bin/bash
# A/B on ONE box of a bmt_amd.ops module attribute: bench.py (train_cap, hipgraph, no CPU baseline) with ops.<ATTR> = each value, twice, interleaved.
# usage: tools/gpu_ab_attr.sh ATTR VALUE_A VALUE_B ...        (values are Python literals)
mkdir -p gpurun_out
ATTR=$1; shift
for round in 1 2; do
  for v in "$@"; do
    out=$(timeout 300 python -c "
import sys
import bmt_amd.ops as o
o.$ATTR = $v
sys.argv = ['bench.py', '--no-cpu-baseline', '--steps', '30', '--warmup', '8']
import bench
bench.main()
" 2>/dev/null | tail -1)
    python - "$ATTR=$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"{sys.argv[1]:40s} {d['ms_per_step']:7.3f} ms/step  {d['value']:.0f} tokens/s")
PY
  done
done
