#!/bin/bash
# A/B on ONE box of bmt_amd.ops module attributes: bench.py (train_cap, hipgraph, no CPU baseline) under each setting, twice, interleaved.
# usage: tools/gpu_ab_attr.sh "o.ATTR = value; o.OTHER = value" "o.ATTR = other value" ...     (each argument: Python statements, o = bmt_amd.ops; sys.argv is bench.py's command line)
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    out=$(timeout 300 python -c "
import sys
import bmt_amd.ops as o
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-kernel-timer', '--no-clock-probe', '--steps', '40', '--warmup', '8']
$v
import bench
bench.main()
" 2>/dev/null | tail -1)
    python - "$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f"{sys.argv[1]:70s} {d['ms_per_step']:7.3f} ms/step  {d['value']:.0f} tokens/s")
PY
  done
done
