#!/bin/bash
# A/B on ONE box: bench.py (train_cap, hipgraph, no CPU baseline) under each of the given environment settings, twice, interleaved.
# usage: tools/gpu_ab.sh "<env A>" "<env B>" ...      e.g. tools/gpu_ab.sh "BMT_PACK_ROWS=1" "BMT_PACK_ROWS=0"
mkdir -p gpurun_out
for round in 1 2; do
  for e in "$@"; do
    out=$(env $e timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | tail -1)
    python - "$e" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
kc = d["kernel_classes"]
print(f"{sys.argv[1]:40s} {d['ms_per_step']:7.3f} ms/step  " + "  ".join(f"{k.split('_planes_')[-1][:14]}={v['ms_per_step']:.3f}" for k, v in kc.items() if k.startswith('gemm')))
PY
  done
done
