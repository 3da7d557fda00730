#!/bin/bash
# A/B on ONE box of library builds: bench.py (train_cap, hipgraph, kernel timer on) under each bmt_amd/lib/<name>, twice, interleaved;
# prints ms/step and the encoder attention classes.   usage: tools/gpu_ab_lib.sh libbmt_hip.so libbmt_hip_<variant>.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    out=$(BMT_LIB_PATH=$R/bmt_amd/lib/$v timeout 300 python bench.py --no-cpu-baseline --no-clock-probe --steps 30 --warmup 8 2>/dev/null | tail -1)
    python - "$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); kc = d.get("kernel_classes", {})
print(sys.argv[1].ljust(24), f"{d['ms_per_step']:.3f} ms/step ", "  ".join(f"{k[5:]}={v['ms_per_step']:.3f}" for k, v in kc.items() if k.startswith("attn_") and "_enc_" in k))
PY
  done
done
