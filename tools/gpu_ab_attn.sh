#!/bin/bash
# same-box A/B of environment settings on the attention-backward kernels alone: tools/probes/zero_rows.py's kernel-level timing of the audio
# self-attention (dense dO / zero suffix) + the bench's class times.   usage: tools/gpu_ab_attn.sh "<env A>" "<env B>" ...
for round in 1 2; do
  for e in "$@"; do
    out=$(env $e timeout 300 python bench.py --no-cpu-baseline --no-clock-probe --steps 30 --warmup 8 2>/dev/null | tail -1)
    python - "$e" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
kc = d["kernel_classes"]
print(f"{sys.argv[1]:32s} {d['ms_per_step']:7.3f} ms/step  " + "  ".join(f"{k[:22]}={v['ms_per_step']:.3f}" for k, v in kc.items() if k.startswith('attn')))
PY
  done
done
