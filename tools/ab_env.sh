#!/bin/bash
# A/B of environment toggles on the headline bench in ONE gpurun call (same box): tools/ab_env.sh "VAR=1" "VAR2=1 VAR3=0" ...
# the baseline (no toggle) is run first and last
run() { echo "[$1] $(env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timer 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
run "BMT_AB=base"
for t in "$@"; do run "$t"; done
run "BMT_AB=base"
