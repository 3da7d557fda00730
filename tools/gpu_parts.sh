#!/bin/bash
# the batch in parts (CaptioningTrainStep(microbatches=M)): the staged probe, the parity tests, then the captured step under each setting on ONE box
# (twice, interleaved) and the replay timeline at M = 2.     usage: tools/gpu_parts.sh <tag>
TAG=${1:-parts}
mkdir -p gpurun_out
timeout 120 python tools/probes/parts_debug.py 2 4 0.0 tiny > gpurun_out/${TAG}_debug.log 2>&1; echo "tiny probe rc=$?"; grep -v "amdgpu.ids\|Extension\|^$" gpurun_out/${TAG}_debug.log | tail -8 | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parts.py -q -m gpu -s > gpurun_out/${TAG}_parts_tests.log 2>&1; echo "tests rc=$?"
grep -v "^$" gpurun_out/${TAG}_parts_tests.log | tail -12 | cut -c1-250
run() { env $1 timeout 400 python -X faulthandler bench.py --no-cpu-baseline --no-kernel-timer --no-clock-probe --steps 20 --warmup 5 $2 2>gpurun_out/${TAG}_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1 $2'.ljust(60), f\"{d['ms_per_step']:.3f} ms/step  {d['value']:.0f} tok/s  loss {d['config']['final_loss']:.4f} parts {d['config'].get('parts_in_flight')}\")
except Exception as e: print('$1 $2 FAILED', e)
"; }
for round in 1 2; do
  run "BMT_MICROBATCHES=1" ""
  run "BMT_MICROBATCHES=2" ""
  run "BMT_MICROBATCHES=2 BMT_PARTS_STAGGER=0" ""
  run "BMT_MICROBATCHES=2 BMT_PARTS_SIDE=0" ""
  run "BMT_MICROBATCHES=3" ""
  run "BMT_MICROBATCHES=4" ""
done 2>&1 | tee gpurun_out/${TAG}_ab_parts.txt
tail -5 gpurun_out/${TAG}_err.log | cut -c1-200
BMT_MICROBATCHES=2 bash tools/gpu_timeline.sh ${TAG}_m2
head -8 gpurun_out/${TAG}_m2_timeline.txt | cut -c1-150
