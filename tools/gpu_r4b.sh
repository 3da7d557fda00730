#!/bin/bash
# round 4, second GPU visit: new tests, whole suite, zero-row probe, gradient report, bench lines (train_cap driver form, train_prop graph), kernel stats
TAG=${1:-r04_b}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 500 python -m pytest tests/test_gpu_round4.py -q > gpurun_out/${TAG}_new.log 2>&1; echo "new tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_new.log | tail -20
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_round4.py > gpurun_out/${TAG}_gputest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_gputest.log | tail -30
timeout 300 python tools/probes/zero_rows.py > gpurun_out/${TAG}_zero_rows.txt 2>&1; echo "zero rows rc=$?"; grep -v amdgpu.ids gpurun_out/${TAG}_zero_rows.txt | tail -22
BMT_GRAD_REPORT=1 timeout 400 python -m pytest tests/test_gpu_model.py -q -s -k "captioning or config or deep or full or mid" > gpurun_out/${TAG}_grad_report.txt 2>&1; echo "grad report rc=$?"
grep -E "global relative gradient error|^ +[0-9.]+%" gpurun_out/${TAG}_grad_report.txt | sort -k1,1 | awk '/global/ {print} !/global/ {n++; if (n<=0) print}' | head -40
grep -E "^ +[0-9.]+%" gpurun_out/${TAG}_grad_report.txt | sort -rn | head -12
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
grep "kernel timer\|timed region" gpurun_out/${TAG}_bench.err
python tools/bench_summary.py gpurun_out/${TAG}_bench.json 2>/dev/null | head -40
timeout 300 python bench.py --procedure train_prop --steps 10 --warmup 3 --no-cpu-baseline --no-clock-probe > gpurun_out/${TAG}_prop.json 2> gpurun_out/${TAG}_prop.err; echo "train_prop rc=$?"; grep "bench\]" gpurun_out/${TAG}_prop.err | tail -4
python tools/bench_summary.py gpurun_out/${TAG}_prop.json 2>/dev/null | head -12
bash tools/gpu_prof.sh $TAG 6 2>&1 | head -60
