#!/bin/bash
# per-kernel time of the eagerly issued train_cap step: rocprofv3 --kernel-trace --stats -> gpurun_out/<tag>_kernel_stats.csv (+ raw stats csv)
TAG=${1:-x}; STEPS=${2:-6}; PROC=${3:-train_cap}
mkdir -p gpurun_out; export TMPDIR=/tmp; export BMT_ENC_STREAMS=1   # kernels one at a time: isolated durations / counters (the bench's timed region forks two streams)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python $R/bench.py --procedure $PROC --steps $STEPS --warmup 3 --no-graph --no-cpu-baseline --no-kernel-timer --no-clock-probe > $R/gpurun_out/${TAG}_prof_run.log 2>&1; echo "rocprof rc=$?"
cd $R
t=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
s=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && cp "$s" gpurun_out/${TAG}_rocprofv3_kernel_stats_raw.csv
python tools/prof_summary.py "$t" gpurun_out/${TAG}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --procedure $PROC --steps $STEPS --warmup 3 --no-graph --no-cpu-baseline --no-kernel-timer --no-clock-probe  ($((STEPS + 3)) eager steps incl. warm-up)" $((STEPS + 3)) gpurun_out/${TAG}_last_step_trace.csv
head -45 gpurun_out/${TAG}_kernel_stats.csv
tail -2 gpurun_out/${TAG}_prof_run.log
rm -rf gpurun_out/prof_$TAG
