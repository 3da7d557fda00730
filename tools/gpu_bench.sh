#!/bin/bash
# One gpurun call: smoke, bench (N=1), rocprofv3 kernel-trace of a short bench run.  Logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "=== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "=== bench"; timeout ${BENCH_TIMEOUT:-400} python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo rc=$?; tail -c 6000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err
if [ -z "$NO_PROF" ]; then
echo "=== rocprofv3 --kernel-trace --stats"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-clock-probe > $R/gpurun_out/prof_run.log 2>&1; echo rc=$?
cd $R; find gpurun_out/prof -type f | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" | cut -c1-200; grep -o '{"metric.*' gpurun_out/prof_run.log | cut -c1-400
fi
