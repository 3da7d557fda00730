#!/bin/bash
# round-6 GPU call: <tag> <what...>; what = attn | attntime | model | packed | suite | bench | ab "<env A>" "<env B>" ...
TAG=${1:-r6}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
while [ $# -gt 0 ]; do
  case "$1" in
    attn) timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round4.py -q --maxfail=12 -p no:cacheprovider -k "split or skips or attention_backward" > gpurun_out/${TAG}_attn.log 2>&1; echo "attn rc=$?"; tail -60 gpurun_out/${TAG}_attn.log; shift;;
    attntime) timeout 600 python tools/probes/attn_bwd_forms_time.py > gpurun_out/${TAG}_attn_bwd_forms_time.txt 2>&1; echo "attntime rc=$?"; tail -12 gpurun_out/${TAG}_attn_bwd_forms_time.txt; shift;;
    attnprof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python $R/tools/probes/attn_bwd_forms_time.py --reps 5 > $R/gpurun_out/${TAG}_attnprof_run.log 2>&1); echo "attnprof rc=$?"
      t=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1)
      python tools/prof_summary.py "$t" gpurun_out/${TAG}_attn_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/probes/attn_bwd_forms_time.py --reps 5"; grep -i "attn\|calls" gpurun_out/${TAG}_attn_kernel_stats.csv | head -20; rm -rf gpurun_out/prof_$TAG; shift;;
    packed) timeout 900 python -m pytest tests/test_gpu_packed.py -q --maxfail=25 -x -p no:cacheprovider > gpurun_out/${TAG}_packed.log 2>&1; echo "packed rc=$?"; tail -40 gpurun_out/${TAG}_packed.log; shift;;
    model) timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_round4.py tests/test_gpu_packed.py tests/test_gpu_raw_memory.py -q --maxfail=10 -p no:cacheprovider > gpurun_out/${TAG}_model.log 2>&1; echo "model rc=$?"; tail -30 gpurun_out/${TAG}_model.log; shift;;
    suite) timeout 1800 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/${TAG}_gputest.log 2>&1; echo "suite rc=$?"; tail -25 gpurun_out/${TAG}_gputest.log; shift;;
    bench) timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; python tools/bench_summary.py gpurun_out/${TAG}_bench.json 2>/dev/null || tail -c 1500 gpurun_out/${TAG}_bench.json; shift;;
    benchq) timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; python tools/bench_summary.py gpurun_out/${TAG}_bench.json 2>/dev/null || tail -c 1500 gpurun_out/${TAG}_bench.json; shift;;
    ab) shift; bash tools/gpu_ab.sh "$@" 2>&1 | tee gpurun_out/${TAG}_ab.txt; break;;
    *) echo "unknown $1"; shift;;
  esac
done
