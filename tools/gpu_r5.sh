#!/bin/bash
# round-5 GPU call: <tag> <what...>; what = packed | model | suite | kernels | ab "<env A>" "<env B>" ... | ablib <variant name>
# (ablib: same-box A/B of the default library against bmt_amd/lib/libbmt_hip_<variant>.so, built beforehand with
#  BMT_VARIANT=<variant> BMT_VARIANT_FLAGS="-D..." bash bmt_amd/csrc/build.sh)
TAG=${1:-r5}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
while [ $# -gt 0 ]; do
  case "$1" in
    packed) timeout 900 python -m pytest tests/test_gpu_packed.py -q --maxfail=25 -x -p no:cacheprovider > gpurun_out/${TAG}_packed.log 2>&1; echo "packed rc=$?"; tail -60 gpurun_out/${TAG}_packed.log; shift;;
    packed_all) timeout 900 python -m pytest tests/test_gpu_packed.py -q --maxfail=40 -p no:cacheprovider > gpurun_out/${TAG}_packed.log 2>&1; echo "packed rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_packed.log | tail -50; shift;;
    model) timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_round4.py -q --maxfail=10 -p no:cacheprovider > gpurun_out/${TAG}_model.log 2>&1; echo "model rc=$?"; tail -30 gpurun_out/${TAG}_model.log; shift;;
    suite) timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider > gpurun_out/${TAG}_gputest.log 2>&1; echo "suite rc=$?"; tail -25 gpurun_out/${TAG}_gputest.log; shift;;
    kernels) timeout 1200 python -m pytest tests/test_gpu_kernels.py -q --maxfail=15 -p no:cacheprovider -k "gemm or dx or planes" > gpurun_out/${TAG}_kernels.log 2>&1; echo "kernels rc=$?"; tail -30 gpurun_out/${TAG}_kernels.log; shift;;
    ablib) bash tools/gpu_ab.sh "BMT_LIB_PATH=$R/bmt_amd/lib/libbmt_hip.so" "BMT_LIB_PATH=$R/bmt_amd/lib/libbmt_hip_$2.so" 2>&1 | tee gpurun_out/${TAG}_ablib.txt; shift; shift;;
    ab) shift; bash tools/gpu_ab.sh "$@" 2>&1 | tee gpurun_out/${TAG}_ab.txt; break;;
    *) echo "unknown $1"; shift;;
  esac
done
