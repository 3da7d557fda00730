#!/bin/bash
# L2 behaviour of the GEMM on one microbenchmark shape.  usage: MB_FILTER="v proj" tools/gpu_pmc_l2.sh
mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/l2pmc$i -o p -- python $R/tools/microbench.py gemm --iters 2 > $R/gpurun_out/l2pmc$i.log 2>&1
  echo "pmc pass $i rc=$? ($set)"
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/l2pmc*')):
    for f in glob.glob(d+'/*counter_collection.csv'):
        agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name']
            if 'gemm_bf16_kernel' not in n: continue
            key=('x3' if '<3>' in n else 'x1', r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X'))
            a=agg[key][r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
        for k,v in agg.items():
            print(d, k, {c:(round(x[1]/x[0])) for c,x in v.items()})
PY
