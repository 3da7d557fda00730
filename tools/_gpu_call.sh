OFF="o.EARLY_REFRESH = False; o.CONST_TABLES = False; o.COLSUM_BESIDE_DW = False; o.DEFER_ZERO = False; sys.argv.append('--no-prefetch')"
bash tools/gpu_ab_attr.sh "$OFF" "pass" "o.EARLY_REFRESH = False" "sys.argv.append('--no-prefetch')" "o.EARLY_REFRESH = False; o.COLSUM_BESIDE_DW = False" 2>&1 | tee gpurun_out/r06_r_ab_step_edges.txt
