bash tools/gpu_final.sh r06_zz
