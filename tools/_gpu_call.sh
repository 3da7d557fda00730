bash tools/gpu_r6.sh r06_ab suite benchq
