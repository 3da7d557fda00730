R=$PWD
bash tools/gpu_ab.sh "BMT_LIB_PATH=$R/bmt_amd/lib/libbmt_hip.so" "BMT_LIB_PATH=$R/bmt_amd/lib/libbmt_hip_sk2.so" "BMT_LIB_PATH=$R/bmt_amd/lib/libbmt_hip_sk4.so" 2>&1 | sed "s#$R/bmt_amd/lib/##" | tee gpurun_out/r06_z_ab_splitk_audio_dx.txt
