// experiment driver (tools/experiments/build.sh -> tools/experiments/libbmt_exp.so; never loaded by the product): the product's attention
// translation unit + the forward-kernel variants and probe copies of tools/probes/attn_fwd32_check.py
#include "../../bmt_amd/csrc/attention_bf16.hip"
#include "attn_fwd32.hip"
