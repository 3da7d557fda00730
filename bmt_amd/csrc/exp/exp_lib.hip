// the experiment library as one translation unit: attention_bf16.hip (the product file, for its helpers and kernels) once, then the drivers
#define BMT_EXP_LIB 1
#include "attn_fwd32.hip"
#include "attn_bwd32.hip"
#include "attn_bwd_split.hip"
