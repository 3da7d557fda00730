// EXPERIMENT driver (not part of libbmt_hip.so; built by exp/build.sh into bmt_amd/lib/libbmt_exp.so, driven by
// tools/probes/attn_fwd32_check.py): the instruction-placement variants of attn_fwd32_kernel (attention_bf16.hip) and a switch back to
// the 16-query kernel, behind the product's argument block -- old and new kernel, and every variant, timed in ONE process on one box.
//   variant v < 8: DMAV = v % 4 (where the next tile's DMA requests are issued), PRIO = v < 4 (s_setprio around the MFMA phases);
//   variant 100:   attn_fwd64_kernel (the 16-query kernel), whatever the shape.
// Result (profiles/r02_q_attn_fwd32_variants.txt): the eight variants are within +-3 % of each other on every shape.
#include "../attention_bf16.hip"

namespace {
int g_variant = 0;
}

extern "C" void bmt_exp_set_variant(int v) { g_variant = v; }

// same argument block as bmt_attn_fwd_bf16 (include/bmt_hip.h); single-pass precisions, d_k 128 / 256, key-padding masks only
extern "C" int bmt_exp_attn_fwd32(const bmt_attn_fwd_bf16_args* a, void* stream) {
    BMT_CHECK_ARG(a && a->Qh && a->Kh && a->Vh && (a->O || a->Oh) && a->lse, "bmt_exp_attn_fwd32: null pointer");
    BMT_CHECK_ARG(a->dk == 128 || a->dk == 256, "bmt_exp_attn_fwd32: d_k=%d not in {128,256}", a->dk);
    BMT_CHECK_ARG(a->precision == BMT_PREC_BF16 || a->precision == BMT_PREC_F16, "bmt_exp_attn_fwd32: single-pass precisions only");
    BMT_CHECK_ARG(a->mask == nullptr || a->mask_qs == 0, "bmt_exp_attn_fwd32: key-padding masks only");
    BMT_CHECK_ARG(a->Sk <= 8192 && (int64_t)a->Sk * a->ldk * 2 < (1ll << 31) && (int64_t)a->Sk * a->ldv * 2 < (1ll << 31), "bmt_exp_attn_fwd32: Sk too large");
    AttnPB p;
    memset(&p, 0, sizeof(p));
    p.Qh = a->Qh; p.Kh = a->Kh; p.Vh = a->Vh;
    p.Ow = a->O; p.lsew = a->lse;
    p.Owh = a->Oh; p.Owl = a->Of ? a->Of : a->Ol; p.ow_f16 = a->Of != nullptr; p.ldop = a->ldop; p.bsop = a->bsop;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.bsq = a->bsq; p.bsk = a->bsk; p.bsv = a->bsv; p.bso = a->bso;
    p.mask = a->mask; p.mask_bs = a->mask_bs; p.mask_qs = a->mask_qs;
    p.B = a->B; p.H = a->H; p.Sq = a->Sq; p.Sk = a->Sk;
    p.scale = a->scale; p.drop_p = a->drop_p; p.rng = a->rng; p.site = a->site;
    hipStream_t st = (hipStream_t)stream;
    const bool f16 = a->precision == BMT_PREC_F16;
    if (g_variant == 100) {
        if (a->dk == 256) return f16 ? launch_fwd<256, 1, true>(p, st, 0) : launch_fwd<256, 1, false>(p, st, 0);
        return f16 ? launch_fwd<128, 1, true>(p, st, 0) : launch_fwd<128, 1, false>(p, st, 0);
    }
    if (a->dk == 256 && f16) {
        switch (g_variant) {
            case 1: return launch_fwd32<256, true, 1, true>(p, st);
            case 2: return launch_fwd32<256, true, 2, true>(p, st);
            case 3: return launch_fwd32<256, true, 3, true>(p, st);
            case 4: return launch_fwd32<256, true, 0, false>(p, st);
            case 5: return launch_fwd32<256, true, 1, false>(p, st);
            case 6: return launch_fwd32<256, true, 2, false>(p, st);
            case 7: return launch_fwd32<256, true, 3, false>(p, st);
            default: return launch_fwd32<256, true, 0, true>(p, st);
        }
    }
    if (a->dk == 256) return launch_fwd32<256, false>(p, st);
    return f16 ? launch_fwd32<128, true>(p, st) : launch_fwd32<128, false>(p, st);
}
