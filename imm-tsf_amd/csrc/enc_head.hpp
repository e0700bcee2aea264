// The attention half of tPatchGNN's transformer layer at d_model 32 as one kernel per direction (enc_head.hip)
#pragma once
#include "common.hpp"

bool enc_head32_ok(int Bs, int S, int D, int H);             // D == 32, S <= 8, H in {1, 2, 4}
size_t enc_head32_slab_floats(int Bs, int S, int H);         // `slabs` of the backward
// x (Bs*S, 32) -> x1 = LayerNorm1(x + dropout1(out_proj(attention(in_proj(x))))), xhat, rstd
int launch_enc_head32_fwd(const float* x, int Bs, int S, int H, const float* in_w, const float* in_b, const float* out_w, const float* out_b,
                          const float* ln_w, const float* ln_b, float eps, DropCfg da, DropCfg dd, uint64_t site, float* x1, float* xhat,
                          float* rstd, hipStream_t s);
// d1 (gradient wrt x1) -> dx and the six parameter gradients (written)
int launch_enc_head32_bwd(const float* x, const float* d1, const float* xhat, const float* rstd, int Bs, int S, int H, const float* in_w,
                          const float* in_b, const float* out_w, const float* out_b, const float* ln_w, float eps, DropCfg da, DropCfg dd,
                          uint64_t site, float* dx, float* g_in_w, float* g_in_b, float* g_out_w, float* g_out_b, float* g_ln_w, float* g_ln_b,
                          float* slabs, hipStream_t s);
