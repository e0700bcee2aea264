// tPatchGNN LearnableTE + TTCN patch encoder (fused forward / backward).
#pragma once
#include "common.hpp"

struct TtcnParams {
    const float *te_ws, *te_bs;   // te_scale      Linear(1,1)
    const float *te_wp, *te_bp;   // te_periodic   Linear(1, te_dim-1)
    const float *W1, *b1;         // Filter_Generators.0  (K, F)
    const float *W2, *b2;         // Filter_Generators.2  (K, K)
    const float *W3, *b3;         // Filter_Generators.4  (F*K, K)
    const float* T_bias;          // (K)
};

// length of one gradient row: W3 | b3 | W2 | b2 | W1 | b1 | T_bias | te_ws, te_bs | te_wp | te_bp
int ttcn_grad_len(int F, int K);
// x, tt, mask: (P, L); out: (P, K); stat: (P, 3, F*K) saved for backward
int launch_ttcn_fwd(int P, int L, int F, int K, const float* x, const float* tt, const float* mask, const TtcnParams& w,
                    float* out, float* stat, hipStream_t s);
// partial: (nblocks, ttcn_grad_len) -- the caller column-sums it
int launch_ttcn_bwd(int P, int L, int F, int K, const float* x, const float* tt, const float* mask, const TtcnParams& w,
                    const float* out, const float* stat, const float* dout, float* partial, int nblocks, hipStream_t s);
