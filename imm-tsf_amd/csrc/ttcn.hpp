// tPatchGNN LearnableTE + TTCN patch encoder: see ttcn.hip (entry points are declared in include/immtsf.h).
#pragma once
#include "common.hpp"

// ttcn_fused.hip: third filter layer + masked softmax + pooling in one kernel per direction (bf16 mode, ttcn_dim <= 32,
// L <= 64, F*K <= 384); the streaming formulation in ttcn.hip serves every other shape and the fp32 parity mode.
bool ttcn_fused_supported(int precision, int L, int F, int K);
int launch_ttcn3_fwd(int P, int L, int F, int K, const float* h2, const float* W3p, const float* b3p, const float* X,
                     const float* mask, const float* Tb, float* ctr, float* out, int out_ld, int flag_col, hipStream_t s);
int launch_ttcn3_bwd(int P, int L, int F, int K, const float* h2, const float* W3p, const float* b3p, const float* X,
                     const float* mask, const float* ctr, const float* out, const float* dout, int out_ld, float* dX,
                     float* dpool, float* dz2, float* gW3p, float* gb3p, hipStream_t s);
