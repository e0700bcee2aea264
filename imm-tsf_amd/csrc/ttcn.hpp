// tPatchGNN LearnableTE + TTCN patch encoder: see ttcn.hip (entry points are declared in include/immtsf.h).
#pragma once
#include "common.hpp"

// ttcn_full.hip: the WHOLE patch encoder (time embedding, three filter layers, masked softmax, pooling) of one patch on one
// CU, one kernel per direction (bf16 mode, ttcn_dim <= 32, 1 + te_dim <= 16, L <= 64).  `pack` / `slab`: caller scratch of
// ttcn_full_pack_floats / ttcn_full_slab_floats floats; ctr: (P, F*32) floats saved between forward and backward.
struct immtsf_ttcn_params;
bool ttcn_full_supported(int precision, int L, int F, int K);
size_t ttcn_full_pack_floats(int F);
size_t ttcn_full_slab_floats(int F);
int launch_ttcn_full_fwd(int P, int L, int F, int K, const float* x, const float* tt, const float* mask, const immtsf_ttcn_params* p,
                         float* pack, float* ctr, float* out, int out_ld, int flag_col, hipStream_t s);
int launch_ttcn_full_bwd(int P, int L, int F, int K, const float* x, const float* tt, const float* mask, const immtsf_ttcn_params* p,
                         const float* pack, const float* ctr, const float* out, const float* dout, int out_ld, float* slab,
                         const immtsf_ttcn_params* gr, hipStream_t s, int te_acc = 0);
