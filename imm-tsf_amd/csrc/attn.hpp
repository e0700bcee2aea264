// Attention-specific kernels: the single-query ragged cross-attention of TTF_T2V_XAttn (one learned query per
// window against that window's packed notes) and the row softmax used by the dense T x T attention of
// MMF_XAttn_Add / FullAttention (whose QK^T and A*V contractions run on the batched MFMA GEMM).
#pragma once
#include "common.hpp"

struct RaggedAttnDims {
    int B, T, H, hd, N;   // N = padded notes per window (only used for the dropout index)
};

// forward: KVp packed rows [R, 2d] = (k | v); qs = scaled query [d]; P out [R, H]; ctx out [B*T, d]
int launch_ragged_attn_fwd(RaggedAttnDims dm, const int* offsets, const int* rowmap, const float* KVp, const float* qs,
                           float* P, float* ctx, DropCfg drop, uint64_t site, hipStream_t s, void* ctx_h = nullptr,      // ctx_h: bf16 ctx (ctx may be null)
                           float* part = nullptr, const void* KVp_h = nullptr);      // KVp_h: read K | V from this bf16 image instead    // chunk partial sums, ragged_attn_part_floats(B, T, H * hd, N) floats (0 for short windows)
size_t ragged_attn_part_floats(int B, int T, int d, int N);
// backward: dctx [B*T, d] -> dKVp [R, 2d] (dk | dv), dqs_part [B, d]; dp_buf: scratch of ragged_attn_dp_floats(B, N, H, hd) floats
size_t ragged_attn_dp_floats(int B, int N, int H, int hd);
int launch_ragged_attn_bwd(RaggedAttnDims dm, const int* offsets, const int* rowmap, const float* KVp, const float* qs,
                           const float* P, const float* dctx, float* dKVp, float* dqs_part, float* dp_buf, DropCfg drop,
                           uint64_t site, hipStream_t s, void* dKVp_h = nullptr, const void* KVp_h = nullptr);   // dKVp_h: bf16 copy (dKVp may be null)

// rows = B*H*L, each of length S.  In place on `sc`: P = softmax(sc) (0 where !live[b]); A = P*dropscale
// written to `A` (may alias sc when drop.p == 0).
int launch_softmax_rows_fwd(float* sc, float* A, int B, int H, int L, int S, const unsigned char* live, DropCfg drop,
                            uint64_t site, int causal, hipStream_t s);
// dA -> dS in place: dP = dA*dropscale; dS = P*(dP - sum(P*dP))
int launch_softmax_rows_bwd(float* dA, const float* P, int B, int H, int L, int S, DropCfg drop, uint64_t site,
                            hipStream_t s);

// self-attention over sequences of at most IMMTSF_ATTN_SHORT_MAX positions on the packed in-projection output
// qkv (B, L, 3, H, E): out (B, L, H, E); backward recomputes the softmax and writes dqkv (B, L, 3, H, E)
int launch_attn_short_fwd(const float* qkv, int B, int L, int H, int E, float scale, int causal, DropCfg drop, uint64_t site,
                          float* out, hipStream_t s);
int launch_attn_short_bwd(const float* qkv, const float* dout, int B, int L, int H, int E, float scale, int causal, DropCfg drop,
                          uint64_t site, float* dqkv, hipStream_t s);

// dense T x T cross-attention for T <= 32 keys / queries per window (MMF_XAttn_Add over the prediction steps) with bf16 MFMA
// operands, one launch per direction: Q (B*T, H*hd), KV (B*T, 2*H*hd) = (k | v); P / A (B, H, T, T) are written by the forward
// and read by the backward; live[b] == 0: zero attention, zero context, zero gradients.  dKV_h: optional bf16 image of dKV.
// gen (optional, xattn_small_generates(hd, C)): Q = Y WQ^T + bq (Y (B*T, C), WQ (H*hd, C), bq (H*hd)) and, backward,
// dO = dd WO (dd (B*T, C), WO (C, H*hd)) are formed inside the kernels from their C-column inputs; Q / dO may then be null.
struct XattnGen { const float *Y, *WQ, *bq, *dd, *WO; int C; };
bool xattn_small_supported(int T, int H, int hd);       // T <= 32, hd % 16 == 0
bool xattn_small_generates(int hd, int C);              // hd <= 768, hd % 32 == 0, C in {4, 8, 12, 16}
int launch_xattn_small_fwd(const float* Q, const float* KV, const unsigned char* live, int B, int T, int H, int hd, float scale, DropCfg drop,
                           uint64_t site, float* Pm, float* Am, float* O, hipStream_t s, const XattnGen* gen = nullptr);
int launch_xattn_small_bwd(const float* Q, const float* KV, const float* dO, const float* Pm, const float* Am, const unsigned char* live,
                           int B, int T, int H, int hd, float scale, DropCfg drop, uint64_t site, float* dQ, float* dKV, void* dKV_h,
                           hipStream_t s, const XattnGen* gen = nullptr);
