// MMF_GR_Add kernels: concat/split of [Y_ts ; E_txt], the per-window GRU recurrence (nn.GRU, gate order r,z,n),
// and the fused residual-head + LayerNorm(C) + dropout + sigmoid-gate blend tail.
#pragma once
#include "common.hpp"

int launch_concat2(const float* a, int wa, const float* b, int wb, int rows, float* out, hipStream_t s);
// da (may be null) += / = x[:, :wa] ; db = x[:, wa:]
int launch_split2(const float* x, int wa, int wb, int rows, float* da, int accumulate_a, float* db, hipStream_t s);

// gi [B*T, 3Hd] (input side incl. b_ih) -> r,z,n,hn (hn = W_hn h_prev + b_hn), h, hprev : each [B*T, Hd]
int launch_gru_fwd(int B, int T, int Hd, const float* gi, const float* w_hh, const float* b_hh, float* r, float* z,
                   float* n, float* hn, float* h, float* hprev, hipStream_t s);
// dh_in [B*T,Hd] (gradient from the head at every step) -> dgi [B*T,3Hd], dgh [B*T,3Hd]
int launch_gru_bwd(int B, int T, int Hd, const float* dh_in, const float* w_hh, const float* r, const float* z,
                   const float* n, const float* hn, const float* hprev, float* dgi, float* dgh, hipStream_t s);

// tail forward: delta = W_r h + b_r ; LN(C) ; dropout ; g = sigmoid(gl) (1 where no text) ; out = Y + (1-g)*dd
int launch_gr_tail_fwd(int BT, int T, int C, int Hd, const float* h, const float* res_w, const float* res_b,
                       const float* gamma, const float* beta, const float* gl, const float* Y, const unsigned char* mtxt,
                       float* xhat, float* rstd, float* g_out, float* dd_out, float* Yout, DropCfg drop, uint64_t site,
                       hipStream_t s);
// tail backward: dYout -> dn (grad wrt LN output), ddelta, dgl, dh_in (= W_r^T ddelta)
int launch_gr_tail_bwd(int BT, int T, int C, int Hd, const float* dYout, const float* res_w, const float* gamma,
                       const float* xhat, const float* rstd, const float* g, const float* dd, const unsigned char* mtxt,
                       float* dn, float* ddelta, float* dgl, float* dh_in, DropCfg drop, uint64_t site, hipStream_t s);
