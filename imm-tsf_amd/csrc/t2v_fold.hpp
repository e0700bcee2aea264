// Kernels of TTF_T2V_XAttn's FOLDED form (t2v_fold.hip; orchestration in fusion_blocks.hip).  reference: fusions/TTF_T2V_XAttn.py:120-182.
#pragma once
#include "common.hpp"

// ---- several independent small vector jobs as ONE launch (the parameter-only chains of the fold and of its chain rule) ----------
enum { VJ_MV = 0, VJ_MVT = 1, VJ_COPY = 2, VJ_RANK1 = 3 };
struct VecJob {
    int type;            // VJ_MV:   y[i] = scale * (sum_j W[i ld + j] x_i[j] + (b ? b[i] : 0)),  x_i = x + (i / xdiv) xld      i < rows, j < cols
                         // VJ_MVT:  y[j] = [y[j] +] scale * (sum_i W[i ld + j] (x ? x[i] : 1) + (b ? b[j] : 0))   (acc: added to y)    i < rows, j < cols
                         // VJ_COPY: y[i ldy + j] = scale * W[i ld + j] (W == null: 0)                                            i < rows, j < cols
                         // VJ_RANK1: y[i ldy + j] = base + scale * x[i] b[j]; base = acc 0: 0, 1: y[i ldy + j], 2: W[i ld + j]   (x == null: base only)
    const float* W;
    int ld;
    const float* x;
    const float* b;
    float* y;            // may be null when yh is given
    void* yh;            // optional bf16 copy of y (same indexing)
    int rows, cols, ldy;
    int xdiv, xld;       // VJ_MV only (xdiv <= 0: one x for every row)
    float scale;
    int acc;             // VJ_MVT: 1 = add to y; VJ_RANK1: the base (0 / 1 / 2)
};
constexpr int VJ_MAX = 20;
struct VecJobList {
    VecJob j[VJ_MAX];
    int n;
    VecJobList() : n(0) {}
    VecJob& add(int type, const float* W, int ld, const float* x, const float* b, float* y, int rows, int cols) {
        VecJob& v = j[n++];
        v.type = type; v.W = W; v.ld = ld; v.x = x; v.b = b; v.y = y; v.yh = nullptr; v.rows = rows; v.cols = cols; v.ldy = cols;
        v.xdiv = 0; v.xld = 0; v.scale = 1.f; v.acc = 0;
        return v;
    }
    // y[i ldy + j] = base + scale a[i] b[j]   (base 0: 0, 1: y, 2: src)
    VecJob& rank1(float* y, int ldy, int rows, int cols, int base, const float* a, const float* b, const float* src = nullptr, int lds = 0) {
        VecJob& v = add(VJ_RANK1, src, lds, a, b, y, rows, cols);
        v.ldy = ldy; v.acc = base;
        return v;
    }
};
int launch_vecjobs(const VecJobList& l, hipStream_t s);

// ---- the data path.  X (R, dmc) = [note embedding (d_m) ; Time2Vec (d/2)] per packed note, fp32 or bf16 (xh != 0); z (R, H d) the
// folded value rows per head; S / P (R, H) scores / softmax weights
struct T2VFoldDims { int B, T, H, d, N, dmc; };     // N: padded notes per window (dropout index only)
bool t2v_fold_shape_ok(int N, int T, int d, int H);
// S[r, h] = X[r, :] . U[h, :]   (U fp32, pitch ldu)
int launch_t2v_scores(const void* X, int x_is_bf16, int dmc, const float* U, int ldu, int H, const int* total, int max_rows, float* S, hipStream_t s);
// softmax over the window's notes per head (P written), attention dropout per (window, step, head, note), mix of the z rows:
// xpre[b, t, :] = (n_b > 0 ? sum_h sum_n a~ z[n, h] + b_o : 0) + q_res
int launch_t2v_mix_fwd(T2VFoldDims dm, const int* offsets, const int* rowmap, const float* S, const void* z, int z_is_bf16, const float* b_o,
                       const float* q_res, float* P, float* xpre, DropCfg drop, uint64_t site, hipStream_t s);
// the mix AND the LayerNorm + output dropout behind it in one launch (d % 8 == 0, d <= 1024: t2v_mix_wide_ok): x_hat (fp32 or, xhat_f ==
// null, bf16 alone), rstd, Z fp32 and / or bf16 -- x_pre never reaches memory
bool t2v_mix_wide_ok(int d);
int launch_t2v_mix_ln_fwd(T2VFoldDims dm, const int* offsets, const int* rowmap, const float* S, const void* z, int z_is_bf16, const float* b_o,
                          const float* q_res, float* P, const float* gamma, const float* beta, float eps, float* xhat_f, void* xhat_h,
                          float* rstd, float* z_f, void* z_h, DropCfg drop, uint64_t site, DropCfg odrop, uint64_t osite, hipStream_t s,
                          void* keep = nullptr);      // keep: optional (B, 2, 256) 64-bit words, the output dropout's keep bits (see MixLn)
extern int t2v_mix_bwd_wide;
// backward of the mix and the softmax: dz_aug (R, H d + 8): columns [h d, (h+1) d) = dz of head h, column H d + h = ds of head h, the
// rest 0 (fp32 or bf16 like z); dbo_part (B, d) = sum_t dx[b, t, :] of the windows with notes (0 otherwise)
int launch_t2v_mix_bwd(T2VFoldDims dm, const int* offsets, const int* rowmap, const float* P, const void* z, int z_is_bf16, const void* dx,
                       int dx_is_bf16, void* dz_aug, float* dbo_part, DropCfg drop, uint64_t site, hipStream_t s);

// ---- the MIX-FIRST form for long windows (t2v_premix.hip): one head, T <= 32, bf16 operands.  At (R, 32) bf16 = the dropped attention
// weights of (note, step); xbar (B T, dmc) bf16 = the mix of the raw rows; wbar (B T) = the weights' sums
bool t2v_premix_shape_ok(int T, int d, int H, int d_m);
size_t t2v_premix_du_scratch_floats(int dmc);
int t2v_premix_chunks(int N);      // note chunks of the weights kernel: wpart is (B, chunks, 32) floats
int launch_t2v_premix_weights(int B, int T, int N, const int* offsets, const int* rowmap, const float* S, float* P, void* At, float* wpart,
                              DropCfg drop, uint64_t site, hipStream_t s);
int launch_t2v_premix_fwd(int B, int T, int dmc, const int* offsets, const void* X, const void* At, void* xbar, hipStream_t s);
// x_pre (in place on the B T x d product x_bar W_tot^T): + b_o + c wbar + the residual query; windows without notes: the query alone;
// wbar (B T) = the chunks of wpart added up, written
int launch_t2v_premix_finish(int BT, int T, int N, int d, float* xpre, const float* b_o, const float* cvec, const float* wpart, float* wbar,
                             const float* q_res, const unsigned char* mtxt, hipStream_t s);
// from dxbar (B T, dmc) bf16 and dwbar (B T): g, ds (R), dXt (R, dmc - d_m) and du (dmc) -- four launches
int launch_t2v_premix_bwd(int B, int T, int N, int dmc, int d_m, const int* offsets, const int* total, const void* X, const void* At,
                          const float* P, const void* dxbar, const float* dwbar, const float* u, float* g, float* ds, float* dXt, float* du,
                          float* du_slab, hipStream_t s);
