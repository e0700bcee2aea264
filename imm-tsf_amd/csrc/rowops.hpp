// Row-wise / ragged-index kernels of the fusion path (HBM-bound pieces).  All launchers are asynchronous
// on `stream`, own no memory and return 0 or an error code.
#pragma once
#include "common.hpp"

// a2: note_mask[b,n] = (sum_e |V[b,n,e]|) > 0 ; nan_flag (device int, may be null) |= 1 if any NaN in V
int launch_note_mask(const float* V, int rows, int d_m, unsigned char* mask, int* nan_flag, hipStream_t s);
// lengths/offsets/rowmap/seg/M_txt from the mask.  offsets has B+1 entries, offsets[B] = total notes.
int launch_ragged_index(const unsigned char* mask, int B, int N, int* lengths, int* offsets, int* rowmap, int* seg,
                        unsigned char* mtxt, hipStream_t s, unsigned char* mtxt2 = nullptr);
// mask[b,n] = n < lengths[b]  (packed-input mode: the batch builder already knows every window's note count)
int launch_mask_from_lengths(const int* lengths, int B, int N, unsigned char* mask, hipStream_t s);
// gather packed rows: dst[r, 0:d_m] = src[rowmap[r], 0:d_m]   (r < *total)
int launch_gather_rows(const float* src, int ld_src, const int* rowmap, const int* total, int max_rows, int width,
                       float* dst, int ld_dst, hipStream_t s, void* dst_h = nullptr);
// a3 Time2Vec on packed rows: dst[r, j] = j==0 ? w0*tau+b0 : sin(w[j-1]*tau+b[j-1]),  tau = tau_pad[rowmap[r]]
int launch_time2vec_fwd(const float* tau_pad, const int* rowmap, const int* total, int max_rows, int d_tau,
                        const float* w0, const float* b0, const float* w, const float* b, float* dst, int ld_dst,
                        hipStream_t s, void* dst_h = nullptr);      // dst_h: optional bf16 copy (same ld); dst may then be null
// Time2Vec parameter gradients from dFeat (packed rows, ld): nslabs row slabs (0 = 32), scratch >= 2*nslabs*d_tau floats;
// rowmap/total may be null (rows 0..max_rows-1 used directly)
// gather + cast of the packed notes (bf16 image only), Time2Vec of their time stamps and the learned query's in-projection in one launch
int launch_notes_stage(const float* src, int ld_src, const int* gmap, const int* total, int max_rows, int width, void* dst_h, int ld_dst,
                       const float* tau_pad, const int* rowmap, int d_tau, const float* w0, const float* b0, const float* w, const float* b,
                       float* t_dst, int t_ld, void* t_dst_h, const float* W, int ldw, const float* x, const float* bias, int rows, int cols,
                       float* y, float* ys, float scale, hipStream_t s,
                       const float* W2 = nullptr, int ldw2 = 0, const float* x2 = nullptr, const float* bias2 = nullptr, int rows2 = 0,
                       int cols2 = 0, float* y2 = nullptr,
                       const float* score_u = nullptr, float* score_out = nullptr);      // (W2 ...: an optional second mat-vec y2 = W2 x2 + bias2 in the same launch)
int launch_query_t2v_bwd(const float* dqs_part, int B, int d, float scale, const float* Wq, int ldw, const float* Q, float* dWq, int ldg,
                         float* dbq, float* dQ, const float* tau_pad, const int* rowmap, const int* total, int max_rows, int d_tau,
                         const float* w, const float* b, const float* dfeat, int ld, float* dw0, float* db0, float* dw, float* db,
                         float* scratch, int nslabs, hipStream_t s, int accumulate = 0);
int launch_time2vec_bwd(const float* tau_pad, const int* rowmap, const int* total, int max_rows, int d_tau,
                        const float* w, const float* b, const float* dfeat, int ld, float* dw0, float* db0, float* dw,
                        float* db, float* scratch, int nslabs, hipStream_t s, int accumulate = 0);
// out[n] = sum_{m < M} X[m,n] * (Y ? Y[m,n] : 1)   (M may come from *dyn).  scratch >= 32*N floats.
int launch_colsum(const float* X, const float* Y, int M, const int* dyn, int N, int ld, float* out, int accumulate,
                  float* scratch, hipStream_t s, bool big_scratch = false);       // big_scratch: as for launch_colsum2 (k = 1)
// out_xy[n] = sum_m X*Y, out_x[n] = sum_m X in one pass (LayerNorm's gamma / beta gradients).  scratch >= 64*N floats.
// big_scratch: scratch holds colsum_scratch_floats(N, 2) floats -- enables the many-row path (M >= 4096: 16-byte loads, <= 256 slabs)
// sums of X * Y and of X (a LayerNorm's parameter gradients) and of Z (a bias gradient) over the same M rows of N <= 32 columns in one
// launch (M * N <= 2^17)
bool colsum_small_pair_ok(int M, int N);
int launch_colsum_small_pair(const float* X, const float* Y, const float* Z, int M, int N, int ld, float* out_xy, float* out_x, float* out_z,
                             hipStream_t s, int outputs_zero = 0);      // outputs_zero: the three outputs read zero (pre-zeroed sinks): spread over the chip, atomics
int launch_colsum2(const float* X, const float* Y, int M, int N, int ld, float* out_xy, float* out_x, float* scratch,
                   hipStream_t s, bool big_scratch = false);
inline size_t colsum_scratch_floats(size_t ncols, int k) { return (size_t)256 * k * ((ncols + 3) & ~(size_t)3) + 64 * 8; }
// out_xy = sum_m X*Y, out_x = sum_m X, out_z = sum_m Z in one pass; afterwards Z's rows with row_flag[m / flag_div] == 0 are zero
// (row_flag may be null) and Zh (may be null) holds Z's bf16 image.  scratch >= 64 * 3 * N floats.
int launch_colsum3(const float* X, const float* Y, float* Z, int M, int N, int ld, float* out_xy, float* out_x, float* out_z,
                   float* scratch, const unsigned char* row_flag, int flag_div, void* Zh, hipStream_t s, bool big_scratch = false);
// LayerNorm over the last dim with fused dropout: xhat, rstd saved; z = drop(xhat*gamma+beta)
int launch_layernorm_fwd(const float* x, int rows, int d, const float* gamma, const float* beta, float eps, float* xhat,
                         float* rstd, float* z, DropCfg drop, uint64_t site, hipStream_t s, void* zh = nullptr,      // zh: bf16 z (z may be null)
                         const float* res = nullptr, DropCfg pre = DropCfg{0, 0.f, 1.f, nullptr}, uint64_t presite = 0,
                         void* xhat_h = nullptr);        // xhat_h: x_hat as bf16 (xhat may then be null: the backward with sums reads either)
// (res != null: the row normalised is res + dropout_pre(x) -- LayerNorm(x_in + Dropout(branch)) of a post-norm block)
// in: dz (grad wrt z).  out: dy written IN PLACE over dz (dy = dz*dropscale), dx.
int launch_layernorm_bwd(float* dz_dy, int rows, int d, const float* gamma, const float* xhat, const float* rstd,
                         float* dx, DropCfg drop, uint64_t site, hipStream_t s, float* dbranch = nullptr,
                         DropCfg pre = DropCfg{0, 0.f, 1.f, nullptr}, uint64_t presite = 0);
// LayerNorm backward fused with the column sums that give d gamma / d beta (/ the sum of dx: K = 3, which also zeroes flagged rows of
// dx and writes its bf16 image, like launch_colsum3): IMMTSF_EUNSUPPORTED -> launch_layernorm_bwd + launch_colsum2 / 3
constexpr int kLnSlabsMax = 2048;
inline size_t ln_sums_scratch_floats(size_t d, int k) { return (size_t)kLnSlabsMax * k * ((d + 3) & ~(size_t)3) + 64 * 8; }      // `scratch` of the call below
int launch_layernorm_bwd_sums(float* dz_dy, int rows, int d, const float* gamma, const float* xhat, const float* rstd, float* dx,
                              DropCfg drop, uint64_t site, float* out_gw, float* out_gb, float* out_q, float* scratch,
                              const unsigned char* row_flag, int flag_div, void* dxh, hipStream_t s, const void* xhat_h = nullptr);
// x (B, L, C) -> (x - mean_t) / sqrt(var_t + 1e-5), means (B, C), stdev (B, C): one launch
int launch_instance_norm(const float* x, int B, int L, int C, float* xn, float* means, float* stdev, hipStream_t s);
// ... with the upstream gradient in low-rank form dy = G Wb (never written): see layernorm_bwd_lr_kernel
bool ln_lr_ok(int rows, int d, int PW);
int launch_layernorm_bwd_lr(const float* G, int ldg, int PW, const float* Wb, int rows, int d, const float* gamma, const float* xhat,
                            const void* xhat_h, const float* rstd, float* dx, void* dxh, DropCfg drop, uint64_t site, float* out_gw, float* out_gb,
                            float* out_q, float* scratch, const unsigned char* row_flag, int flag_div, hipStream_t s, const void* keep = nullptr,
                            int keep_T = 0);      // keep / keep_T: the forward's keep bits of this site (t2v_mix_ln_fwd_kernel), rows = windows x keep_T
// (xhat_h: the bf16 x_hat image of launch_layernorm_fwd instead of xhat; dx may be null when dxh is all the consumer reads)
bool ln_sums_compact_ok(int rows, int d);     // the sums kernel takes (rows, d) whatever the pointers: a forward may store x_hat as bf16 only      // dbranch = dx * dropout_pre mask
// y[i] = sum_j W[i,j] x[j] + b[i]  (tiny mat-vec, e.g. q = W_q Q_param + b_q), then scaled copy ys = y*scale
int launch_matvec(const float* W, int ldw, const float* x, const float* b, int rows, int cols, float* y, float* ys,
                  float scale, hipStream_t s, float* y_nobias = nullptr);      // y_nobias: W x without the bias
// y[j] = sum_i W[i,j] x[i]   (transposed mat-vec)
int launch_matvec_t(const float* W, int ldw, const float* x, int rows, int cols, float* y, int accumulate,
                    hipStream_t s);
// out[i,j] = a[i]*b[j]
int launch_outer(const float* a, const float* b, int rows, int cols, float* out, int ld, hipStream_t s);
// dst = alpha*src (n elements)  /  dst += alpha*src
int launch_axpy(const float* src, float alpha, float* dst, int n, int accumulate, hipStream_t s);
// backward of q = W_q Q + b_q, qs = scale q from per-window partials dqs_part (B, d): dW_q (d, d; written), db_q (d; written),
// dQ (d; ADDED to what it holds)
int launch_query_bwd(const float* dqs_part, int B, int d, float scale, const float* Wq, int ldw, const float* Q, float* dWq, int ldg,
                     float* dbq, float* dQ, hipStream_t s);
int launch_fill(float* dst, float v, size_t n, hipStream_t s);
// keep-mask export for tests: out[i] = 1 if element i of `site` is kept
int launch_dropout_mask(uint64_t seed, uint64_t site, size_t n, float p, unsigned char* out, hipStream_t s);
// the small reductions of MMF_XAttn_Add's parameter gradients (see rowops.hip): C <= 32 and M * C <= 2^17
bool head_sums_supported(int M, int C);
int launch_head_sums(const float* dn, const float* xhat, const float* ddelta, const unsigned char* flag, int flag_div, int M, int C,
                     float* out_xy, float* out_x, float* slive, hipStream_t s);
int launch_head_outer(const float* Wres, int d, const float* slive, const float* bout, int C, float* dbout, float* dWres, hipStream_t s);
