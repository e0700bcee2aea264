// Small fused epilogue kernels of the MMF blocks, the masked-MSE loss and the fused clip+Adam update.
#pragma once
#include "common.hpp"

// zero the rows m whose window flag[m / div] == 0
int launch_mask_rows(float* x, int rows, int d, const unsigned char* flag, int div, hipStream_t s, void* xh = nullptr);   // xh: bf16 copy of the result

// MMF_XAttn_Add tail (fusions/MMF_XAttn_Add.py:93-102): LN over C, dropout, zero no-text windows, kappa blend
int launch_ln_blend_fwd(const float* delta, const float* Y, const unsigned char* mtxt, int BT, int T, int C,
                        const float* gamma, const float* beta, float kappa, float* xhat, float* rstd, float* Yout,
                        DropCfg drop, uint64_t site, hipStream_t s);
// dYout -> dY (= dYout/(1+kappa)), dn (grad wrt LN output, for the gamma/beta column sums), ddelta
int launch_ln_blend_bwd(const float* dYout, const unsigned char* mtxt, int BT, int T, int C, const float* gamma,
                        const float* xhat, const float* rstd, float kappa, float* dY, float* dn, float* ddelta,
                        DropCfg drop, uint64_t site, hipStream_t s);

// the forward output head of MMF_XAttn_Add in one kernel (xadd_head.hip): residual_head Linear(d -> C) on U, LayerNorm(C),
// dropout, no-text zeroing, kappa blend; saves xhat / rstd like launch_ln_blend_fwd.  C <= 16, d <= 1024, d % 4 == 0.
bool xadd_head_supported(int C, int d);
// bW: bias of the rows whose window has text, bWdead: bias of the others (U is zero there)
int launch_xadd_head_fwd(const float* U, const float* W, const float* bW, const float* bWdead, const float* Y, const unsigned char* mtxt, int BT, int T,
                         int C, int d, const float* gamma, const float* beta, float kappa, float* xhat, float* rstd, float* Yout,
                         DropCfg drop, uint64_t site, hipStream_t s);

int launch_mse_sums(const float* truth, const float* pred, const float* mask, int rows, int C, float* err_sum,
                    float* cnt, float* scratch, hipStream_t s);
int launch_mse_small(const float* truth, const float* pred, const float* mask, int rows, int C, const float* cnt_in,
                     float* err_sum, float* cnt, float* loss, float* dpred, float grad_scale, hipStream_t s);
// cnt known beforehand, C <= 64: several workgroups, last-to-finish adds the partial losses (partial: >= 64 floats, ticket: one
// zero-initialised word that the call leaves zero)
int launch_mse_counted(const float* truth, const float* pred, const float* mask, int rows, int C, const float* cnt, float* partial,
                       unsigned int* ticket, float* loss, float* dpred, float grad_scale, hipStream_t s);
int launch_mse_finish(const float* truth, const float* pred, const float* mask, int rows, int C, const float* err_sum,
                      const float* cnt, float* loss, float* dpred, float grad_scale, hipStream_t s);

int launch_adam(float* param, const float* grad, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                float wd, int step, float max_norm, float* norm_scratch, hipStream_t s);
int launch_adam_dev(float* param, const float* grad, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                    float wd, long long* step_dev, float max_norm, float* norm_scratch, unsigned long long* drop_dev,
                    hipStream_t s, int zero_grad = 0,      // zero_grad: the gradient buffer is left zero (it is written)
                    const int* skip = nullptr);            // skip: device word, non-zero = drop this step (no update, step not counted)

int launch_adam_sqnorm(const float* grad, size_t n, float* norm_scratch, long long* step_dev, unsigned long long* drop_dev,
                       hipStream_t s);
int launch_adam_apply(float* param, const float* grad, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                      float wd, int step, const long long* step_dev, float max_norm, const float* norm_scratch, void* twin,
                      hipStream_t s);

// clip + Adam as launches the caller places (include/immtsf.h immtsf_adam_prepare / _range / immtsf_guard_pack)
int launch_adam_prepare(const float* grad, const void* grad_h, size_t n, float* norm_scratch, long long* step_dev, unsigned long long* drop_dev,
                        int* pending, const int* err, const void* guard_h, const float* guard_f, int* skip_out, hipStream_t s);
int launch_adam_range(float* param, float* grad, const void* grad_h, float* m, float* v, size_t n, size_t lo, size_t hi, float lr, float b1,
                      float b2, float eps, float wd, const long long* step_dev, float max_norm, const float* norm_scratch, int zero_grad,
                      const int* skip, hipStream_t s);
int launch_guard_pack(const int* err, void* slot, int is_bf16, hipStream_t s);

// dst[i] = bf16(src[i]) (round to nearest even): refresh of a bf16 twin
int launch_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t s);
// dst[i] = float(src[i]) (exact)
int launch_bf16_to_f32(const void* src, float* dst, size_t n, hipStream_t s);
