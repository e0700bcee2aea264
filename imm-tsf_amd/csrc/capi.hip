// extern "C" wrappers of the building-block kernels (tests, layers/ mirror, loss, optimizer).
#include "../../include/immtsf.h"
#include "attn.hpp"
#include "gemm.hpp"
#include "rowops.hpp"
#include "tail.hpp"
#include "block_util.hpp"
#include <atomic>

namespace {
inline DropCfg mk_drop(float p, uint64_t seed) {
    DropCfg d;
    d.seed = seed;
    d.p = p > 0.f ? p : 0.f;
    d.inv_keep = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    d.seed_dev = nullptr;
    return d;
}
}  // namespace

namespace { std::atomic<int> g_side_enabled{0}; }   // measured slower under hipGraph replay at the cfg2 shapes (3.18 vs 2.64 ms/step): off by default

extern "C" {

int immtsf_side_stream_enabled(void) { return g_side_enabled; }
int immtsf_set_side_stream(int32_t on) { g_side_enabled.store(on ? 1 : 0); return 0; }

int immtsf_gemm(int32_t layout, int32_t precision, const float* A, int32_t lda, const float* B, int32_t ldb, float* C,
                int32_t ldc, const float* bias, int32_t M, int32_t N, int32_t K, float alpha, int32_t accumulate,
                int32_t act, immtsf_stream_t stream) {
    if (!A || !B || !C || layout < 0 || layout > 2) return IMMTSF_EINVAL;
    GemmArgs g = gemm_args(M, N, K, lda, ldb, ldc);
    set_problem(g, 0, A, B, C, bias);
    g.alpha = alpha;
    g.accumulate = accumulate;
    g.act = act;
    return immtsf_launch_gemm(layout, precision, g, static_cast<hipStream_t>(stream));
}

int immtsf_linear_backward(int32_t precision, const float* x, const float* W, const float* dy, int32_t M, int32_t N,
                           int32_t K, float* dx, const float* relu_x, float* dW, float* db, int32_t grads_prezeroed,
                           immtsf_stream_t stream) {
    if (!dy || M <= 0 || N <= 0 || K <= 0 || (db && !dW) || (dx && !W) || (dW && !x)) return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // a small layer (tPatchGNN's temporal aggregation): both products in one launch, exact fp32 (linear_small.hip)
    if (linear_small_ok(M, N, K) && (!dW || grads_prezeroed)) return launch_linear_small_bwd(x, W, dy, M, N, K, dx, relu_x, dW, db, s);
    if (dx) {                   // dx (M,K) = dy (M,N) @ W (N,K)
        GemmArgs g = gemm_args(M, K, N, N, K, K);
        set_problem(g, 0, dy, W, dx, nullptr);
        g.relu_ref = relu_x;
        g.ld_ref = K;
        if (int rc = immtsf_launch_gemm(GEMM_NN, precision, g, s)) return rc;
    }
    if (dW) {                   // dW (N,K) = dy^T (N,M) @ x (M,K); db = dy^T 1
        GemmArgs g = gemm_args(N, K, M, N, K, K);
        set_problem(g, 0, dy, x, dW, nullptr, db);
        g.c_prezeroed = grads_prezeroed ? 1 : 0;
        if (int rc = immtsf_launch_gemm(GEMM_TN, precision, g, s)) return rc;
    }
    return IMMTSF_OK;
}

int immtsf_linear_bf16_forward(int32_t nl, const float* x, void* x16, const float* const* W, void* const* w16, const float* const* b,
                               float* const* y, int32_t M, int32_t N, int32_t K, int32_t act, immtsf_stream_t stream) {
    if (nl < 1 || nl > 3 || !x || !x16 || !W || !y || M <= 0 || N <= 0 || K <= 0 || (act && nl != 1)) return IMMTSF_EINVAL;
    if ((K & 7) || (N & 7)) return IMMTSF_EUNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    CHECK(launch_f32_to_bf16(x, x16, (size_t)M * K, s));
    GemmArgs g = gemm_args(M, N, K, K, K, N);
    g.nprob = nl;
    g.act = act;
    for (int i = 0; i < nl; ++i) {
        if (!W[i] || !y[i]) return IMMTSF_EINVAL;
        Mat Wm;
        CHECK(weight_mat(true, W[i], (size_t)N * K, w16 ? w16[i] : nullptr, s, &Wm));
        set_problem2(g, i, cmat(x, x16), Wm, mat(y[i]), b ? b[i] : nullptr);
    }
    return immtsf_launch_gemm(GEMM_NT, 1, g, s);
}

int immtsf_linear_bf16_backward(int32_t nl, const void* x16, const float* const* W, void* const* w16, const float* const* dy, void* dy16,
                                float* dx, float* const* dW, float* const* db, int32_t M, int32_t N, int32_t K,
                                int32_t grads_prezeroed, immtsf_stream_t stream) {
    if (nl < 1 || nl > 3 || !x16 || !W || !dy || !dy16 || M <= 0 || N <= 0 || K <= 0) return IMMTSF_EINVAL;
    if ((K & 7) || (N & 7)) return IMMTSF_EUNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* d16 = static_cast<unsigned short*>(dy16);
    for (int i = 0; i < nl; ++i) {
        if (!dy[i] || !W[i]) return IMMTSF_EINVAL;
        // (bit 1 of grads_prezeroed: dy16 holds the images already -- a second call that only forms the weight gradients, on another stream)
        if (!(grads_prezeroed & 2)) CHECK(launch_f32_to_bf16(dy[i], d16 + (size_t)i * M * N, (size_t)M * N, s));
    }
    if (dx) {                   // dx (M, K) = sum_i dy_i (M, N) @ W_i (N, K)
        // (as ONE launch -- the layers as problems of grid.z adding into a zeroed dx by fp32 atomics -- PatchTST's q | k | v step got 50 us
        // SLOWER: 5.9 M atomics on 7.9 MB; the accumulating chain stays)
        for (int i = 0; i < nl; ++i) {
            Mat Wm;
            CHECK(weight_mat(true, W[i], (size_t)N * K, w16 ? w16[i] : nullptr, s, &Wm));
            GemmArgs g = gemm_args(M, K, N, N, K, K);
            set_problem2(g, 0, cmat(dy[i], d16 + (size_t)i * M * N), Wm, mat(dx), nullptr);
            g.accumulate = i > 0 ? 1 : 0;
            CHECK(immtsf_launch_gemm(GEMM_NN, 1, g, s));
        }
    }
    if (dW) {                   // dW_i (N, K) = dy_i^T (N, M) @ x (M, K); db_i = dy_i^T 1 -- one grouped launch
        GemmArgs wg[3];
        int n = 0;
        for (int i = 0; i < nl; ++i) {
            if (!dW[i]) continue;
            GemmArgs h = gemm_args(N, K, M, N, K, K);
            set_problem2(h, 0, cmat(dy[i], d16 + (size_t)i * M * N), cmat(nullptr, x16), mat(dW[i]), nullptr, db ? db[i] : nullptr);
            h.c_prezeroed = (grads_prezeroed & 1) ? 1 : 0;
            wg[n++] = h;
        }
        CHECK(immtsf_launch_gemm_tn_list(1, wg, n, s));
    }
    return IMMTSF_OK;
}

int immtsf_time2vec_forward(const float* t, int32_t rows, int32_t d, const float* w0, const float* b0, const float* w,
                            const float* b, float* out, immtsf_stream_t stream) {
    if (!t || !w0 || !b0 || !out || d < 1 || (d > 1 && (!w || !b))) return IMMTSF_EINVAL;
    return launch_time2vec_fwd(t, nullptr, nullptr, rows, d, w0, b0, w, b, out, d, static_cast<hipStream_t>(stream));
}

int immtsf_time2vec_backward(const float* t, int32_t rows, int32_t d, const float* w, const float* b, const float* dout,
                             float* dw0, float* db0, float* dw, float* db, float* scratch, int32_t accumulate,
                             immtsf_stream_t stream) {
    if (!t || !dout || !dw0 || !db0 || !scratch || d < 1 || (d > 1 && (!w || !b || !dw || !db))) return IMMTSF_EINVAL;
    return launch_time2vec_bwd(t, nullptr, nullptr, rows, d, w, b, dout, d, dw0, db0, dw, db, scratch, 64,
                               static_cast<hipStream_t>(stream), accumulate);
}

int immtsf_f32_to_bf16(const float* src, void* dst, size_t n, immtsf_stream_t stream) {
    if (!src || !dst) return IMMTSF_EINVAL;
    return launch_f32_to_bf16(src, dst, n, static_cast<hipStream_t>(stream));
}

int immtsf_bf16_to_f32(const void* src, float* dst, size_t n, immtsf_stream_t stream) {
    if (!src || !dst) return IMMTSF_EINVAL;
    return launch_bf16_to_f32(src, dst, n, static_cast<hipStream_t>(stream));
}

int immtsf_gemm_batched(int32_t layout, int32_t precision, const float* A, int32_t lda, int64_t sA_o, int64_t sA_i,
                        const float* B, int32_t ldb, int64_t sB_o, int64_t sB_i, float* C, int32_t ldc, int64_t sC_o,
                        int64_t sC_i, int32_t n_outer, int32_t n_inner, int32_t M, int32_t N, int32_t K, float alpha,
                        immtsf_stream_t stream) {
    if (!A || !B || !C || layout < 0 || layout > 2 || n_outer <= 0 || n_inner <= 0) return IMMTSF_EINVAL;
    GemmArgs g = gemm_args(M, N, K, lda, ldb, ldc);
    set_problem(g, 0, A, B, C, nullptr);
    g.alpha = alpha;
    g.nbatch = n_outer * n_inner;
    g.batch_inner = n_inner;
    g.sA_o = sA_o; g.sA_i = sA_i; g.sB_o = sB_o; g.sB_i = sB_i; g.sC_o = sC_o; g.sC_i = sC_i;
    if (g.nbatch == 1) { g.nbatch = 1; }
    return immtsf_launch_gemm(layout, precision, g, static_cast<hipStream_t>(stream));
}

int immtsf_softmax_rows_forward(float* sc, float* A, int32_t B, int32_t H, int32_t L, int32_t S, const uint8_t* live,
                                float p_drop, uint64_t seed, uint64_t site, int32_t causal, const uint64_t* seed_step_dev,
                                immtsf_stream_t stream) {
    if (!sc || !A) return IMMTSF_EINVAL;
    DropCfg d = mk_drop(p_drop, seed);
    d.seed_dev = seed_step_dev;
    return launch_softmax_rows_fwd(sc, A, B, H, L, S, live, d, site, causal, static_cast<hipStream_t>(stream));
}

int immtsf_softmax_rows_backward(float* dA, const float* P, int32_t B, int32_t H, int32_t L, int32_t S, float p_drop,
                                 uint64_t seed, uint64_t site, const uint64_t* seed_step_dev, immtsf_stream_t stream) {
    if (!dA || !P) return IMMTSF_EINVAL;
    DropCfg d = mk_drop(p_drop, seed);
    d.seed_dev = seed_step_dev;
    return launch_softmax_rows_bwd(dA, P, B, H, L, S, d, site, static_cast<hipStream_t>(stream));
}

int immtsf_attention_short_forward(const float* qkv, int32_t B, int32_t L, int32_t H, int32_t E, float scale, int32_t causal,
                                   float p_drop, uint64_t seed, uint64_t site, const uint64_t* seed_step_dev, float* out,
                                   immtsf_stream_t stream) {
    if (!qkv || !out || B < 0 || L <= 0 || H <= 0 || E <= 0) return IMMTSF_EINVAL;
    DropCfg d = mk_drop(p_drop, seed);
    d.seed_dev = seed_step_dev;
    return launch_attn_short_fwd(qkv, B, L, H, E, scale, causal, d, site, out, static_cast<hipStream_t>(stream));
}

int immtsf_attention_short_backward(const float* qkv, const float* dout, int32_t B, int32_t L, int32_t H, int32_t E, float scale,
                                    int32_t causal, float p_drop, uint64_t seed, uint64_t site, const uint64_t* seed_step_dev,
                                    float* dqkv, immtsf_stream_t stream) {
    if (!qkv || !dout || !dqkv || B < 0 || L <= 0 || H <= 0 || E <= 0) return IMMTSF_EINVAL;
    DropCfg d = mk_drop(p_drop, seed);
    d.seed_dev = seed_step_dev;
    return launch_attn_short_bwd(qkv, dout, B, L, H, E, scale, causal, d, site, dqkv, static_cast<hipStream_t>(stream));
}

int immtsf_layernorm_forward(const float* x, int32_t rows, int32_t d, const float* gamma, const float* beta, float eps,
                             float* xhat, float* rstd, float* z, float p_drop, uint64_t seed, uint64_t site,
                             immtsf_stream_t stream) {
    if (!x || !gamma || !beta || !z || d <= 0) return IMMTSF_EINVAL;
    return launch_layernorm_fwd(x, rows, d, gamma, beta, eps, xhat, rstd, z, mk_drop(p_drop, seed), site,
                                static_cast<hipStream_t>(stream));
}

int immtsf_layernorm_backward(float* dz_dy, int32_t rows, int32_t d, const float* gamma, const float* xhat,
                              const float* rstd, float* dx, float* dgamma, float* dbeta, float* scratch, float p_drop,
                              uint64_t seed, uint64_t site, immtsf_stream_t stream) {
    if (!dz_dy || !gamma || !xhat || !rstd || !dx || d <= 0) return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    CHECK(launch_layernorm_bwd(dz_dy, rows, d, gamma, xhat, rstd, dx, mk_drop(p_drop, seed), site, s));
    if (dgamma && dbeta) {
        if (!scratch) return IMMTSF_EINVAL;
        return launch_colsum2(dz_dy, xhat, rows, d, d, dgamma, dbeta, scratch, s);
    }
    if (dgamma) {
        if (!scratch) return IMMTSF_EINVAL;
        CHECK(launch_colsum(dz_dy, xhat, rows, nullptr, d, d, dgamma, 0, scratch, s));
    }
    if (dbeta) {
        if (!scratch) return IMMTSF_EINVAL;
        CHECK(launch_colsum(dz_dy, nullptr, rows, nullptr, d, d, dbeta, 0, scratch, s));
    }
    return IMMTSF_OK;
}

int immtsf_dropout_mask(uint64_t seed, uint64_t site, uint64_t n, float p_drop, uint8_t* out, immtsf_stream_t stream) {
    if (!out) return IMMTSF_EINVAL;
    return launch_dropout_mask(seed, site, (size_t)n, p_drop, out, static_cast<hipStream_t>(stream));
}

int immtsf_masked_mse_sums(const float* truth, const float* pred, const float* mask, int32_t rows, int32_t C,
                           float* err_sum, float* cnt, float* scratch, immtsf_stream_t stream) {
    if (!truth || !pred || !mask || !err_sum || !cnt || !scratch || rows < 0 || C <= 0) return IMMTSF_EINVAL;
    return launch_mse_sums(truth, pred, mask, rows, C, err_sum, cnt, scratch, static_cast<hipStream_t>(stream));
}

int immtsf_masked_mse(const float* truth, const float* pred, const float* mask, int32_t rows, int32_t C, const float* cnt_global,
                      float* err_sum, float* cnt, float* loss, float* dpred, float grad_scale, immtsf_stream_t stream) {
    if (!truth || !pred || !mask || !loss || rows < 0 || C <= 0) return IMMTSF_EINVAL;
    if ((int64_t)rows * C > IMMTSF_MSE_SMALL_MAX || C > 4096) return IMMTSF_EUNSUPPORTED;
    return launch_mse_small(truth, pred, mask, rows, C, cnt_global, err_sum, cnt, loss, dpred, grad_scale,
                            static_cast<hipStream_t>(stream));
}

int immtsf_masked_mse_counted(const float* truth, const float* pred, const float* mask, int32_t rows, int32_t C, const float* cnt_global,
                              float* scratch, float* loss, float* dpred, float grad_scale, immtsf_stream_t stream) {
    if (!truth || !pred || !mask || !cnt_global || !scratch || !loss || rows < 0 || C <= 0) return IMMTSF_EINVAL;
    if (C > 64) return IMMTSF_EUNSUPPORTED;
    return launch_mse_counted(truth, pred, mask, rows, C, cnt_global, scratch + 1, reinterpret_cast<unsigned int*>(scratch), loss, dpred,
                              grad_scale, static_cast<hipStream_t>(stream));
}

int immtsf_masked_mse_finish(const float* truth, const float* pred, const float* mask, int32_t rows, int32_t C,
                             const float* err_sum, const float* cnt, float* loss, float* dpred, float grad_scale,
                             immtsf_stream_t stream) {
    if (!truth || !pred || !mask || !err_sum || !cnt || rows < 0 || C <= 0) return IMMTSF_EINVAL;
    return launch_mse_finish(truth, pred, mask, rows, C, err_sum, cnt, loss, dpred, grad_scale,
                             static_cast<hipStream_t>(stream));
}

int immtsf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int32_t step, float max_norm,
                     float* norm_scratch, immtsf_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !norm_scratch || step < 1) return IMMTSF_EINVAL;
    return launch_adam(param, grad, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay, step, max_norm,
                       norm_scratch, static_cast<hipStream_t>(stream));
}

int immtsf_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int64_t* step_dev, float max_norm,
                         float* norm_scratch, uint64_t* dropout_step_dev, immtsf_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !norm_scratch || !step_dev) return IMMTSF_EINVAL;
    return launch_adam_dev(param, grad, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay,
                           reinterpret_cast<long long*>(step_dev), max_norm, norm_scratch,
                           reinterpret_cast<unsigned long long*>(dropout_step_dev), static_cast<hipStream_t>(stream));
}

int immtsf_adam_step_dev_zero(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int64_t* step_dev, float max_norm, float* norm_scratch,
                              uint64_t* dropout_step_dev, immtsf_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !norm_scratch || !step_dev) return IMMTSF_EINVAL;
    return launch_adam_dev(param, grad, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay,
                           reinterpret_cast<long long*>(step_dev), max_norm, norm_scratch,
                           reinterpret_cast<unsigned long long*>(dropout_step_dev), static_cast<hipStream_t>(stream), 1);
}

int immtsf_adam_step_guarded(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t* step_dev, float max_norm, float* norm_scratch,
                             uint64_t* dropout_step_dev, const int32_t* skip_flag, int32_t zero_grad, immtsf_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !norm_scratch || !step_dev || !skip_flag) return IMMTSF_EINVAL;
    return launch_adam_dev(param, grad, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay,
                           reinterpret_cast<long long*>(step_dev), max_norm, norm_scratch,
                           reinterpret_cast<unsigned long long*>(dropout_step_dev), static_cast<hipStream_t>(stream), zero_grad ? 1 : 0,
                           skip_flag);
}

int immtsf_adam_sqnorm(const float* grad, uint64_t n, float* norm_scratch, int64_t* step_dev, uint64_t* dropout_step_dev,
                       immtsf_stream_t stream) {
    if (!grad || !norm_scratch) return IMMTSF_EINVAL;
    return launch_adam_sqnorm(grad, (size_t)n, norm_scratch, reinterpret_cast<long long*>(step_dev),
                              reinterpret_cast<unsigned long long*>(dropout_step_dev), static_cast<hipStream_t>(stream));
}

int immtsf_adam_apply(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, uint64_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int32_t step, const int64_t* step_dev, float max_norm,
                      const float* norm_scratch, void* twin, immtsf_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !norm_scratch || (!step_dev && step < 1)) return IMMTSF_EINVAL;
    return launch_adam_apply(param, grad, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay, step,
                             reinterpret_cast<const long long*>(step_dev), max_norm, norm_scratch, twin,
                             static_cast<hipStream_t>(stream));
}

int immtsf_adam_prepare(const float* grad, const void* grad_h, uint64_t n, float* norm_scratch, int64_t* step_dev,
                        uint64_t* dropout_step_dev, int32_t* pending, const int32_t* err, const void* guard_h, const float* guard_f,
                        int32_t* skip_out, immtsf_stream_t stream) {
    if ((!grad && !grad_h) || !norm_scratch) return IMMTSF_EINVAL;
    return launch_adam_prepare(grad, grad_h, (size_t)n, norm_scratch, reinterpret_cast<long long*>(step_dev),
                               reinterpret_cast<unsigned long long*>(dropout_step_dev), pending, err, guard_h, guard_f, skip_out,
                               static_cast<hipStream_t>(stream));
}

int immtsf_adam_range(float* param, float* grad, const void* grad_h, float* exp_avg, float* exp_avg_sq, uint64_t n, uint64_t lo,
                      uint64_t hi, float lr, float beta1, float beta2, float eps, float weight_decay, const int64_t* step_dev,
                      float max_norm, const float* norm_scratch, int32_t zero_grad, const int32_t* skip, immtsf_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !norm_scratch || !step_dev || hi > n || lo > hi || (lo & 7)) return IMMTSF_EINVAL;
    return launch_adam_range(param, grad, grad_h, exp_avg, exp_avg_sq, (size_t)n, (size_t)lo, (size_t)hi, lr, beta1, beta2, eps,
                             weight_decay, reinterpret_cast<const long long*>(step_dev), max_norm, norm_scratch, zero_grad ? 1 : 0, skip,
                             static_cast<hipStream_t>(stream));
}

int immtsf_guard_pack(const int32_t* err, void* slot, int32_t is_bf16, immtsf_stream_t stream) {
    if (!err || !slot) return IMMTSF_EINVAL;
    return launch_guard_pack(err, slot, is_bf16 ? 1 : 0, static_cast<hipStream_t>(stream));
}

int immtsf_instance_norm(const float* x, int32_t B, int32_t L, int32_t C, float* xn, float* means, float* stdev, immtsf_stream_t stream) {
    if (!x || !xn || !means || !stdev || B <= 0 || L <= 0 || C <= 0) return IMMTSF_EINVAL;
    return launch_instance_norm(x, B, L, C, xn, means, stdev, static_cast<hipStream_t>(stream));
}

int immtsf_notes_stage(const float* emb, int32_t d_m, const int32_t* src_rows, const int32_t* total, int32_t max_rows, void* X_h, int32_t ldx,
                       const float* tau, const int32_t* rowmap, int32_t dt, const float* lin_w, const float* lin_b, const float* per_w,
                       const float* per_b, immtsf_stream_t stream) {
    if (!emb || !src_rows || !total || !X_h || !tau || !rowmap || !lin_w || !lin_b || !per_w || !per_b || d_m <= 0 || dt <= 0 || max_rows <= 0 ||
        ldx < d_m + dt)
        return IMMTSF_EINVAL;
    return launch_notes_stage(emb, d_m, src_rows, total, max_rows, d_m, X_h, ldx, tau, rowmap, dt, lin_w, lin_b, per_w, per_b, nullptr, ldx,
                              static_cast<unsigned short*>(X_h) + d_m, nullptr, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, 1.f,
                              static_cast<hipStream_t>(stream));
}

int immtsf_abi_sizes(int32_t* out, int32_t max) {
    const int32_t sz[IMMTSF_ABI_NSTRUCTS] = {
        (int32_t)sizeof(immtsf_fusion_cfg), (int32_t)sizeof(immtsf_t2v_params), (int32_t)sizeof(immtsf_recavg_params),
        (int32_t)sizeof(immtsf_xadd_params), (int32_t)sizeof(immtsf_gr_params), (int32_t)sizeof(immtsf_ttcn_params),
        (int32_t)sizeof(immtsf_gcn_params), (int32_t)sizeof(immtsf_decoder_params), (int32_t)sizeof(immtsf_time2vec_params),
        (int32_t)sizeof(immtsf_encoder_layer_cfg), (int32_t)sizeof(immtsf_encoder_layer_params), (int32_t)sizeof(immtsf_ffn_block_cfg),
        (int32_t)sizeof(immtsf_ffn_block_params), (int32_t)sizeof(immtsf_store), (int32_t)sizeof(immtsf_note_index),
        (int32_t)sizeof(immtsf_lowrank_grad)};
    if (!out || max <= 0) return IMMTSF_ABI_NSTRUCTS;
    for (int i = 0; i < IMMTSF_ABI_NSTRUCTS && i < max; ++i) out[i] = sz[i];
    return IMMTSF_ABI_NSTRUCTS;
}

}  // extern "C"
