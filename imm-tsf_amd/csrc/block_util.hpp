// helpers shared by the block-level entry points
#pragma once
#include "../../include/immtsf.h"
#include "gemm.hpp"
#include <string.h>

namespace {

struct Carver {
    char* base;
    size_t off;
    explicit Carver(void* b) : base(static_cast<char*>(b)), off(0) {}
    template <typename T> T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
    size_t bytes() const { return (off + 255) & ~size_t(255); }
};

inline DropCfg drop_of(const immtsf_fusion_cfg* c) {
    DropCfg d;
    d.seed = c->seed;
    d.p = (c->training && c->p_drop > 0.f) ? c->p_drop : 0.f;
    d.inv_keep = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    d.seed_dev = c->seed_step_dev;
    return d;
}

inline GemmArgs gemm_args(int M, int N, int K, int lda, int ldb, int ldc) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.nprob = 1;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = 1.f;
    g.row_flag_div = 1;
    g.nbatch = 1;
    g.batch_inner = 1;
    return g;
}
inline void prezeroed(GemmArgs& g, const immtsf_fusion_cfg* c) { g.c_prezeroed = c->grads_prezeroed ? 1 : 0; }
inline void set_problem(GemmArgs& g, int i, const float* A, const float* B, float* C, const float* bias, float* bias_grad = nullptr) {
    g.p[i].A = A; g.p[i].B = B; g.p[i].C = C; g.p[i].bias = bias; g.p[i].bias_grad = bias_grad;
    if (bias_grad) g.ones_col = 1;
}

#define CHECK(x) do { int rc__ = (x); if (rc__ != 0) return rc__; } while (0)

inline bool bad_cfg(const immtsf_fusion_cfg* c) {
    return !c || c->B <= 0 || c->T <= 0 || c->d <= 0 || c->H <= 0 || (c->d % c->H) != 0 || c->precision < 0 ||
           c->precision > 1 || c->p_drop < 0.f || c->p_drop >= 1.f;
}


}  // namespace
