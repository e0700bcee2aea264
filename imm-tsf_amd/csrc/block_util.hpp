// helpers shared by the block-level entry points
#pragma once
#include "../../include/immtsf.h"
#include "gemm.hpp"
#include "tail.hpp"
#include <string.h>

namespace {

// ---- bf16 mode keeps every GEMM operand as bf16 in HBM (gemm2.hip reads it by LDS-DMA): a logical matrix is an fp32
// image (for the row kernels, the parity mode and the round-1 GEMM), a bf16 image (for the bf16-in-memory GEMM), or both
struct Mat {
    float* f;
    void* h;
};
inline Mat mat(float* f, void* h = nullptr) { return Mat{f, h}; }
inline Mat cmat(const float* f, const void* h = nullptr) { return Mat{const_cast<float*>(f), const_cast<void*>(h)}; }
inline Mat mat_off(Mat m, size_t elems) {
    return Mat{m.f ? m.f + elems : nullptr, m.h ? static_cast<void*>(static_cast<unsigned short*>(m.h) + elems) : nullptr};
}
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
    // fp32 and / or bf16 image of an n-element matrix
    Mat take_mat(size_t n, bool want_f, bool want_h) {
        Mat m{nullptr, nullptr};
        if (want_f) m.f = take<float>(n);
        if (want_h) m.h = take<unsigned short>(n);
        return m;
    }
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

inline void set_problem2(GemmArgs& g, int i, Mat A, Mat B, Mat C, const float* bias, float* bias_grad = nullptr) {
    g.p[i].A = A.f; g.p[i].Ah = A.h;
    g.p[i].B = B.f; g.p[i].Bh = B.h;
    g.p[i].C = C.f; g.p[i].Ch = C.h;
    g.p[i].bias = bias; g.p[i].bias_grad = bias_grad;
    if (bias_grad) g.ones_col = 1;
}
// bf16 image of a weight matrix for the bf16-in-memory GEMM: its registered twin (FlatTrainer keeps one current), else a
// cast into the caller's `slot` (enqueued on `s`).  hf == false: fp32 only.
inline int weight_mat(bool hf, const float* W, size_t n, void* slot, hipStream_t s, Mat* out) {
    out->f = const_cast<float*>(W);
    out->h = nullptr;
    if (!hf || !W) return 0;
    const void* tw = immtsf_twin_lookup(W, n);
    if (tw && (reinterpret_cast<uintptr_t>(tw) & 15) == 0) { out->h = const_cast<void*>(tw); return 0; }
    if (!slot) return IMMTSF_EWORKSPACE;
    const int rc = launch_f32_to_bf16(W, slot, n, s);
    out->h = slot;
    return rc;
}

inline void set_problem(GemmArgs& g, int i, const float* A, const float* B, float* C, const float* bias, float* bias_grad = nullptr) {
    g.p[i].A = A; g.p[i].B = B; g.p[i].C = C; g.p[i].bias = bias; g.p[i].bias_grad = bias_grad;
    if (bias_grad) g.ones_col = 1;
}

// ---- fork/join onto a library-owned side stream (off unless immtsf_set_side_stream(1)) ---------------------------
// A linear layer's two backward GEMMs (data gradient NN, weight gradient TN) are independent and each under-fills
// the 256 CUs at the fusion shapes, so the weight-gradient GEMM is enqueued on a side stream that is forked from and
// joined back into the caller's stream inside the same call (works eagerly and under hipGraph capture: the side
// stream joins the capture through the event wait and is merged back before the call returns).  From the caller's
// point of view everything is still ordered on `stream`.
struct SideStream {
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int device = -1;
    bool ok = false;
};
inline SideStream& side_stream_state() {
    static thread_local SideStream st[16];      // per calling thread and device: two threads never share fork/join events
    int dev = 0;
    (void)hipGetDevice(&dev);
    SideStream& s = st[dev & 15];
    if (!s.ok && s.device != -2) {
        s.device = -2;   // tried
        if (hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&s.ev_fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&s.ev_join, hipEventDisableTiming) == hipSuccess)
            s.ok = true;
    }
    return s;
}
extern "C" int immtsf_side_stream_enabled(void);
class Fork {
public:
    explicit Fork(hipStream_t main) : main_(main), st_(side_stream_state()), used_(false) {
        active_ = st_.ok && immtsf_side_stream_enabled();
    }
    bool forking() const { return active_; }
    // stream for work that may run concurrently with what follows on the main stream; everything enqueued on the main
    // stream so far is a dependency
    hipStream_t fork() {
        if (!active_) return main_;
        if (hipEventRecord(st_.ev_fork, main_) != hipSuccess || hipStreamWaitEvent(st_.side, st_.ev_fork, 0) != hipSuccess) {
            active_ = false;
            return main_;
        }
        used_ = true;
        return st_.side;
    }
    // main stream waits for all side work
    int join() {
        if (!used_) return 0;
        used_ = false;
        hipError_t e = hipEventRecord(st_.ev_join, st_.side);
        if (e == hipSuccess) e = hipStreamWaitEvent(main_, st_.ev_join, 0);
        return e == hipSuccess ? 0 : (int)e;
    }
private:
    hipStream_t main_;
    SideStream& st_;
    bool active_, used_;
};

#define CHECK(x) do { int rc__ = (x); if (rc__ != 0) return rc__; } while (0)

inline bool bad_cfg(const immtsf_fusion_cfg* c) {
    return !c || c->B <= 0 || c->T <= 0 || c->d <= 0 || c->H <= 0 || (c->d % c->H) != 0 || c->precision < 0 ||
           c->precision > 1 || c->p_drop < 0.f || c->p_drop >= 1.f;
}


}  // namespace
