// tPatchGNN time-aware patch encoder on MI355X: LearnableTE + TTCN (reference models/tPatchGNN.py:176-195).
//
// Per patch p (one variable's observations inside one time patch; L padded slots, mask marks the real ones):
//   te[l]   = [ws*t+bs ; sin(wp*t+bp)]                  (te_dim)         LearnableTE
//   X[l]    = [x[l] ; te[l]]                             (F = 1+te_dim)
//   filt[l] = W3 relu(W2 relu(W1 X[l] + b1) + b2) + b3   (F*K, K = ttcn_dim)   Filter_Generators
//   v[l,c]  = mask[l] ? filt[l,c] : -1e8 ; sm = softmax over l (per column c)
//   out[k]  = relu( sum_f sum_l X[l,f] * sm[l, k*F+f] + T_bias[k] )
//
// Mapping to the machine: the three filter-generator layers are row-wise linear maps over ALL R = P*L observation
// slots, so they (and their six backward products) run on the MFMA GEMM with the small weights zero-padded to
// 16-byte-aligned leading dimensions (F->Fp, K->Kp, F*K->NCp); what is specific to TTCN -- the masked softmax over
// each patch's slots and the meta-filter pooling -- is two HBM-bound streaming kernels over the (R, NCp) filter
// tensor (softmax weights overwrite the logits in place, the backward overwrites them with d(logits)).
// An all-masked (empty) patch gives the uniform softmax over identical pad rows, exactly like the reference.
#include "ttcn.hpp"
#include "../../include/immtsf.h"
#include "block_util.hpp"
#include "rowops.hpp"

int g_immtsf_ttcn_fused = 1;    // A/B switch: immtsf_debug_gemm_config bit 15 selects the streaming formulation below

namespace {

constexpr int LC = 32;   // slots per LDS chunk in the pooling backward

struct Dims { int P, L, F, K, NC, Fp, Kp, NCp, R; };
inline int r4(int x) { return (x + 31) / 32 * 32; }   // pad to 32: 16-byte aligned rows AND whole 32-deep GEMM K tiles
inline Dims mk_dims(int P, int L, int te_dim, int K) {
    Dims d;
    d.P = P; d.L = L; d.F = 1 + te_dim; d.K = K; d.NC = d.F * K;
    d.Fp = r4(d.F); d.Kp = r4(K); d.NCp = r4(d.NC); d.R = P * L;
    return d;
}

// X[r, 0] = x ; X[r, 1] = ws*t+bs ; X[r, 1+j] = sin(wp_j t + bp_j) ; padded columns = 0
__global__ __launch_bounds__(256) void build_x_kernel(Dims d, const float* __restrict__ x, const float* __restrict__ tt,
                                                       const float* ws, const float* bs, const float* __restrict__ wp,
                                                       const float* __restrict__ bp, float* __restrict__ X) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)d.R * d.Fp) return;
    const int r = (int)(idx / d.Fp), f = (int)(idx % d.Fp);
    const float t = tt[r];
    float v = 0.f;
    if (f == 0) v = x[r];
    else if (f == 1) v = fmaf(ws[0], t, bs[0]);
    else if (f < d.F) v = sinf(fmaf(wp[f - 2], t, bp[f - 2]));
    X[idx] = v;
}

// zero-padded, aligned copies of the small weights: W1p (Kp,Fp) b1p (Kp) W2p (Kp,Kp) b2p (Kp) W3p (NCp,Kp) b3p (NCp)
struct PackPtrs { const float *W1, *b1, *W2, *b2, *W3, *b3; float *W1p, *b1p, *W2p, *b2p, *W3p, *b3p; };
__global__ __launch_bounds__(256) void pack_w_kernel(Dims d, PackPtrs q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n1 = d.Kp * d.Fp, n2 = d.Kp * d.Kp, n3 = d.NCp * d.Kp;
    if (i < n1) { const int r = i / d.Fp, c = i % d.Fp; q.W1p[i] = (r < d.K && c < d.F) ? q.W1[r * d.F + c] : 0.f; }
    if (i < n2) { const int r = i / d.Kp, c = i % d.Kp; q.W2p[i] = (r < d.K && c < d.K) ? q.W2[r * d.K + c] : 0.f; }
    if (i < n3) { const int r = i / d.Kp, c = i % d.Kp; q.W3p[i] = (r < d.NC && c < d.K) ? q.W3[(size_t)r * d.K + c] : 0.f; }
    if (i < d.Kp) { q.b1p[i] = i < d.K ? q.b1[i] : 0.f; q.b2p[i] = i < d.K ? q.b2[i] : 0.f; }
    if (i < d.NCp) q.b3p[i] = i < d.NC ? q.b3[i] : 0.f;
}
// gradients back from the padded layout
struct UnpackPtrs { const float *gW1p, *gb1p, *gW2p, *gb2p, *gW3p, *gb3p; float *gW1, *gb1, *gW2, *gb2, *gW3, *gb3; };
__global__ __launch_bounds__(256) void unpack_g_kernel(Dims d, UnpackPtrs q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < d.K * d.F) q.gW1[i] = q.gW1p[(i / d.F) * d.Fp + i % d.F];
    if (i < d.K * d.K) q.gW2[i] = q.gW2p[(i / d.K) * d.Kp + i % d.K];
    if (i < d.NC * d.K) q.gW3[i] = q.gW3p[(size_t)(i / d.K) * d.Kp + i % d.K];
    if (i < d.K) { q.gb1[i] = q.gb1p[i]; q.gb2[i] = q.gb2p[i]; }
    if (i < d.NC) q.gb3[i] = q.gb3p[i];
}

// Pooling forward.  grid = P, block = ceil(NC/64)*64; thread c owns filter column c.  S (R, NCp): logits in, softmax
// weights out (in place).  ctr (P, NC) = sum_l sm*X, out (P, K) = relu(sum_f ctr + T_bias).
__global__ void pool_fwd_kernel(Dims d, float* __restrict__ S, const float* __restrict__ X, const float* __restrict__ mask,
                                const float* __restrict__ Tb, float* __restrict__ ctr, float* __restrict__ out, int out_ld,
                                int flag_col) {
    extern __shared__ float lds[];   // [NC] contributions
    const int p = blockIdx.x, c = threadIdx.x;
    const bool col = c < d.NC;
    const int fc = col ? c % d.F : 0;
    float* Sp = S + (size_t)p * d.L * d.NCp + c;
    const float* Xp = X + (size_t)p * d.L * d.Fp + fc;
    const float* mp = mask + (size_t)p * d.L;
    if (col) {
        float m = -INFINITY, s = 0.f;
        for (int l = 0; l < d.L; ++l) {
            const float mk = mp[l];
            const float v = Sp[(size_t)l * d.NCp] * mk + (1.f - mk) * (-1e8f);
            const float mn = fmaxf(m, v);
            s = s * expf(m - mn) + expf(v - mn);
            m = mn;
        }
        const float inv = 1.f / s;
        float acc = 0.f;
        for (int l = 0; l < d.L; ++l) {
            const float mk = mp[l];
            const float v = Sp[(size_t)l * d.NCp] * mk + (1.f - mk) * (-1e8f);
            const float sm = expf(v - m) * inv;
            Sp[(size_t)l * d.NCp] = sm;
            acc = fmaf(sm, Xp[(size_t)l * d.Fp], acc);
        }
        lds[c] = acc;
        ctr[(size_t)p * d.NC + c] = acc;
    }
    __syncthreads();
    if (c < d.K) {
        float a = Tb[c];
        for (int f = 0; f < d.F; ++f) a += lds[c * d.F + f];
        out[(size_t)p * out_ld + c] = fmaxf(a, 0.f);
    }
    if (flag_col >= 0 && c == 0) {          // patch-non-empty flag (models/tPatchGNN.py:268-270) written beside the embedding
        float any = 0.f;
        for (int l = 0; l < d.L; ++l) any += mp[l];
        out[(size_t)p * out_ld + flag_col] = any > 0.f ? 1.f : 0.f;
    }
}

// Pooling backward.  S: softmax weights in, d(logits) out (in place).  dX (R, Fp) = pooling-path gradient of X.
// dpool (P, K) = dout * [out > 0] is also written (for the T_bias column sum).
__global__ void pool_bwd_kernel(Dims d, float* __restrict__ S, const float* __restrict__ X, const float* __restrict__ mask,
                                const float* __restrict__ ctr, const float* __restrict__ out, const float* __restrict__ dout,
                                int out_ld, float* __restrict__ dX, float* __restrict__ dpool) {
    extern __shared__ float lds[];   // dp[K] | smt[LC][NC]
    float* dp = lds;
    float* smt = lds + d.K;
    const int p = blockIdx.x, c = threadIdx.x, nt = blockDim.x;
    const bool col = c < d.NC;
    const int fc = col ? c % d.F : 0, kc = col ? c / d.F : 0;
    if (c < d.K) {
        const float g = out[(size_t)p * out_ld + c] > 0.f ? dout[(size_t)p * out_ld + c] : 0.f;
        dp[c] = g;
        dpool[(size_t)p * d.K + c] = g;
    }
    __syncthreads();
    const float dpk = col ? dp[kc] : 0.f, ct = col ? ctr[(size_t)p * d.NC + c] : 0.f;
    float* Sp = S + (size_t)p * d.L * d.NCp + c;
    const float* Xp = X + (size_t)p * d.L * d.Fp;
    const float* mp = mask + (size_t)p * d.L;
    for (int l0 = 0; l0 < d.L; l0 += LC) {
        const int lcnt = min(LC, d.L - l0);
        if (col) {
            for (int l = 0; l < lcnt; ++l) {
                const float smd = Sp[(size_t)(l0 + l) * d.NCp] * dpk;
                smt[l * d.NC + c] = smd;
                Sp[(size_t)(l0 + l) * d.NCp] = smd * (Xp[(size_t)(l0 + l) * d.Fp + fc] - ct) * mp[l0 + l];
            }
        }
        __syncthreads();
        for (int i = c; i < lcnt * d.Fp; i += nt) {
            const int l = i / d.Fp, f = i % d.Fp;
            float a = 0.f;
            if (f < d.F) for (int k = 0; k < d.K; ++k) a += smt[l * d.NC + k * d.F + f];
            dX[((size_t)p * d.L + l0 + l) * d.Fp + f] = a;
        }
        __syncthreads();
    }
}

struct Ws {   // forward workspace = saved for backward
    float *X, *h1, *h2, *S, *ctr, *W1p, *b1p, *W2p, *b2p, *W3p, *b3p, *pack;
    size_t bytes;
};
Ws carve_ws(const Dims& d, void* base) {
    Carver k(base);
    Ws w;
    w.X = k.take<float>((size_t)d.R * d.Fp);
    w.h1 = k.take<float>((size_t)d.R * d.Kp);
    w.h2 = k.take<float>((size_t)d.R * d.Kp);
    w.S = k.take<float>((size_t)d.R * d.NCp);
    w.ctr = k.take<float>((size_t)d.P * (d.NC > d.F * 32 ? d.NC : d.F * 32));      // (P, F*32) in the on-chip formulation
    w.W1p = k.take<float>(d.Kp * d.Fp);
    w.b1p = k.take<float>(d.Kp);
    w.W2p = k.take<float>(d.Kp * d.Kp);
    w.b2p = k.take<float>(d.Kp);
    w.W3p = k.take<float>((size_t)d.NCp * d.Kp);
    w.b3p = k.take<float>(d.NCp);
    w.pack = k.take<float>(ttcn_full_pack_floats(d.F));
    w.bytes = k.bytes();
    return w;
}
struct Sc {
    float *dX, *dpool, *dz2, *dz1, *gW1p, *gb1p, *gW2p, *gb2p, *gW3p, *gb3p, *red, *slab;
    size_t bytes;
};
Sc carve_sc(const Dims& d, void* base) {
    Carver k(base);
    Sc s;
    s.dX = k.take<float>((size_t)d.R * d.Fp);
    s.dpool = k.take<float>((size_t)d.P * d.K);
    s.dz2 = k.take<float>((size_t)d.R * d.Kp);
    s.dz1 = k.take<float>((size_t)d.R * d.Kp);
    s.gW1p = k.take<float>(d.Kp * d.Fp);
    s.gb1p = k.take<float>(d.Kp);
    s.gW2p = k.take<float>(d.Kp * d.Kp);
    s.gb2p = k.take<float>(d.Kp);
    s.gW3p = k.take<float>((size_t)d.NCp * d.Kp);
    s.gb3p = k.take<float>(d.NCp);
    s.red = k.take<float>(64 * (d.NCp + 64) + 2 * 256 * d.F);
    s.slab = k.take<float>(ttcn_full_slab_floats(d.F));
    s.bytes = k.bytes();
    return s;
}

bool bad_dims(int P, int L, int te_dim, int K) { return P < 0 || L <= 0 || te_dim < 2 || K < 1 || (1 + te_dim) * K > 1024; }

}  // namespace

// The patch encoder wants (B*N*M, L) rows; the batch holds (B, M, L, N) tensors (values, time stamps, mask).  One launch for the
// three of them (as three permute + contiguous copies they are three stock element-wise launches in front of the backbone).
namespace {
__global__ __launch_bounds__(256) void patch_flatten3_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ c, int B, int M, int L, int N,
                                                              float* __restrict__ oa, float* __restrict__ ob, float* __restrict__ oc) {
    const size_t total = (size_t)B * M * L * N, o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int l = (int)(o % L);                     // o = ((b*N + n)*M + m)*L + l
    size_t r = o / L;
    const int m = (int)(r % M); r /= M;
    const int n = (int)(r % N);
    const size_t bb = r / N, i = ((bb * M + m) * L + l) * N + n;
    oa[o] = a[i];
    ob[o] = b[i];
    oc[o] = c[i];
}
}  // namespace

extern "C" {

size_t immtsf_ttcn_workspace_bytes(int32_t P, int32_t L, int32_t te_dim, int32_t ttcn_dim) {
    return bad_dims(P, L, te_dim, ttcn_dim) ? 0 : carve_ws(mk_dims(P, L, te_dim, ttcn_dim), nullptr).bytes;
}
size_t immtsf_ttcn_scratch_bytes(int32_t P, int32_t L, int32_t te_dim, int32_t ttcn_dim) {
    return bad_dims(P, L, te_dim, ttcn_dim) ? 0 : carve_sc(mk_dims(P, L, te_dim, ttcn_dim), nullptr).bytes;
}

int immtsf_patch_flatten3(const float* x, const float* tt, const float* mask, int32_t B, int32_t M, int32_t L, int32_t N, float* ox,
                          float* ott, float* omask, immtsf_stream_t stream) {
    if (!x || !tt || !mask || !ox || !ott || !omask || B < 0 || M <= 0 || L <= 0 || N <= 0) return IMMTSF_EINVAL;
    const size_t total = (size_t)B * M * L * N;
    if (total == 0) return IMMTSF_OK;
    hipLaunchKernelGGL(patch_flatten3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, tt, mask,
                       B, M, L, N, ox, ott, omask);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int immtsf_ttcn_forward(int32_t P, int32_t L, int32_t te_dim, int32_t ttcn_dim, int32_t precision, const float* x,
                        const float* tt, const float* mask, const immtsf_ttcn_params* p, float* out, int32_t out_ld,
                        int32_t flag_col, void* workspace, size_t workspace_bytes, immtsf_stream_t stream) {
    if (!x || !tt || !mask || !p || !out || !workspace) return IMMTSF_EINVAL;
    if (out_ld < ttcn_dim || flag_col >= out_ld || (flag_col >= 0 && flag_col < ttcn_dim)) return IMMTSF_EINVAL;
    if (bad_dims(P, L, te_dim, ttcn_dim)) return (1 + te_dim) * ttcn_dim > 1024 ? IMMTSF_EUNSUPPORTED : IMMTSF_EINVAL;
    if (P == 0) return IMMTSF_OK;
    const Dims d = mk_dims(P, L, te_dim, ttcn_dim);
    Ws w = carve_ws(d, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g_immtsf_ttcn_fused && ttcn_full_supported(precision, L, d.F, d.K))      // the whole encoder on chip, one kernel
        return launch_ttcn_full_fwd(P, L, d.F, d.K, x, tt, mask, p, w.pack, w.ctr, out, out_ld, flag_col, s);
    PackPtrs q{p->W1, p->b1, p->W2, p->b2, p->W3, p->b3, w.W1p, w.b1p, w.W2p, w.b2p, w.W3p, w.b3p};
    hipLaunchKernelGGL(pack_w_kernel, dim3(cdiv(d.NCp * d.Kp, 256)), dim3(256), 0, s, d, q);
    IMMTSF_LAUNCH_CHECK();
    const long nx = (long)d.R * d.Fp;
    hipLaunchKernelGGL(build_x_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, s, d, x, tt, p->te_scale_w, p->te_scale_b,
                       p->te_per_w, p->te_per_b, w.X);
    IMMTSF_LAUNCH_CHECK();
    {   // h1 = relu(X W1^T + b1)
        GemmArgs g = gemm_args(d.R, d.Kp, d.Fp, d.Fp, d.Fp, d.Kp);
        set_problem(g, 0, w.X, w.W1p, w.h1, w.b1p);
        g.act = 1;
        CHECK(immtsf_launch_gemm(GEMM_NT, precision, g, s));
    }
    {   // h2 = relu(h1 W2^T + b2)
        GemmArgs g = gemm_args(d.R, d.Kp, d.Kp, d.Kp, d.Kp, d.Kp);
        set_problem(g, 0, w.h1, w.W2p, w.h2, w.b2p);
        g.act = 1;
        CHECK(immtsf_launch_gemm(GEMM_NT, precision, g, s));
    }
    {   // filt = h2 W3^T + b3
        GemmArgs g = gemm_args(d.R, d.NCp, d.Kp, d.Kp, d.Kp, d.NCp);
        set_problem(g, 0, w.h2, w.W3p, w.S, w.b3p);
        CHECK(immtsf_launch_gemm(GEMM_NT, precision, g, s));
    }
    const int threads = cdiv(d.NC, 64) * 64;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(P), dim3(threads), d.NC * sizeof(float), s, d, w.S, w.X, mask, p->T_bias, w.ctr, out, out_ld, flag_col);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int immtsf_ttcn_backward(int32_t P, int32_t L, int32_t te_dim, int32_t ttcn_dim, int32_t precision, const float* x,
                         const float* tt, const float* mask, const immtsf_ttcn_params* p, const float* out,
                         const float* dout, int32_t out_ld, void* workspace, size_t workspace_bytes, void* scratch,
                         size_t scratch_bytes, const immtsf_ttcn_params* gr, int32_t te_accumulate, immtsf_stream_t stream) {
    if (!tt || !mask || !p || !out || !dout || !gr || !workspace || !scratch || out_ld < ttcn_dim) return IMMTSF_EINVAL;
    if (bad_dims(P, L, te_dim, ttcn_dim)) return IMMTSF_EINVAL;
    if (P == 0) return IMMTSF_OK;
    const Dims d = mk_dims(P, L, te_dim, ttcn_dim);
    Ws w = carve_ws(d, workspace);
    Sc sc = carve_sc(d, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (g_immtsf_ttcn_fused && ttcn_full_supported(precision, L, d.F, d.K))
        return launch_ttcn_full_bwd(P, L, d.F, d.K, x, tt, mask, p, w.pack, w.ctr, out, dout, out_ld, sc.slab, gr,
                                    static_cast<hipStream_t>(stream), te_accumulate);
    {   // the padded weight-gradient slab (gW1p .. gb3p, carved back to back) is zeroed once (split-K GEMMs / atomics)
        const size_t nbytes = (size_t)((char*)(sc.gb3p + d.NCp) - (char*)sc.gW1p);
        if (int rc = launch_fill(sc.gW1p, 0.f, nbytes / sizeof(float), s)) return rc;       // (a kernel, not a memset node: see gemm.hip)
    }
    Fork fk(s);
    {
        const int threads = cdiv(d.NC, 64) * 64;
        const size_t lds = (size_t)(d.K + LC * d.NC) * sizeof(float);
        if (lds > 160 * 1024) return IMMTSF_EUNSUPPORTED;
        hipLaunchKernelGGL(pool_bwd_kernel, dim3(P), dim3(threads), lds, s, d, w.S, w.X, mask, w.ctr, out, dout, out_ld, sc.dX, sc.dpool);
        IMMTSF_LAUNCH_CHECK();
    }
    CHECK(launch_colsum(sc.dpool, nullptr, P, nullptr, d.K, d.K, gr->T_bias, 0, sc.red, s));
    {   // layer 3: dW3 = dF^T h2 (+db3) ; dz2 = (dF W3) * [h2 > 0]
        GemmArgs h = gemm_args(d.NCp, d.Kp, d.R, d.NCp, d.Kp, d.Kp);
        set_problem(h, 0, w.S, w.h2, sc.gW3p, nullptr, sc.gb3p);
        h.c_prezeroed = 1;
        CHECK(immtsf_launch_gemm(GEMM_TN, precision, h, fk.fork()));
        GemmArgs g = gemm_args(d.R, d.Kp, d.NCp, d.NCp, d.Kp, d.Kp);
        set_problem(g, 0, w.S, w.W3p, sc.dz2, nullptr);
        g.relu_ref = w.h2; g.ld_ref = d.Kp;
        CHECK(immtsf_launch_gemm(GEMM_NN, precision, g, s));
    }
    {   // layer 2
        GemmArgs h = gemm_args(d.Kp, d.Kp, d.R, d.Kp, d.Kp, d.Kp);
        set_problem(h, 0, sc.dz2, w.h1, sc.gW2p, nullptr, sc.gb2p);
        h.c_prezeroed = 1;
        CHECK(immtsf_launch_gemm(GEMM_TN, precision, h, fk.fork()));
        GemmArgs g = gemm_args(d.R, d.Kp, d.Kp, d.Kp, d.Kp, d.Kp);
        set_problem(g, 0, sc.dz2, w.W2p, sc.dz1, nullptr);
        g.relu_ref = w.h1; g.ld_ref = d.Kp;
        CHECK(immtsf_launch_gemm(GEMM_NN, precision, g, s));
    }
    {   // layer 1: dW1 = dz1^T X (+db1) ; dX += dz1 W1
        GemmArgs h = gemm_args(d.Kp, d.Fp, d.R, d.Kp, d.Fp, d.Fp);
        set_problem(h, 0, sc.dz1, w.X, sc.gW1p, nullptr, sc.gb1p);
        h.c_prezeroed = 1;
        CHECK(immtsf_launch_gemm(GEMM_TN, precision, h, fk.fork()));
        GemmArgs g = gemm_args(d.R, d.Fp, d.Kp, d.Kp, d.Fp, d.Fp);
        set_problem(g, 0, sc.dz1, w.W1p, sc.dX, nullptr);
        g.accumulate = 1;
        CHECK(immtsf_launch_gemm(GEMM_NN, precision, g, s));
    }
    CHECK(fk.join());      // the padded weight gradients come from the side stream
    UnpackPtrs u{sc.gW1p, sc.gb1p, sc.gW2p, sc.gb2p, sc.gW3p, sc.gb3p, gr->W1, gr->b1, gr->W2, gr->b2, gr->W3, gr->b3};
    hipLaunchKernelGGL(unpack_g_kernel, dim3(cdiv(d.NC * d.K, 256)), dim3(256), 0, s, d, u);
    IMMTSF_LAUNCH_CHECK();
    // time-embedding parameters: same reduction as Time2Vec's backward, on dX[:, 1:F] with the slot times
    return launch_time2vec_bwd(tt, nullptr, nullptr, d.R, d.F - 1, p->te_per_w, p->te_per_b, sc.dX + 1, d.Fp, gr->te_scale_w,
                               gr->te_scale_b, gr->te_per_w, gr->te_per_b, sc.red, 256, s, te_accumulate);
}

}  // extern "C"
