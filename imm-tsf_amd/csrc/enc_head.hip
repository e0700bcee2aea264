// The attention half of tPatchGNN's transformer layer (models/tPatchGNN.py:118-121, 200-205: nn.TransformerEncoderLayer, post-norm,
// d_model = hid_dim = 32, sequences of M <= 8 patches) as ONE kernel per direction:
//
//     qkv = x W_in^T + b_in ;  a = softmax(q k^T / sqrt(E)) v  (attention-weight dropout) ;  sa = a W_o^T + b_o ;  x1 = LayerNorm1(x + Dropout(sa))
//
// At d_model 32 every product of this half is a few thousand FMAs per sequence, but as launches it was a GEMM, the short-sequence
// attention, a GEMM and a LayerNorm forward (30 us) and seven launches backward (LayerNorm backward + column sums, two data-gradient
// and two weight-gradient GEMMs, the attention backward: 65 us at 64 windows, all of it on the backbone's dependent chain; at 4096
// windows the 65 536-row skinny products and their reduce launches were ~0.5 ms).  Here a row is a group of 32 lanes (lane = column),
// the two weight matrices live in LDS (pitch 33: row and column walks are both conflict-free), a sequence's rows exchange k, v, q, the
// attention weights and their gradients through LDS, and the parameter gradients stay in registers over all the sequences a
// persistent workgroup walks, then meet in an LDS image and leave as one slab per workgroup (summed by a small second launch).
// The backward recomputes the forward from x: nothing but x1, xhat1 and rstd1 (which the feed-forward half and the LayerNorm
// backward need anyway) is saved.  Exact fp32 in both precision modes; dropout sites and indices as in the composed path
// (attention weights: site, ((b H + h) S + l) S + s; dropout1: site + 1, row * 32 + column).
#include "../../include/immtsf.h"
#include "common.hpp"
#include "enc_head.hpp"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int EH_D = 32;
constexpr int EH_P = 33;                 // LDS pitch of the weight images
constexpr int EH_S = 8;                  // longest sequence
constexpr int EH_NG = 3 * EH_D * EH_D + EH_D * EH_D + 3 * EH_D + EH_D + 2 * EH_D;      // floats of a gradient slab: W_in, W_o, b_in, b_o, ln_w, ln_b

struct EhDims { int Bs, S, H, E, spb; float scale, eps; };
struct EhP { const float *in_w, *in_b, *out_w, *out_b, *ln_w, *ln_b; };

struct DropE { uint64_t seed; float p, inv_keep; };
__device__ __forceinline__ DropE drop_e(const DropCfg& d) {
    DropE l;
    l.seed = d.seed + (d.seed_dev ? *d.seed_dev : 0ull);
    l.p = d.p; l.inv_keep = d.inv_keep;
    return l;
}
__device__ __forceinline__ float drop1(const DropE& d, uint64_t site, uint64_t idx) { return dropout_scale(d.seed, site, idx, d.p, d.inv_keep); }

// LDS layout (floats): Win [96][33] | Wo [32][33] | per row group (8): xs[32] qs[32] ks[32] vs[32] as[32] t0[32] t1[96] | As[8][4][8] Ds[8][4][8]
struct EhLds {
    float *Win, *Wo, *xs, *qs, *ks, *vs, *as, *t0, *t1, *As, *Ds;
};
__device__ __forceinline__ EhLds eh_lds(float* base) {
    EhLds l;
    l.Win = base;
    l.Wo = l.Win + 3 * EH_D * EH_P;
    l.xs = l.Wo + EH_D * EH_P;
    l.qs = l.xs + 8 * EH_D;
    l.ks = l.qs + 8 * EH_D;
    l.vs = l.ks + 8 * EH_D;
    l.as = l.vs + 8 * EH_D;
    l.t0 = l.as + 8 * EH_D;
    l.t1 = l.t0 + 8 * EH_D;
    l.As = l.t1 + 8 * 3 * EH_D;
    l.Ds = l.As + 8 * 4 * EH_S;
    return l;
}
constexpr int EH_LDS_FLOATS = 3 * EH_D * EH_P + EH_D * EH_P + 6 * 8 * EH_D + 8 * 3 * EH_D + 2 * 8 * 4 * EH_S;

__device__ __forceinline__ void eh_stage_weights(const EhP& p, const EhLds& L) {
    for (int i = threadIdx.x; i < 3 * EH_D * EH_D; i += 256) L.Win[(i >> 5) * EH_P + (i & 31)] = p.in_w[i];
    for (int i = threadIdx.x; i < EH_D * EH_D; i += 256) L.Wo[(i >> 5) * EH_P + (i & 31)] = p.out_w[i];
}

// the forward of one row up to the LayerNorm input r = x + dropout1(sa).  On return: q, k, v of the row (registers), A / Ad = the
// attention weights of the row's head before / after dropout (every lane of a head group holds them), a (attention output, also in
// L.as), L.qs / ks / vs / xs hold the sequence's rows.  Contains block barriers: every thread of the workgroup must call it.
template <bool KEEP_Q>
__device__ __forceinline__ float eh_row_forward(const EhDims& d, const EhP& p, const EhLds& L, const DropE& da, const DropE& dd, uint64_t site,
                                                bool on, int rg, int j, int rg0, long seq, int l, float xj, float& q, float& k, float& v,
                                                float (&A)[EH_S], float (&Ad)[EH_S], float& a) {
    const int S = d.S, E = d.E, h = j / E;
    L.xs[rg * EH_D + j] = xj;
    __syncthreads();
    q = p.in_b[j]; k = p.in_b[EH_D + j]; v = p.in_b[2 * EH_D + j];
#pragma unroll 8
    for (int c = 0; c < EH_D; ++c) {
        const float xv = L.xs[rg * EH_D + c];
        q = fmaf(xv, L.Win[j * EH_P + c], q);
        k = fmaf(xv, L.Win[(EH_D + j) * EH_P + c], k);
        v = fmaf(xv, L.Win[(2 * EH_D + j) * EH_P + c], v);
    }
    L.ks[rg * EH_D + j] = k;
    L.vs[rg * EH_D + j] = v;
    if (KEEP_Q) L.qs[rg * EH_D + j] = q;
    __syncthreads();
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < EH_S; ++s) {
        A[s] = 0.f; Ad[s] = 0.f;
        if (s < S) {
            A[s] = d.scale * group_sum(q * L.ks[(rg0 + s) * EH_D + j], E);
            mx = fmaxf(mx, A[s]);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < EH_S; ++s)
        if (s < S) { A[s] = expf(A[s] - mx); sum += A[s]; }
    const float inv = 1.f / sum;
    a = 0.f;
#pragma unroll
    for (int s = 0; s < EH_S; ++s)
        if (s < S) {
            A[s] *= inv;
            Ad[s] = on ? A[s] * drop1(da, site, (((uint64_t)seq * d.H + h) * S + l) * S + s) : 0.f;
            a = fmaf(Ad[s], L.vs[(rg0 + s) * EH_D + j], a);
        }
    L.as[rg * EH_D + j] = a;
    __syncthreads();
    float sa = p.out_b[j];
#pragma unroll 8
    for (int c = 0; c < EH_D; ++c) sa = fmaf(L.as[rg * EH_D + c], L.Wo[j * EH_P + c], sa);
    const uint64_t row = (uint64_t)seq * S + l;
    return on ? fmaf(sa, drop1(dd, site + 1, row * EH_D + j), xj) : 0.f;
}

__global__ __launch_bounds__(256) void enc_head32_fwd_kernel(EhDims d, EhP p, const float* __restrict__ x, float* __restrict__ x1,
                                                              float* __restrict__ xhat, float* __restrict__ rstd, DropCfg dca, DropCfg dcd,
                                                              uint64_t site) {
    __shared__ __attribute__((aligned(16))) float lds[EH_LDS_FLOATS];
    const EhLds L = eh_lds(lds);
    eh_stage_weights(p, L);
    const DropE da = drop_e(dca), dd = drop_e(dcd);
    const int rg = threadIdx.x >> 5, j = threadIdx.x & 31, sl = rg / d.S, l = rg - sl * d.S, rg0 = sl < d.spb ? sl * d.S : 0;      // (idle groups read group 0's rows)
    const float gm = p.ln_w[j], bt = p.ln_b[j];
    __syncthreads();
    const long npass = ((long)d.Bs + d.spb - 1) / d.spb;
    for (long ps = blockIdx.x; ps < npass; ps += gridDim.x) {
        const long seq = ps * d.spb + sl;
        const bool on = sl < d.spb && seq < d.Bs;
        const long row = seq * d.S + l;
        const float xj = on ? x[row * EH_D + j] : 0.f;
        float q, k, v, A[EH_S], Ad[EH_S], a;
        const float r = eh_row_forward<false>(d, p, L, da, dd, site, on, rg, j, rg0, seq, l, xj, q, k, v, A, Ad, a);
        const float mu = group_sum(r, EH_D) * (1.f / EH_D);
        const float c = r - mu;
        const float rs = 1.0f / sqrtf(group_sum(c * c, EH_D) * (1.f / EH_D) + d.eps);
        if (on) {
            const float hh = c * rs;
            xhat[row * EH_D + j] = hh;
            x1[row * EH_D + j] = fmaf(hh, gm, bt);
            if (j == 0) rstd[row] = rs;
        }
        __syncthreads();        // the row images are rewritten by the next pass
    }
}

__global__ __launch_bounds__(256) void enc_head32_bwd_kernel(EhDims d, EhP p, const float* __restrict__ x, const float* __restrict__ d1,
                                                              const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                              float* __restrict__ dx, float* __restrict__ slabs, DropCfg dca, DropCfg dcd,
                                                              uint64_t site) {
    __shared__ __attribute__((aligned(16))) float lds[EH_LDS_FLOATS];
    __shared__ float G[4 * EH_D * EH_P + 6 * EH_D];                 // the workgroup's parameter gradients
    const EhLds L = eh_lds(lds);
    eh_stage_weights(p, L);
    const DropE da = drop_e(dca), dd = drop_e(dcd);
    const int S = d.S, E = d.E;
    const int rg = threadIdx.x >> 5, j = threadIdx.x & 31, sl = rg / S, l = rg - sl * S, rg0 = sl < d.spb ? sl * S : 0, h = j / E;
    const float gm = p.ln_w[j];
    float gWo[EH_D], gWq[EH_D], gWk[EH_D], gWv[EH_D];        // rows j of dW_o and of the three blocks of dW_in
#pragma unroll
    for (int c = 0; c < EH_D; ++c) gWo[c] = gWq[c] = gWk[c] = gWv[c] = 0.f;
    float gbq = 0.f, gbk = 0.f, gbv = 0.f, gbo = 0.f, glw = 0.f, glb = 0.f;
    __syncthreads();
    const long npass = ((long)d.Bs + d.spb - 1) / d.spb;
    for (long ps = blockIdx.x; ps < npass; ps += gridDim.x) {
        const long seq = ps * d.spb + sl;
        const bool on = sl < d.spb && seq < d.Bs;
        const long row = seq * S + l;
        const float xj = on ? x[row * EH_D + j] : 0.f;
        const float g1 = on ? d1[row * EH_D + j] : 0.f, xh = on ? xhat[row * EH_D + j] : 0.f, rs = on ? rstd[row] : 0.f;
        float q, k, v, A[EH_S], Ad[EH_S], a;
        (void)eh_row_forward<true>(d, p, L, da, dd, site, on, rg, j, rg0, seq, l, xj, q, k, v, A, Ad, a);
        // LayerNorm1 backward: dres = gradient of r = x + dropout1(sa)
        glw = fmaf(g1, xh, glw); glb += g1;
        const float g = g1 * gm;
        const float m1 = group_sum(g, EH_D) * (1.f / EH_D), m2 = group_sum(g * xh, EH_D) * (1.f / EH_D);
        const float dres = rs * (g - m1 - xh * m2);
        float dxj = dres;
        const float dsa = on ? dres * drop1(dd, site + 1, (uint64_t)row * EH_D + j) : 0.f;
        gbo += dsa;
        L.t0[rg * EH_D + j] = dsa;
        __syncthreads();
        float daj = 0.f;       // da[c = j] = sum_jj dsa[jj] W_o[jj][j]
#pragma unroll 8
        for (int c = 0; c < EH_D; ++c) {
            gWo[c] = fmaf(dsa, L.as[rg * EH_D + c], gWo[c]);
            daj = fmaf(L.t0[rg * EH_D + c], L.Wo[c * EH_P + j], daj);
        }
        // attention backward of row l (every lane of a head group holds the same dA / dS)
        float dS[EH_S], dot = 0.f;
#pragma unroll
        for (int s = 0; s < EH_S; ++s) {
            dS[s] = 0.f;
            if (s < S) {
                const float dA = group_sum(daj * L.vs[(rg0 + s) * EH_D + j], E);
                dS[s] = A[s] > 0.f ? dA * (Ad[s] / A[s]) : 0.f;          // x the dropout scale of (l, s)
                dot = fmaf(A[s], dS[s], dot);
            }
        }
        float dq = 0.f;
#pragma unroll
        for (int s = 0; s < EH_S; ++s)
            if (s < S) {
                dS[s] = A[s] * (dS[s] - dot);
                dq = fmaf(dS[s], L.ks[(rg0 + s) * EH_D + j], dq);
                if ((j & (E - 1)) == 0) { L.Ds[(rg * 4 + (h & 3)) * EH_S + s] = dS[s]; L.As[(rg * 4 + (h & 3)) * EH_S + s] = Ad[s]; }
            }
        dq *= d.scale;
        __syncthreads();        // t0 (dsa) has been read by every lane of the row; dS / Ad of the sequence's rows are in LDS
        L.t0[rg * EH_D + j] = daj;
        __syncthreads();
        float dk = 0.f, dv = 0.f;      // this row as key / value: contributions of the sequence's query rows
#pragma unroll
        for (int s = 0; s < EH_S; ++s)
            if (s < S) {
                dk = fmaf(L.Ds[((rg0 + s) * 4 + (h & 3)) * EH_S + l], L.qs[(rg0 + s) * EH_D + j], dk);
                dv = fmaf(L.As[((rg0 + s) * 4 + (h & 3)) * EH_S + l], L.t0[(rg0 + s) * EH_D + j], dv);
            }
        dk *= d.scale;
        if (!on) { dq = 0.f; dk = 0.f; dv = 0.f; }
        gbq += dq; gbk += dk; gbv += dv;
        L.t1[rg * 3 * EH_D + j] = dq;
        L.t1[rg * 3 * EH_D + EH_D + j] = dk;
        L.t1[rg * 3 * EH_D + 2 * EH_D + j] = dv;
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < EH_D; ++c) {
            const float xc = L.xs[rg * EH_D + c];
            gWq[c] = fmaf(dq, xc, gWq[c]); gWk[c] = fmaf(dk, xc, gWk[c]); gWv[c] = fmaf(dv, xc, gWv[c]);
        }
#pragma unroll 8
        for (int c = 0; c < 3 * EH_D; ++c) dxj = fmaf(L.t1[rg * 3 * EH_D + c], L.Win[c * EH_P + j], dxj);
        if (on) dx[row * EH_D + j] = dxj;
        __syncthreads();        // the row images are rewritten by the next pass
    }
    // the eight row groups' partial gradients: the two groups of a wave meet by a lane swap, the four waves add into an LDS image one
    // after the other (pitch 33: a first version added all eight groups with LDS atomics at [row j][column c] -- every lane of an
    // instruction on ONE bank, 8 waves on one address: 50 of the kernel's 60 us), the workgroup's sums leave as one slab
    const int wave = threadIdx.x >> 6;
    float gb[6] = {gbq, gbk, gbv, gbo, glw, glb};
#pragma unroll
    for (int c = 0; c < EH_D; ++c) { gWq[c] = xor32_sum(gWq[c]); gWk[c] = xor32_sum(gWk[c]); gWv[c] = xor32_sum(gWv[c]); gWo[c] = xor32_sum(gWo[c]); }
#pragma unroll
    for (int i = 0; i < 6; ++i) gb[i] = xor32_sum(gb[i]);
    float* Gw = G;                              // [4][32 columns c][33]: (block, c, j)
    float* Gb = G + 4 * EH_D * EH_P;            // [6][32]
    for (int w = 0; w < 4; ++w) {
        if (wave == w && (threadIdx.x & 63) < 32) {
#pragma unroll
            for (int c = 0; c < EH_D; ++c) {
                float* g0 = Gw + c * EH_P + j;
                if (w == 0) { g0[0] = gWq[c]; g0[EH_D * EH_P] = gWk[c]; g0[2 * EH_D * EH_P] = gWv[c]; g0[3 * EH_D * EH_P] = gWo[c]; }
                else { g0[0] += gWq[c]; g0[EH_D * EH_P] += gWk[c]; g0[2 * EH_D * EH_P] += gWv[c]; g0[3 * EH_D * EH_P] += gWo[c]; }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (w == 0) Gb[i * EH_D + j] = gb[i];
                else Gb[i * EH_D + j] += gb[i];
            }
        }
        __syncthreads();
    }
    float* slab = slabs + (size_t)blockIdx.x * EH_NG;
    for (int i = threadIdx.x; i < 4 * EH_D * EH_D; i += 256) {          // slab order: W_in (96 x 32), W_o (32 x 32), row-major
        const int blk = i >> 10, jj = (i >> 5) & 31, c = i & 31;
        slab[i] = Gw[blk * EH_D * EH_P + c * EH_P + jj];
    }
    for (int i = threadIdx.x; i < 6 * EH_D; i += 256) slab[4 * EH_D * EH_D + i] = Gb[i];
}

// out = sum over the slabs, scattered to the six parameter-gradient buffers (written)
__global__ __launch_bounds__(256) void enc_head32_reduce_kernel(const float* __restrict__ slabs, int nslab, EhP g) {
    const int t = blockIdx.x * 256 + threadIdx.x, i = t >> 3, part = t & 7;          // eight lanes per output split the slabs
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < EH_NG) {
        int s = part;
        for (; s + 24 < nslab; s += 32) {
            a0 += slabs[(size_t)s * EH_NG + i]; a1 += slabs[(size_t)(s + 8) * EH_NG + i];
            a2 += slabs[(size_t)(s + 16) * EH_NG + i]; a3 += slabs[(size_t)(s + 24) * EH_NG + i];
        }
        for (; s < nslab; s += 8) a0 += slabs[(size_t)s * EH_NG + i];
    }
    const float v = group_sum((a0 + a1) + (a2 + a3), 8);
    if (i >= EH_NG || part != 0) return;
    constexpr int nWi = 3 * EH_D * EH_D, nWo = EH_D * EH_D;
    if (i < nWi) const_cast<float*>(g.in_w)[i] = v;
    else if (i < nWi + nWo) const_cast<float*>(g.out_w)[i - nWi] = v;
    else {
        const int b = i - nWi - nWo;
        if (b < 3 * EH_D) const_cast<float*>(g.in_b)[b] = v;
        else if (b < 4 * EH_D) const_cast<float*>(g.out_b)[b - 3 * EH_D] = v;
        else if (b < 5 * EH_D) const_cast<float*>(g.ln_w)[b - 4 * EH_D] = v;
        else const_cast<float*>(g.ln_b)[b - 5 * EH_D] = v;
    }
}

inline EhDims eh_dims(int Bs, int S, int H, float eps) {
    EhDims d;
    d.Bs = Bs; d.S = S; d.H = H; d.E = EH_D / H; d.spb = 8 / S; d.scale = 1.0f / sqrtf((float)(EH_D / H)); d.eps = eps;
    return d;
}
inline int eh_grid(const EhDims& d) {
    const long npass = ((long)d.Bs + d.spb - 1) / d.spb;
    return (int)(npass < 512 ? npass : 512);
}

}  // namespace

bool enc_head32_ok(int Bs, int S, int D, int H) {
    constexpr bool on = true;
    return on && D == EH_D && S >= 1 && S <= EH_S && H >= 1 && H <= 4 && (EH_D % H) == 0 && Bs >= 1;
}
size_t enc_head32_slab_floats(int Bs, int S, int H) { return (size_t)eh_grid(eh_dims(Bs, S, H, 0.f)) * EH_NG; }

int launch_enc_head32_fwd(const float* x, int Bs, int S, int H, const float* in_w, const float* in_b, const float* out_w, const float* out_b,
                          const float* ln_w, const float* ln_b, float eps, DropCfg da, DropCfg dd, uint64_t site, float* x1, float* xhat,
                          float* rstd, hipStream_t s) {
    const EhDims d = eh_dims(Bs, S, H, eps);
    const EhP p{in_w, in_b, out_w, out_b, ln_w, ln_b};
    hipLaunchKernelGGL(enc_head32_fwd_kernel, dim3(eh_grid(d)), dim3(256), 0, s, d, p, x, x1, xhat, rstd, da, dd, site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_enc_head32_bwd(const float* x, const float* d1, const float* xhat, const float* rstd, int Bs, int S, int H, const float* in_w,
                          const float* in_b, const float* out_w, const float* out_b, const float* ln_w, float eps, DropCfg da, DropCfg dd,
                          uint64_t site, float* dx, float* g_in_w, float* g_in_b, float* g_out_w, float* g_out_b, float* g_ln_w, float* g_ln_b,
                          float* slabs, hipStream_t s) {
    const EhDims d = eh_dims(Bs, S, H, eps);
    const EhP p{in_w, in_b, out_w, out_b, ln_w, nullptr};
    const int grid = eh_grid(d);
    hipLaunchKernelGGL(enc_head32_bwd_kernel, dim3(grid), dim3(256), 0, s, d, p, x, d1, xhat, rstd, dx, slabs, da, dd, site);
    IMMTSF_LAUNCH_CHECK();
    const EhP g{g_in_w, g_in_b, g_out_w, g_out_b, g_ln_w, g_ln_b};
    hipLaunchKernelGGL(enc_head32_reduce_kernel, dim3((EH_NG * 8 + 255) / 256), dim3(256), 0, s, slabs, grid, g);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
