// TTF_T2V_XAttn in FOLDED form (reference: fusions/TTF_T2V_XAttn.py:120-182).
//
// The block's query is ONE learned vector for every (window, forecast step) (:91, :143 -- t_hat only gives the shape), and nothing
// between input_proj and out_proj is nonlinear except the softmax over a window's notes.  With x_n = [V_n ; Time2Vec(tau_n)] the raw
// (d_m + d/2)-vector of a note:
//
//   score  s[n, h] = u_h . x_n + const        u_h = D^T W_KV^T W_k,h^T (scale q_h)      (the constant drops out of the softmax)
//   value  z[n, h] = W_tot,h x_n + c_h        W_tot,h = W_o[:, h] W_v[h, :] W_KV D      D = blockdiag(W_in, I)
//   E_attn[b, t]   = sum_h sum_n a~[b, t, h, n] z[n, h] + b_o                           a~ = dropout(softmax_n(s))
//
// so the ΣN-row GEMM chain input_proj -> KV_proj -> in-projection (k | v) and the (B T)-row out_proj collapse into ONE ΣN x (H d) x
// (d_m + d/2) product; W_tot, u, c are parameter-only products (d x d x d class), and the original parameters' gradients follow from
// dW_tot, du, dc by the chain rule through the same factors.  This file holds the kernels that are not GEMMs: the multi-job vector
// and rank-1 launches of the parameter chains, the scores, and the softmax + dropout + mix in both directions.
#include "t2v_fold.hpp"

namespace {

constexpr int TT = 32;        // forecast steps per window (limit of the folded form)

// ------------------------------------------------------------------------------------------------ multi-job vector kernel
struct VecJobsK { VecJob j[VJ_MAX]; int wg0[VJ_MAX + 1]; int n; };

__device__ __forceinline__ void vj_store(const VecJob& J, size_t idx, float v) {
    if (J.y) J.y[idx] = v;
    if (J.yh) static_cast<bf16_t*>(J.yh)[idx] = (bf16_t)v;
}

__global__ __launch_bounds__(256) void vecjobs_kernel(VecJobsK L) {
    __shared__ float red[4][64];
    int ji = 0;
#pragma unroll 1
    while (ji + 1 < L.n && (int)blockIdx.x >= L.wg0[ji + 1]) ++ji;
    const VecJob& J = L.j[ji];
    const int bid = blockIdx.x - L.wg0[ji];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (J.type == VJ_MV) {
        const int row = bid * 4 + wave;
        if (row >= J.rows) return;
        const float* x = J.x + (J.xdiv > 0 ? (size_t)(row / J.xdiv) * J.xld : 0);
        const float* w = J.W + (size_t)row * J.ld;
        float a = 0.f;
        if (((J.ld | J.cols) & 3) == 0 && ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(x)) & 15) == 0) {
#pragma unroll 4
            for (int j = lane * 4; j < J.cols; j += 256) {
                const float4 wv = *reinterpret_cast<const float4*>(w + j), xv = *reinterpret_cast<const float4*>(x + j);
                a = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, a))));
            }
        } else {
            for (int j = lane; j < J.cols; j += 64) a = fmaf(w[j], x[j], a);
        }
        a = wave_sum(a);
        if (lane == 0) vj_store(J, row, J.scale * (a + (J.b ? J.b[row] : 0.f)));
        return;
    }
    if (J.type == VJ_MVT) {
        const int j = bid * 64 + lane;
        float a = 0.f;
        if (j < J.cols) {
#pragma unroll 8
            for (int i = wave; i < J.rows; i += 4) a = fmaf(J.W[(size_t)i * J.ld + j], J.x ? J.x[i] : 1.f, a);
        }
        red[wave][lane] = a;
        __syncthreads();
        if (wave == 0 && j < J.cols) {
            const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
            vj_store(J, j, (J.acc ? J.y[j] : 0.f) + J.scale * (t + (J.b ? J.b[j] : 0.f)));
        }
        return;
    }
    const long nwg = L.wg0[ji + 1] - L.wg0[ji];
    if (J.type == VJ_COPY) {
        const long n = (long)J.rows * J.cols;
        for (long x = (long)bid * 256 + threadIdx.x; x < n; x += nwg * 256) {
            const int i = (int)(x / J.cols), j = (int)(x - (long)i * J.cols);
            vj_store(J, (size_t)i * J.ldy + j, J.W ? J.scale * J.W[(size_t)i * J.ld + j] : 0.f);
        }
        return;
    }
    // VJ_RANK1: y[i ldy + j] = base + scale x[i] b[j]
    const int c4 = J.cols >> 2;
    const bool vec = (J.cols & 3) == 0 && (J.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(J.y) & 15) == 0 &&
                     (!J.b || (reinterpret_cast<uintptr_t>(J.b) & 15) == 0) &&
                     (J.acc != 2 || ((J.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(J.W) & 15) == 0));
    if (vec) {
        const long n = (long)J.rows * c4;
        for (long x = (long)bid * 256 + threadIdx.x; x < n; x += nwg * 256) {
            const int i = (int)(x / c4), j = (int)(x - (long)i * c4) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            float4* o = reinterpret_cast<float4*>(J.y + (size_t)i * J.ldy + j);
            if (J.acc == 1) v = *o;
            else if (J.acc == 2) v = *reinterpret_cast<const float4*>(J.W + (size_t)i * J.ld + j);
            if (J.x) {
                const float ai = J.scale * J.x[i];
                const float4 bv = *reinterpret_cast<const float4*>(J.b + j);
                v.x = fmaf(ai, bv.x, v.x); v.y = fmaf(ai, bv.y, v.y); v.z = fmaf(ai, bv.z, v.z); v.w = fmaf(ai, bv.w, v.w);
            }
            *o = v;
        }
        return;
    }
    const long n = (long)J.rows * J.cols;
    for (long x = (long)bid * 256 + threadIdx.x; x < n; x += nwg * 256) {
        const int i = (int)(x / J.cols), j = (int)(x - (long)i * J.cols);
        float v = J.acc == 1 ? J.y[(size_t)i * J.ldy + j] : J.acc == 2 ? J.W[(size_t)i * J.ld + j] : 0.f;
        if (J.x) v = fmaf(J.scale * J.x[i], J.b[j], v);
        J.y[(size_t)i * J.ldy + j] = v;
    }
}

// ------------------------------------------------------------------------------------------------ scores
__device__ __forceinline__ void ld8(const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t* p, float (&o)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const unsigned int w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(w[j] << 16); o[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
}
// one wave per packed row, every head: S[r, h] = X[r, :] . U[h, :]  (dmc a multiple of 8)
template <typename KT, int HMAX>
__global__ __launch_bounds__(256) void t2v_scores_kernel(const KT* __restrict__ X, int dmc, const float* __restrict__ U, int ldu, int H,
                                                          const int* __restrict__ total, int max_rows, float* __restrict__ S) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int rows = total ? min(*total, max_rows) : max_rows;
    if (row >= rows) return;
    const KT* xr = X + (size_t)row * dmc;
    float a[HMAX];
#pragma unroll
    for (int h = 0; h < HMAX; ++h) a[h] = 0.f;
#pragma unroll 2
    for (int c = lane * 8; c < dmc; c += 512) {
        float xv[8];
        ld8(xr + c, xv);
#pragma unroll
        for (int h = 0; h < HMAX; ++h) {
            if (h < H) {
                float uv[8];
                ld8(U + (size_t)h * ldu + c, uv);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[h] = fmaf(xv[j], uv[j], a[h]);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
        if (h < H) {
            const float t = wave_sum(a[h]);
            if (lane == 0) S[(size_t)row * H + h] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------ mix, forward
// grid (B, ceil(d / 256)), 256 threads: a thread owns one output column e of its window for all T steps.  Per head: the window's
// n <= NV scores -> softmax in every wave's own lanes, the (step, note) dropout scales as an LDS tile, the value column z[n, h d + e]
// of the window's notes in registers (every load issued before the first use), acc[t] += sum_n mt[t, n] p[n] z[n].
template <typename KT, int NV>
__global__ __launch_bounds__(256) void t2v_mix_fwd_kernel(T2VFoldDims dm, const int* __restrict__ offsets, const int* __restrict__ rowmap,
                                                           const float* __restrict__ S, const KT* __restrict__ z,
                                                           const float* __restrict__ b_o, const float* __restrict__ q_res,
                                                           float* __restrict__ P, float* __restrict__ xpre, DropCfg drop, uint64_t site) {
    __shared__ __attribute__((aligned(16))) float mt[TT * NV];       // dropout scale of (forecast step, note); 0 past T / n
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = dm.d, H = dm.H, Hd = H * d, T = dm.T;
    const int ob = offsets[b], n = offsets[b + 1] - ob;
    const int e = blockIdx.y * 256 + tid;
    const bool valid = e < d;
    const float qr = valid ? q_res[e] : 0.f;
    if (n == 0) {        // no notes: E_attn is zero (M_txt), the residual query remains
        if (valid) for (int t = 0; t < T; ++t) xpre[(size_t)(b * T + t) * d + e] = qr;
        return;
    }
    const uint64_t seed = drop.seed + ((drop.p > 0.f && drop.seed_dev) ? *drop.seed_dev : 0ull);
    const size_t obase = (size_t)b * T * d + (valid ? e : 0);
    const float add = valid ? b_o[e] + qr : 0.f;
    for (int h = 0; h < H; ++h) {
        // this thread's value column of head h (rows past n: the last row again, weight 0)
        KT vr[NV];
        {
            const KT* vbase = z + (size_t)ob * Hd + (size_t)h * d + (valid ? e : 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) vr[i] = vbase[(size_t)(i < n ? i : n - 1) * Hd];
        }
        const float s_l = lane < n ? S[(size_t)(ob + lane) * H + h] : -INFINITY;
        if (h > 0) __syncthreads();            // the previous head's tile has been read
        for (int x = tid; x < TT * NV; x += 256) {
            const int tt = x / NV, ii = x - tt * NV;
            float a = 0.f;
            if (tt < T && ii < n) {
                a = 1.f;
                if (drop.p > 0.f) {
                    const int n_orig = rowmap[ob + ii] - b * dm.N;
                    const uint64_t idx = ((uint64_t)(b * T + tt) * H + h) * dm.N + n_orig;
                    a = dropout_scale(seed, site, idx, drop.p, drop.inv_keep);
                }
            }
            mt[x] = a;
        }
        const float m = wave_max(s_l);
        float p = lane < n ? expf(s_l - m) : 0.f;
        p *= 1.f / wave_sum(p);
        if (wave == 0 && blockIdx.y == 0 && lane < n) P[(size_t)(ob + lane) * H + h] = p;
        __syncthreads();
        float pv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) pv[i] = lane_bcast(p, i) * (float)vr[i];        // (weight 0 past n)
        // one output row per step: head 0 starts from b_o + the residual query, the others add to what is there (same thread, same
        // element: no race).  Rolled over the steps on purpose -- fully unrolled the 32 x NV tile reads took every register (259 VGPRs,
        // one wave per SIMD: 0.81 ms at 4096 windows)
#pragma unroll 2
        for (int tt = 0; tt < T; ++tt) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < NV; i += 4) {
                const float4 m4 = *reinterpret_cast<const float4*>(mt + tt * NV + i);
                a = fmaf(m4.x, pv[i], a); a = fmaf(m4.y, pv[i + 1], a); a = fmaf(m4.z, pv[i + 2], a); a = fmaf(m4.w, pv[i + 3], a);
            }
            if (valid) {
                float* o = xpre + obase + (size_t)tt * d;
                *o = a + (h == 0 ? add : *o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ mix, backward
// grid (B), 256 threads: the workgroup owns its window -- every column, every head -- so the softmax backward needs no second launch:
//   g[n, h, e]  = sum_t mt[t, h, n] dx[b, t, e]          dz[n, h d + e] = p[n, h] g        dp[n, h] = sum_e g z[n, h d + e]
//   ds[n, h]    = p[n, h] (dp[n, h] - sum_j p[j, h] dp[j, h])                                  -> column H d + h of dz_aug
// The window's dropout tiles (all heads) are built once, note-major, so a note's T scales are eight 16-byte broadcast reads; a
// thread walks the columns e = tid, tid + 256, ... with the T upstream values of its column in registers; the per-note dot products
// meet in wave-private LDS slabs (plain adds by lane 0, summed in wave order).
// DT: the type dx is stored in (float, or bf16 when the LayerNorm backward in front wrote its compact image only)
template <typename KT, int NV, int HMAX, typename DT>
__global__ __launch_bounds__(256) void t2v_mix_bwd_kernel(T2VFoldDims dm, const int* __restrict__ offsets, const int* __restrict__ rowmap,
                                                           const float* __restrict__ P, const KT* __restrict__ z,
                                                           const DT* __restrict__ dx, KT* __restrict__ dz, float* __restrict__ dbo_part,
                                                           DropCfg drop, uint64_t site) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = dm.H;
    float* mt = lds;                                   // [H][NV][TT]
    float* pl = mt + (size_t)H * NV * TT;              // [H][NV]
    float* dpw = pl + H * NV;                          // [4 waves][H][NV]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = dm.d, Hd = H * d, Ma = Hd + 8, T = dm.T;
    const int ob = offsets[b], n = offsets[b + 1] - ob;
    if (n == 0) {
        for (int e = tid; e < d; e += 256) dbo_part[(size_t)b * d + e] = 0.f;
        return;
    }
    const uint64_t seed = drop.seed + ((drop.p > 0.f && drop.seed_dev) ? *drop.seed_dev : 0ull);
    for (int x = tid; x < H * NV * TT; x += 256) {
        const int h = x / (NV * TT), r = x - h * NV * TT, ii = r / TT, tt = r - ii * TT;
        float a = 0.f;
        if (tt < T && ii < n) {
            a = 1.f;
            if (drop.p > 0.f) {
                const int n_orig = rowmap[ob + ii] - b * dm.N;
                const uint64_t idx = ((uint64_t)(b * T + tt) * H + h) * dm.N + n_orig;
                a = dropout_scale(seed, site, idx, drop.p, drop.inv_keep);
            }
        }
        mt[x] = a;
    }
    for (int x = tid; x < H * NV; x += 256) {
        const int h = x / NV, ii = x - h * NV;
        pl[x] = ii < n ? P[(size_t)(ob + ii) * H + h] : 0.f;
    }
    for (int x = tid; x < 4 * H * NV; x += 256) dpw[x] = 0.f;
    __syncthreads();
    for (int e = tid; e < ((d + 255) & ~255); e += 256) {
        const bool valid = e < d;
        float dcv[TT];
        float gsum = 0.f;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            dcv[t] = (valid && t < T) ? (float)dx[(size_t)(b * T + t) * d + e] : 0.f;
            gsum += dcv[t];
        }
        if (valid) dbo_part[(size_t)b * d + e] = gsum;
        for (int h = 0; h < H; ++h) {
            KT vr[NV];
            {
                const KT* vbase = z + (size_t)ob * Hd + (size_t)h * d + (valid ? e : 0);
#pragma unroll
                for (int i = 0; i < NV; ++i) vr[i] = vbase[(size_t)(i < n ? i : n - 1) * Hd];
            }
#pragma unroll 4
            for (int i = 0; i < NV; ++i) {
                if (i >= n) break;              // (n is workgroup-uniform)
                const float4* m4 = reinterpret_cast<const float4*>(mt + ((size_t)h * NV + i) * TT);
                float g = 0.f;
#pragma unroll
                for (int t4 = 0; t4 < TT / 4; ++t4) {
                    const float4 m = m4[t4];
                    g = fmaf(m.x, dcv[4 * t4], fmaf(m.y, dcv[4 * t4 + 1], fmaf(m.z, dcv[4 * t4 + 2], fmaf(m.w, dcv[4 * t4 + 3], g))));
                }
                float a = valid ? g * (float)vr[i] : 0.f;
                if (valid) dz[(size_t)(ob + i) * Ma + (size_t)h * d + e] = (KT)(pl[h * NV + i] * g);
                a = wave_sum(a);
                if (lane == 0) dpw[(wave * H + h) * NV + i] += a;
            }
        }
    }
    __syncthreads();
    // softmax backward per head, notes in the lanes of wave h (h < H <= 4); the pad columns of dz_aug are zeroed by the same lanes
    if (wave < H) {            // (wave-uniform: the wave reductions below run with every lane active)
        const int h = wave;
        float dp = 0.f, p = 0.f;
        if (lane < n) {
            dp = dpw[(0 * H + h) * NV + lane] + dpw[(1 * H + h) * NV + lane] + dpw[(2 * H + h) * NV + lane] + dpw[(3 * H + h) * NV + lane];
            p = pl[h * NV + lane];
        }
        const float dot = wave_sum(p * dp);
        if (lane < n) {
            KT* row = dz + (size_t)(ob + lane) * Ma + Hd;
            row[h] = (KT)(p * (dp - dot));
            if (h == 0)
                for (int j = H; j < 8; ++j) row[j] = (KT)0.f;
        }
    }
}


// ------------------------------------------------------------------------------------------------ mix + LayerNorm, forward (wide form)
// grid (B): the workgroup owns its window's T rows over ALL d <= 1024 columns -- a thread owns FOUR adjacent columns (8- / 16-byte loads
// of the value rows, 8- / 16-byte stores of the outputs) for 16 steps, 4 x 16 accumulators in registers -- so the
// LayerNorm that follows the mix (reference: fusions/TTF_T2V_XAttn.py:176-179, ln(E_attn + Q) then dropout) runs on the accumulators:
// no fp32 x_pre in HBM (403 MB written + read at 4096 windows), x_hat and Z leave once.  The dropout tile is note-major so that a note's T
// scales are eight 16-byte broadcast reads shared by the thread's four columns.
struct MixLn {
    const float *gamma, *beta;
    float eps;
    float* xhat_f;        // x_hat as fp32, or ...
    bf16_t* xhat_h;       // ... as bf16 alone (the compact form of launch_layernorm_fwd)
    float* rstd;
    float* z_f;           // Z = dropout(LayerNorm(.)) fp32 (may be null when z_h is all the consumer reads)
    bf16_t* z_h;          // bf16 image (may be null)
    unsigned long long* keep;   // optional: the output dropout's keep bits, one 64-bit word per (window, half of the steps, column group):
                                // four bits per step -- what the LayerNorm backward would otherwise regenerate with a Philox call per
                                // four elements (launch_layernorm_bwd_lr's `keep`); written only when the dropout is on
};
__device__ __forceinline__ void ld4w(const float* p, float (&o)[4]) { const float4 a = *reinterpret_cast<const float4*>(p); o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; }
__device__ __forceinline__ void ld4w(const bf16_t* p, float (&o)[4]) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u); o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
}
__device__ __forceinline__ void st4w(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void st4w(bf16_t* p, const float (&v)[4]) {
    const bf16x4 h = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = h;
}
// Threads: G = ceil(d / 4) column groups (rounded up to whole waves) x NH halves of the T steps -- a thread owns four adjacent columns
// for TH = 16 consecutive steps (4 x 16 accumulators: ~130 VGPRs, three waves per SIMD; with all 32 steps in one thread the kernel sat
// at 200 VGPRs, two waves per SIMD, and at d = 768 a quarter of its 256 threads were idle: 379 us at 4096 windows).  A half is a set of
// whole waves, so the dropout tile's rows stay broadcast reads.
constexpr int TH = 16;
template <typename KT, int NV>
__global__ __launch_bounds__(512) void t2v_mix_ln_fwd_kernel(T2VFoldDims dm, const int* __restrict__ offsets, const int* __restrict__ rowmap,
                                                              const float* __restrict__ S, const KT* __restrict__ z,
                                                              const float* __restrict__ b_o, const float* __restrict__ q_res,
                                                              float* __restrict__ P, MixLn o, DropCfg drop, uint64_t site, DropCfg odrop,
                                                              uint64_t osite, int G) {
    __shared__ __attribute__((aligned(16))) float mt[NV * TT];       // [note][step] dropout scale; 0 past n / T
    __shared__ float pl[NV];
    __shared__ float red[8][TH];
    __shared__ float mu_s[TT], rs_s[TT];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int d = dm.d, H = dm.H, Hd = H * d, T = dm.T;
    const int ob = offsets[b], n = offsets[b + 1] - ob;
    const int half = tid / G, grp = tid - half * G, t0 = half * TH;       // (G is a multiple of 64: `half` is wave-uniform)
    const int wpg = G >> 6;                                                // waves per half
    const int e0 = grp * 4;
    const bool valid = e0 < d;
    float acc[4][TH];
    {
        float a0[4] = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            float qv[4], bv[4] = {0.f, 0.f, 0.f, 0.f};
            ld4w(q_res + e0, qv);
            if (n > 0) ld4w(b_o + e0, bv);          // (no notes: E_attn is zero, the residual query remains)
#pragma unroll
            for (int c = 0; c < 4; ++c) a0[c] = qv[c] + bv[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < TH; ++t) acc[c][t] = a0[c];
    }
    if (n > 0) {
        const uint64_t seed = drop.seed + ((drop.p > 0.f && drop.seed_dev) ? *drop.seed_dev : 0ull);
        for (int h = 0; h < H; ++h) {
            const float s_l = lane < n ? S[(size_t)(ob + lane) * H + h] : -INFINITY;
            const float m = wave_max(s_l);
            float p = lane < n ? expf(s_l - m) : 0.f;
            p *= 1.f / wave_sum(p);
            if (h > 0) __syncthreads();            // the previous head's tiles have been read
            for (int x = tid; x < NV * TT; x += nthr) {
                const int ii = x / TT, tt = x - ii * TT;
                float a = 0.f;
                if (tt < T && ii < n) {
                    a = 1.f;
                    if (drop.p > 0.f) {
                        const int n_orig = rowmap[ob + ii] - b * dm.N;
                        const uint64_t idx = ((uint64_t)(b * T + tt) * H + h) * dm.N + n_orig;
                        a = dropout_scale(seed, site, idx, drop.p, drop.inv_keep);
                    }
                }
                mt[x] = a;
            }
            if (wave == 0) {
                if (lane < NV) pl[lane] = p;                               // (0 past n)
                if (lane < n) P[(size_t)(ob + lane) * H + h] = p;
            }
            __syncthreads();
            if (valid) {
                const KT* vb = z + (size_t)ob * Hd + (size_t)h * d + e0;
                for (int i0 = 0; i0 < n; i0 += 8) {
                    float zv[8][4];
#pragma unroll
                    for (int k = 0; k < 8; ++k) ld4w(vb + (size_t)(i0 + k < n ? i0 + k : n - 1) * Hd, zv[k]);        // (every load before the first use)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (i0 + k >= n) break;             // (workgroup-uniform)
                        const float pi = pl[i0 + k];
                        float m4[TH];
#pragma unroll
                        for (int t4 = 0; t4 < TH / 4; ++t4) {
                            const float4 q = *reinterpret_cast<const float4*>(mt + (i0 + k) * TT + t0 + 4 * t4);
                            m4[4 * t4] = q.x; m4[4 * t4 + 1] = q.y; m4[4 * t4 + 2] = q.z; m4[4 * t4 + 3] = q.w;
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float pv = pi * zv[k][c];
#pragma unroll
                            for (int t = 0; t < TH; ++t) acc[c][t] = fmaf(m4[t], pv, acc[c][t]);
                        }
                    }
                }
            }
        }
    }
    // ---- LayerNorm over the d columns of each row (two passes over the registers: mean, then the centred squares); a row's columns
    // live in the wpg waves of its half
#pragma unroll
    for (int t = 0; t < TH; ++t) {
        const float ps = wave_sum(valid ? (acc[0][t] + acc[1][t]) + (acc[2][t] + acc[3][t]) : 0.f);
        if (lane == 0) red[wave][t] = ps;
    }
    __syncthreads();
    if (tid < TT) {
        const int hh = tid / TH, tl = tid - hh * TH;
        float a = 0.f;
        if (hh * G < nthr) for (int w = 0; w < wpg; ++w) a += red[hh * wpg + w][tl];
        mu_s[tid] = a / (float)d;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TH; ++t) {
        const float mu = mu_s[t0 + t];
        float v = 0.f;
        if (valid) {
            const float a = acc[0][t] - mu, bb = acc[1][t] - mu, c = acc[2][t] - mu, e = acc[3][t] - mu;
            v = fmaf(a, a, fmaf(bb, bb, fmaf(c, c, e * e)));
        }
        v = wave_sum(v);
        if (lane == 0) red[wave][t] = v;
    }
    __syncthreads();
    if (tid < TT) {
        const int hh = tid / TH, tl = tid - hh * TH;
        float a = 0.f;
        if (hh * G < nthr) for (int w = 0; w < wpg; ++w) a += red[hh * wpg + w][tl];
        const float rs = 1.0f / sqrtf(a / (float)d + o.eps);
        rs_s[tid] = rs;
        if (tid < T && o.rstd) o.rstd[b * T + tid] = rs;
    }
    __syncthreads();
    if (!valid) return;
    float gm[4], bt[4];
    ld4w(o.gamma + e0, gm);
    ld4w(o.beta + e0, bt);
    // output dropout: the keep bits of the thread's 4 x TH elements first, in a ROLLED loop (one Philox call per step; inlined into the
    // store loop below the compiler gave up unrolling it and the accumulators went to scratch), four bits per step in one 64-bit word
    uint64_t kb = ~0ull;
    if (odrop.p > 0.f) {
        kb = 0ull;
#pragma unroll 1
        for (int t = 0; t < TH; ++t) {
            if (t0 + t >= T) break;
            float sc[4];
            dropout_scale4(odrop, osite, (uint64_t)((size_t)(b * T + t0 + t) * d + e0), sc);
            const uint64_t bits = (sc[0] != 0.f ? 1ull : 0ull) | (sc[1] != 0.f ? 2ull : 0ull) | (sc[2] != 0.f ? 4ull : 0ull) | (sc[3] != 0.f ? 8ull : 0ull);
            kb |= bits << (4 * t);
        }
        if (o.keep) o.keep[((size_t)b * 2 + half) * 256 + grp] = kb;
    }
    const float keep = odrop.p > 0.f ? odrop.inv_keep : 1.f;
#pragma unroll
    for (int t = 0; t < TH; ++t) {
        if (t0 + t < T) {              // (a guard, not a break: the loop must unroll for the accumulators to stay in registers)
            const float mu = mu_s[t0 + t], rs = rs_s[t0 + t];
            const size_t at = (size_t)(b * T + t0 + t) * d + e0;
            const uint32_t k4 = (uint32_t)(kb >> (4 * t)) & 15u;
            float hv[4], zo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) hv[c] = (acc[c][t] - mu) * rs;
            if (o.xhat_f) st4w(o.xhat_f + at, hv);
            if (o.xhat_h) st4w(o.xhat_h + at, hv);
#pragma unroll
            for (int c = 0; c < 4; ++c) zo[c] = ((k4 >> c) & 1u) ? fmaf(hv[c], gm[c], bt[c]) * keep : 0.f;
            if (o.z_f) st4w(o.z_f + at, zo);
            if (o.z_h) st4w(o.z_h + at, zo);
        }
    }
}

// ------------------------------------------------------------------------------------------------ mix, backward (wide form)
// the same ownership as the wide forward: a thread holds the T upstream values of FOUR adjacent columns (one 8-byte load per row of a
// bf16 dx -- the two-byte accesses of the narrow kernel made it slower on a bf16 dx than on the fp32 one), writes dz four columns at
// a time, and the per-note dot products meet in wave-private LDS slabs exactly as in the narrow kernel.
template <typename KT, int NV, typename DT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void t2v_mix_bwd_wide_kernel(T2VFoldDims dm, const int* __restrict__ offsets, const int* __restrict__ rowmap,
                                                                const float* __restrict__ P, const KT* __restrict__ z,
                                                                const DT* __restrict__ dx, KT* __restrict__ dz, float* __restrict__ dbo_part,
                                                                DropCfg drop, uint64_t site) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = dm.H;
    float* mt = lds;                                   // [H][NV][TT]
    float* pl = mt + (size_t)H * NV * TT;              // [H][NV]
    float* dpw = pl + H * NV;                          // [4 waves][H][NV]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = dm.d, Hd = H * d, Ma = Hd + 8, T = dm.T;
    const int ob = offsets[b], n = offsets[b + 1] - ob;
    const int e0 = tid * 4;
    const bool valid = e0 < d;
    if (n == 0) {
        if (valid) { const float zr[4] = {0.f, 0.f, 0.f, 0.f}; st4w(dbo_part + (size_t)b * d + e0, zr); }
        return;
    }
    // the upstream rows first: the longest latency of the kernel runs under the tile construction
    float dcv[4][TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (valid && t < T) ld4w(dx + (size_t)(b * T + t) * d + e0, v);
#pragma unroll
        for (int c = 0; c < 4; ++c) dcv[c][t] = v[c];
    }
    const uint64_t seed = drop.seed + ((drop.p > 0.f && drop.seed_dev) ? *drop.seed_dev : 0ull);
    for (int x = tid; x < H * NV * TT; x += 256) {
        const int h = x / (NV * TT), r = x - h * NV * TT, ii = r / TT, tt = r - ii * TT;
        float a = 0.f;
        if (tt < T && ii < n) {
            a = 1.f;
            if (drop.p > 0.f) {
                const int n_orig = rowmap[ob + ii] - b * dm.N;
                const uint64_t idx = ((uint64_t)(b * T + tt) * H + h) * dm.N + n_orig;
                a = dropout_scale(seed, site, idx, drop.p, drop.inv_keep);
            }
        }
        mt[x] = a;
    }
    for (int x = tid; x < H * NV; x += 256) {
        const int h = x / NV, ii = x - h * NV;
        pl[x] = ii < n ? P[(size_t)(ob + ii) * H + h] : 0.f;
    }
    for (int x = tid; x < 4 * H * NV; x += 256) dpw[x] = 0.f;
    __syncthreads();
    if (valid) {
        float gs[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float g = 0.f;
#pragma unroll
            for (int t = 0; t < TT; ++t) g += dcv[c][t];
            gs[c] = g;
        }
        st4w(dbo_part + (size_t)b * d + e0, gs);
    }
    for (int h = 0; h < H; ++h) {
        const KT* vb = z + (size_t)ob * Hd + (size_t)h * d + (valid ? e0 : 0);
        for (int i0 = 0; i0 < n; i0 += 4) {
            float zv[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ld4w(vb + (size_t)(i0 + k < n ? i0 + k : n - 1) * Hd, zv[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k;
                if (i >= n) break;              // (workgroup-uniform)
                float g4[4] = {0.f, 0.f, 0.f, 0.f};
                // the note's T scales in two halves of 16 (the registers the upstream values leave: three waves per SIMD)
#pragma unroll
                for (int hf2 = 0; hf2 < 2; ++hf2) {
                    float m4[16];
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        const float4 q = *reinterpret_cast<const float4*>(mt + ((size_t)h * NV + i) * TT + hf2 * 16 + 4 * t4);
                        m4[4 * t4] = q.x; m4[4 * t4 + 1] = q.y; m4[4 * t4 + 2] = q.z; m4[4 * t4 + 3] = q.w;
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int t = 0; t < 16; ++t) g4[c] = fmaf(m4[t], dcv[c][hf2 * 16 + t], g4[c]);
                }
                const float pi = pl[h * NV + i];
                float a = 0.f, out[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    a = fmaf(g4[c], zv[k][c], a);
                    out[c] = pi * g4[c];
                }
                if (valid) st4w(dz + (size_t)(ob + i) * Ma + (size_t)h * d + e0, out);
                a = wave_sum(valid ? a : 0.f);
                if (lane == 0) dpw[(wave * H + h) * NV + i] += a;
            }
        }
    }
    __syncthreads();
    if (wave < H) {            // (wave-uniform: the wave reductions below run with every lane active)
        const int h = wave;
        float dp = 0.f, p = 0.f;
        if (lane < n) {
            dp = dpw[(0 * H + h) * NV + lane] + dpw[(1 * H + h) * NV + lane] + dpw[(2 * H + h) * NV + lane] + dpw[(3 * H + h) * NV + lane];
            p = pl[h * NV + lane];
        }
        const float dot = wave_sum(p * dp);
        if (lane < n) {
            KT* row = dz + (size_t)(ob + lane) * Ma + Hd;
            row[h] = (KT)(p * (dp - dot));
            if (h == 0)
                for (int j = H; j < 8; ++j) row[j] = (KT)0.f;
        }
    }
}

}  // namespace

int t2v_mix_bwd_wide = 1;      // (tool switch: 0 = the narrow kernel)

int launch_vecjobs(const VecJobList& l, hipStream_t s) {
    if (l.n <= 0) return IMMTSF_OK;
    if (l.n > VJ_MAX) return IMMTSF_EINVAL;
    VecJobsK K;
    int wg = 0;
    for (int i = 0; i < l.n; ++i) {
        const VecJob& J = l.j[i];
        if (J.rows <= 0 || J.cols <= 0 || (!J.y && !J.yh)) return IMMTSF_EINVAL;
        K.j[i] = J;
        K.wg0[i] = wg;
        if (J.type == VJ_MV) wg += cdiv(J.rows, 4);
        else if (J.type == VJ_MVT) wg += cdiv(J.cols, 64);
        else if (J.type == VJ_COPY) {
            const long n = (long)J.rows * J.cols;
            wg += (int)((n + 2047) / 2048 > 128 ? 128 : (n + 2047) / 2048);
        } else {
            if (!J.y || (J.acc == 2 && !J.W) || (J.x && !J.b)) return IMMTSF_EINVAL;
            const long n = (long)J.rows * J.cols;
            wg += (int)((n + 4095) / 4096 > 192 ? 192 : (n + 4095) / 4096);
        }
    }
    for (int i = l.n; i <= VJ_MAX; ++i) K.wg0[i] = wg;
    K.n = l.n;
    hipLaunchKernelGGL(vecjobs_kernel, dim3(wg), dim3(256), 0, s, K);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

bool t2v_fold_shape_ok(int N, int T, int d, int H) { return N >= 1 && N <= 64 && T >= 1 && T <= TT && d >= 8 && d <= 1024 && (d % 8) == 0 && H >= 1 && H <= 4; }

int launch_t2v_scores(const void* X, int x_is_bf16, int dmc, const float* U, int ldu, int H, const int* total, int max_rows, float* S,
                      hipStream_t s) {
    if (max_rows <= 0) return IMMTSF_OK;
    if (H < 1 || H > 4 || (dmc % 8) || (ldu % 4) || (reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(U) & 15)) return IMMTSF_EUNSUPPORTED;
    const dim3 grid(cdiv(max_rows, 4));
    if (x_is_bf16) hipLaunchKernelGGL((t2v_scores_kernel<bf16_t, 4>), grid, dim3(256), 0, s, static_cast<const bf16_t*>(X), dmc, U, ldu, H, total, max_rows, S);
    else hipLaunchKernelGGL((t2v_scores_kernel<float, 4>), grid, dim3(256), 0, s, static_cast<const float*>(X), dmc, U, ldu, H, total, max_rows, S);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_t2v_mix_fwd(T2VFoldDims dm, const int* offsets, const int* rowmap, const float* S, const void* z, int z_is_bf16, const float* b_o,
                       const float* q_res, float* P, float* xpre, DropCfg drop, uint64_t site, hipStream_t s) {
    if (!t2v_fold_shape_ok(dm.N, dm.T, dm.d, dm.H)) return IMMTSF_EUNSUPPORTED;
    const dim3 grid(dm.B, cdiv(dm.d, 256));
#define MIXF(KT, NV) hipLaunchKernelGGL((t2v_mix_fwd_kernel<KT, NV>), grid, dim3(256), 0, s, dm, offsets, rowmap, S, static_cast<const KT*>(z), \
                                        b_o, q_res, P, xpre, drop, site)
    if (z_is_bf16) { if (dm.N <= 32) MIXF(bf16_t, 32); else MIXF(bf16_t, 64); }
    else { if (dm.N <= 32) MIXF(float, 32); else MIXF(float, 64); }
#undef MIXF
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

bool t2v_mix_wide_ok(int d) { return (d % 8) == 0 && d <= 1024; }      // (four adjacent columns per thread, 16-byte rows)

int launch_t2v_mix_ln_fwd(T2VFoldDims dm, const int* offsets, const int* rowmap, const float* S, const void* z, int z_is_bf16, const float* b_o,
                          const float* q_res, float* P, const float* gamma, const float* beta, float eps, float* xhat_f, void* xhat_h,
                          float* rstd, float* z_f, void* z_h, DropCfg drop, uint64_t site, DropCfg odrop, uint64_t osite, hipStream_t s,
                          void* keep) {
    if (!t2v_fold_shape_ok(dm.N, dm.T, dm.d, dm.H) || !t2v_mix_wide_ok(dm.d)) return IMMTSF_EUNSUPPORTED;
    if ((!xhat_f && !xhat_h) || (!z_f && !z_h)) return IMMTSF_EINVAL;
    MixLn o;
    o.gamma = gamma; o.beta = beta; o.eps = eps; o.xhat_f = xhat_f; o.xhat_h = static_cast<bf16_t*>(xhat_h); o.rstd = rstd; o.z_f = z_f;
    o.z_h = static_cast<bf16_t*>(z_h);
    o.keep = static_cast<unsigned long long*>(keep);
    const int G = cdiv(cdiv(dm.d, 4), 64) * 64, NH = dm.T > TH ? 2 : 1;      // column groups (whole waves) x halves of the steps
#define MIXL(KT, NV) hipLaunchKernelGGL((t2v_mix_ln_fwd_kernel<KT, NV>), dim3(dm.B), dim3(G * NH), 0, s, dm, offsets, rowmap, S, static_cast<const KT*>(z), \
                                        b_o, q_res, P, o, drop, site, odrop, osite, G)
    if (z_is_bf16) { if (dm.N <= 32) MIXL(bf16_t, 32); else MIXL(bf16_t, 64); }
    else { if (dm.N <= 32) MIXL(float, 32); else MIXL(float, 64); }
#undef MIXL
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_t2v_mix_bwd(T2VFoldDims dm, const int* offsets, const int* rowmap, const float* P, const void* z, int z_is_bf16, const void* dx,
                       int dx_is_bf16, void* dz_aug, float* dbo_part, DropCfg drop, uint64_t site, hipStream_t s) {
    if (!t2v_fold_shape_ok(dm.N, dm.T, dm.d, dm.H)) return IMMTSF_EUNSUPPORTED;
    if (dx_is_bf16 && !z_is_bf16) return IMMTSF_EINVAL;
    const int NV = dm.N <= 32 ? 32 : 64;
    const size_t lds = ((size_t)dm.H * NV * TT + (size_t)dm.H * NV + 4 * (size_t)dm.H * NV) * sizeof(float);
    if (t2v_mix_wide_ok(dm.d) && t2v_mix_bwd_wide) {
#define MIXW(KT, NVV, DT) hipLaunchKernelGGL((t2v_mix_bwd_wide_kernel<KT, NVV, DT>), dim3(dm.B), dim3(256), lds, s, dm, offsets, rowmap, P, \
                                             static_cast<const KT*>(z), static_cast<const DT*>(dx), static_cast<KT*>(dz_aug), dbo_part, drop, site)
        if (z_is_bf16 && dx_is_bf16) { if (NV == 32) MIXW(bf16_t, 32, bf16_t); else MIXW(bf16_t, 64, bf16_t); }
        else if (z_is_bf16) { if (NV == 32) MIXW(bf16_t, 32, float); else MIXW(bf16_t, 64, float); }
        else { if (NV == 32) MIXW(float, 32, float); else MIXW(float, 64, float); }
#undef MIXW
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
#define MIXB(KT, NVV, DT) hipLaunchKernelGGL((t2v_mix_bwd_kernel<KT, NVV, 4, DT>), dim3(dm.B), dim3(256), lds, s, dm, offsets, rowmap, P, \
                                             static_cast<const KT*>(z), static_cast<const DT*>(dx), static_cast<KT*>(dz_aug), dbo_part, drop, site)
    if (z_is_bf16 && dx_is_bf16) { if (NV == 32) MIXB(bf16_t, 32, bf16_t); else MIXB(bf16_t, 64, bf16_t); }
    else if (z_is_bf16) { if (NV == 32) MIXB(bf16_t, 32, float); else MIXB(bf16_t, 64, float); }
    else { if (NV == 32) MIXB(float, 32, float); else MIXB(float, 64, float); }
#undef MIXB
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
