// Ragged single-query cross-attention (TTF_T2V_XAttn) and row softmax (+attention-weight dropout) kernels.
//
// TTF_T2V_XAttn's query is one learned vector (fusions/TTF_T2V_XAttn.py:91,143), so per window b and head h the
// scores are a mat-vec of the window's packed keys with the scaled query, the softmax runs over that window's
// n_b notes only (offsets[b]..offsets[b+1]) -- no -inf padding, no T-fold copy of K/V (the reference materialises
// T copies, :150-159) -- and only the attention-weight dropout makes the T output rows differ.
#include "attn.hpp"
#include "rowops.hpp"

namespace {

constexpr int TT = 32;   // forecast steps accumulated per pass in registers

// Long windows (cfg5: up to 4096 notes, lengths U{1..4096}) make one workgroup per window the whole story: 64 x H x
// ceil(hd/256) workgroups, the longest one walking 4096 notes.  Above RAGGED_SPLIT_N padded notes the work is cut into
// chunks of RAGGED_CH notes instead (SPLIT = true): scores by one wave per packed row (ragged_scores_kernel), the softmax
// by one workgroup per (window, head) over the n scores (ragged_softmax_kernel), the weighted sum of the values by one
// workgroup per (window, chunk, head, 256 columns) into per-chunk partial sums (this kernel with SPLIT), summed by
// ragged_ctx_reduce_kernel.  Windows of one chunk write ctx directly.
constexpr int RAGGED_SPLIT_N = 256;
constexpr int RAGGED_CH = 256;

// K | V rows are fp32, or -- bf16 mode -- the bf16 image the in-projection GEMM writes (half the bytes of the step's largest tensor)
__device__ __forceinline__ float4 kv_load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 kv_load4(const bf16_t* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
// one wave per (packed row, head): S[row, h] = qs_h . k_row
template <typename KT>
__global__ __launch_bounds__(256) void ragged_scores_kernel(RaggedAttnDims dm, const int* __restrict__ offsets,
                                                             const KT* __restrict__ KVp, const float* __restrict__ qs,
                                                             float* __restrict__ S) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y, lane = threadIdx.x & 63;
    if (row >= offsets[dm.B]) return;
    const int hd = dm.hd, ld = 2 * dm.H * hd;
    const KT* kr = KVp + (size_t)row * ld + h * hd;
    const float* q = qs + h * hd;
    float a = 0.f;
    if ((hd & 3) == 0) {
        for (int c = lane * 4; c < hd; c += 256) {
            const float4 kv = kv_load4(kr + c), qv = *reinterpret_cast<const float4*>(q + c);
            a = fmaf(qv.x, kv.x, fmaf(qv.y, kv.y, fmaf(qv.z, kv.z, fmaf(qv.w, kv.w, a))));
        }
    } else {
        for (int c = lane; c < hd; c += 64) a = fmaf(q[c], (float)kr[c], a);
    }
    a = wave_sum(a);
    if (lane == 0) S[(size_t)row * dm.H + h] = a;
}

// grid (B, H): softmax over the window's n scores, in place (P[row, h], stride H)
__global__ __launch_bounds__(256) void ragged_softmax_kernel(RaggedAttnDims dm, const int* __restrict__ offsets, float* __restrict__ P) {
    __shared__ float red[16];
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const int o0 = offsets[b], n = offsets[b + 1] - o0;
    if (n == 0) return;
    float* p = P + (size_t)o0 * dm.H + h;
    float m = -INFINITY;
    for (int i = tid; i < n; i += 256) m = fmaxf(m, p[(size_t)i * dm.H]);
    m = block_max(m, red);
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) sum += expf(p[(size_t)i * dm.H] - m);
    sum = block_sum(sum, red);
    const float inv = 1.f / sum;
    for (int i = tid; i < n; i += 256) p[(size_t)i * dm.H] = expf(p[(size_t)i * dm.H] - m) * inv;
}

// grid (B, ceil(T d / 256)): ctx[b] = sum of the window's chunk partials (windows of <= 1 chunk were written directly)
__global__ __launch_bounds__(256) void ragged_ctx_reduce_kernel(RaggedAttnDims dm, const int* __restrict__ offsets,
                                                                 const float* __restrict__ part, int maxch, float* __restrict__ ctx,
                                                                 bf16_t* __restrict__ ctx_h) {
    const int b = blockIdx.x;
    const int n = offsets[b + 1] - offsets[b], nch = (n + RAGGED_CH - 1) / RAGGED_CH;
    if (nch <= 1) return;
    const size_t Td = (size_t)dm.T * dm.H * dm.hd, x = (size_t)blockIdx.y * 256 + threadIdx.x;
    if (x >= Td) return;
    float a = 0.f;
    for (int c = 0; c < nch; ++c) a += part[((size_t)b * maxch + c) * Td + x];
    if (ctx) ctx[(size_t)b * Td + x] = a;
    if (ctx_h) ctx_h[(size_t)b * Td + x] = (bf16_t)a;
}

// !SPLIT: grid (B, H, ceil(hd/256)), 256 threads; LDS: sc[N] | atile[TT*64] | red[16]
//  SPLIT: grid (B * maxch, H, ceil(hd/256)); P holds the normalised weights already; LDS: sc[RAGGED_CH] | atile | red
template <bool SPLIT, typename KT>
__global__ __launch_bounds__(256) void ragged_attn_fwd_kernel(RaggedAttnDims dm, const int* __restrict__ offsets,
                                                               const int* __restrict__ rowmap,
                                                               const KT* __restrict__ KVp, const float* __restrict__ qs,
                                                               float* __restrict__ P, float* __restrict__ ctx, DropCfg drop,
                                                               uint64_t site, bf16_t* __restrict__ ctx_h, float* __restrict__ part,
                                                               int maxch) {
    extern __shared__ float lds[];
    float* sc = lds;
    float* atile = lds + (SPLIT ? RAGGED_CH : dm.N);
    float* red = atile + TT * 64;
    const int b = SPLIT ? blockIdx.x / maxch : blockIdx.x, ch = SPLIT ? blockIdx.x % maxch : 0;
    const int h = blockIdx.y, ez = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hd = dm.hd, d = dm.H * hd, ld = 2 * d, T = dm.T;
    const int ob = offsets[b], nfull = offsets[b + 1] - ob;
    const int e = ez * 256 + tid;
    const bool valid = e < hd;
    if (nfull == 0) {   // no notes: the caller zeroes the row after out_proj anyway (M_txt); keep ctx defined
        if (valid && ch == 0) for (int t = 0; t < T; ++t) {
            const size_t o = (size_t)(b * T + t) * d + h * hd + e;
            if (ctx) ctx[o] = 0.f;
            if (ctx_h) ctx_h[o] = (bf16_t)0.f;
        }
        return;
    }
    const int i_lo = ch * RAGGED_CH;
    if (i_lo >= nfull) return;
    const int o0 = ob + i_lo, n = SPLIT ? min(RAGGED_CH, nfull - i_lo) : nfull;
    // where this workgroup's (T, 256-column) block of sums goes: ctx itself, or the chunk's partial
    float* dst = ctx;
    bf16_t* dst_h = ctx_h;
    size_t obase = (size_t)b * T * d;
    if (SPLIT && nfull > RAGGED_CH) { dst = part; dst_h = nullptr; obase = ((size_t)b * maxch + ch) * T * d; }
    if (SPLIT) {
        for (int i = tid; i < n; i += 256) sc[i] = P[(size_t)(o0 + i) * dm.H + h];
        __syncthreads();
    } else {
    // two notes per pass, their key loads (hd/64 each, unrolled) all in flight together: a wave's notes are a chain of
    // global-load round trips otherwise (n/4 of them)
    for (int i = wave; i < n; i += 8) {
        const int i2 = i + 4;
        const KT* kr = KVp + (size_t)(o0 + i) * ld + h * hd;
        const KT* kr2 = KVp + (size_t)(o0 + (i2 < n ? i2 : i)) * ld + h * hd;
        float a = 0.f, a2 = 0.f;
#pragma unroll 12
        for (int c = lane; c < hd; c += 64) {
            const float q = qs[h * hd + c];
            a = fmaf(q, (float)kr[c], a);
            a2 = fmaf(q, (float)kr2[c], a2);
        }
        a = wave_sum(a);
        a2 = wave_sum(a2);
        if (lane == 0) {
            sc[i] = a;
            if (i2 < n) sc[i2] = a2;
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int i = tid; i < n; i += 256) m = fmaxf(m, sc[i]);
    m = block_max(m, red);
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) { const float p = expf(sc[i] - m); sc[i] = p; sum += p; }
    sum = block_sum(sum, red);
    const float inv = 1.f / sum;
    for (int i = tid; i < n; i += 256) {
        const float p = sc[i] * inv;
        sc[i] = p;
        if (ez == 0) P[(size_t)(o0 + i) * dm.H + h] = p;
    }
    __syncthreads();
    }

    const KT* vbase = KVp + (size_t)o0 * ld + d + h * hd + e;
    if (drop.p <= 0.f) {   // every forecast step sees the same weights
        if (!valid) return;
        float acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < n; ++i) acc = fmaf(sc[i], (float)vbase[(size_t)i * ld], acc);
        for (int t = 0; t < T; ++t) {
            const size_t o = obase + (size_t)t * d + h * hd + e;
            if (dst) dst[o] = acc;
            if (dst_h) dst_h[o] = (bf16_t)acc;
        }
        return;
    }
    for (int t0 = 0; t0 < T; t0 += TT) {
        float acc[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[tt] = 0.f;
        for (int i0 = 0; i0 < n; i0 += 64) {
            __syncthreads();
            for (int x = tid; x < TT * 64; x += 256) {
                const int tt = x >> 6, ii = x & 63, t = t0 + tt, i = i0 + ii;
                float a = 0.f;
                if (t < T && i < n) {
                    const int n_orig = rowmap[o0 + i] - b * dm.N;
                    const uint64_t idx = ((uint64_t)(b * T + t) * dm.H + h) * dm.N + n_orig;
                    a = sc[i] * dropout_scale(drop, site, idx);
                }
                atile[x] = a;
            }
            __syncthreads();
            if (valid) {
                const int cnt = min(64, n - i0);
                for (int ii = 0; ii < cnt; ii += 8) {     // eight notes per step: their V loads are in flight together
                    float v[8];                           // (one load -> 32 FMAs -> next load serialises on latency)
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = (ii + u < cnt) ? (float)vbase[(size_t)(i0 + ii + u) * ld] : 0.f;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int tt = 0; tt < TT; ++tt) acc[tt] = fmaf(atile[tt * 64 + ii + u], v[u], acc[tt]);   // tile is 0 past n
                }
            }
        }
        if (valid) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
                if (t0 + tt < T) {
                    const size_t o = obase + (size_t)(t0 + tt) * d + h * hd + e;
                    if (dst) dst[o] = acc[tt];
                    if (dst_h) dst_h[o] = (bf16_t)acc[tt];
                }
        }
    }
}

// 16-byte pieces of a K row
template <typename KT> struct KVec;
template <> struct KVec<float> {
    static constexpr int W = 4;
    typedef float4 raw;
    static __device__ __forceinline__ raw zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ void unpack(const raw& r, float (&o)[4]) { o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w; }
};
template <> struct KVec<bf16_t> {
    static constexpr int W = 8;
    typedef uint4 raw;
    static __device__ __forceinline__ raw zero() { return make_uint4(0u, 0u, 0u, 0u); }
    static __device__ __forceinline__ void unpack(const raw& r, float (&o)[8]) {
        const unsigned int w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(w[j] << 16); o[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
    }
};
constexpr int RS_PCS = 3;       // 16-byte pieces of a K row per lane: hd <= 192 W

// Windows of <= NV (32 / 64) padded notes and <= TT forecast steps, the fusion's own regime: ONE round of global loads.  The general
// kernel above walks a window's notes as dependent rounds -- two notes' keys per wave and pass, eight value rows per step, the
// dropout tile behind the softmax: ~14 global round trips of 1 - 2 us for a 32-note window (27.9 us in the 64-window step).  Here
// every load a thread needs -- its value column (NV rows), eight notes' keys per wave as 16-byte pieces, the row map -- is issued
// up front, the dropout scales are generated while they fly, each wave does the softmax over the <= 64 scores in its own lanes
// (no second barrier; the weights reach the columns by v_readlane), one barrier in all.  grid (B, H, ceil(hd/256)), 256 threads.
template <typename KT, int NV>
__global__ __launch_bounds__(256) void ragged_attn_fwd_short_kernel(RaggedAttnDims dm, const int* __restrict__ offsets,
                                                                     const int* __restrict__ rowmap, const KT* __restrict__ KVp,
                                                                     const float* __restrict__ qs, float* __restrict__ P,
                                                                     float* __restrict__ ctx, DropCfg drop, uint64_t site,
                                                                     bf16_t* __restrict__ ctx_h) {
    typedef KVec<KT> V;
    constexpr int W = V::W;
    __shared__ float sc[NV];
    __shared__ __attribute__((aligned(16))) float mt[TT * NV];       // dropout scale of (forecast step, note); 0 past T / n
    const int b = blockIdx.x, h = blockIdx.y, ez = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hd = dm.hd, d = dm.H * hd, ld = 2 * d, T = dm.T;
    const int ob = offsets[b], n = offsets[b + 1] - ob;
    const int e = ez * 256 + tid;
    const bool valid = e < hd;
    if (n == 0) {   // no notes: the caller zeroes the row after out_proj anyway (M_txt); keep ctx defined
        if (valid) for (int t = 0; t < T; ++t) {
            const size_t o = (size_t)(b * T + t) * d + h * hd + e;
            if (ctx) ctx[o] = 0.f;
            if (ctx_h) ctx_h[o] = (bf16_t)0.f;
        }
        return;
    }
    // this thread's value column (rows past n: the last row again, weight 0)
    KT vr[NV];
    {
        const KT* vbase = KVp + (size_t)ob * ld + d + h * hd + (valid ? e : 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) vr[i] = vbase[(size_t)(i < n ? i : n - 1) * ld];
    }
    // the query's pieces, then eight notes' keys per wave and pass: note i0 + wave + 4 u
    const int nvp = hd / W;
    float q[RS_PCS][W];
#pragma unroll
    for (int pc = 0; pc < RS_PCS; ++pc) {
        const int piece = lane + 64 * pc;
#pragma unroll
        for (int j = 0; j < W; j += 4) {
            const float4 t = piece < nvp ? *reinterpret_cast<const float4*>(qs + h * hd + piece * W + j) : make_float4(0.f, 0.f, 0.f, 0.f);
            q[pc][j] = t.x; q[pc][j + 1] = t.y; q[pc][j + 2] = t.z; q[pc][j + 3] = t.w;
        }
    }
    typename V::raw kr[8][RS_PCS];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i = wave + 4 * u;
        const KT* kp = KVp + (size_t)(ob + (i < n ? i : n - 1)) * ld + h * hd;
#pragma unroll
        for (int pc = 0; pc < RS_PCS; ++pc) {
            const int piece = lane + 64 * pc;
            kr[u][pc] = piece < nvp ? *reinterpret_cast<const typename V::raw*>(kp + piece * W) : V::zero();
        }
    }
    // the dropout scales while the loads are in flight
    {
        const uint64_t seed = drop.seed + ((drop.p > 0.f && drop.seed_dev) ? *drop.seed_dev : 0ull);
        for (int x = tid; x < TT * NV; x += 256) {
            const int tt = x / NV, ii = x - tt * NV;
            float a = 0.f;
            if (tt < T && ii < n) {
                a = 1.f;
                if (drop.p > 0.f) {
                    const int n_orig = rowmap[ob + ii] - b * dm.N;
                    const uint64_t idx = ((uint64_t)(b * T + tt) * dm.H + h) * dm.N + n_orig;
                    a = dropout_scale(seed, site, idx, drop.p, drop.inv_keep);
                }
            }
            mt[x] = a;
        }
    }
    for (int i0 = 0;;) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float a = 0.f;
#pragma unroll
            for (int pc = 0; pc < RS_PCS; ++pc) {
                float k[W];
                V::unpack(kr[u][pc], k);
#pragma unroll
                for (int j = 0; j < W; ++j) a = fmaf(q[pc][j], k[j], a);
            }
            a = wave_sum(a);
            const int i = i0 + wave + 4 * u;
            if (lane == 0 && i < n) sc[i] = a;
        }
        i0 += 32;
        if (i0 >= NV || i0 >= n) break;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + wave + 4 * u;
            const KT* kp = KVp + (size_t)(ob + (i < n ? i : n - 1)) * ld + h * hd;
#pragma unroll
            for (int pc = 0; pc < RS_PCS; ++pc) {
                const int piece = lane + 64 * pc;
                kr[u][pc] = piece < nvp ? *reinterpret_cast<const typename V::raw*>(kp + piece * W) : V::zero();
            }
        }
    }
    __syncthreads();
    // softmax over the window's n <= 64 scores, in every wave's own lanes
    const float s_l = lane < n ? sc[lane] : -INFINITY;
    const float m = wave_max(s_l);
    float p = lane < n ? expf(s_l - m) : 0.f;
    p *= 1.f / wave_sum(p);
    if (wave == 0 && ez == 0 && lane < n) P[(size_t)(ob + lane) * dm.H + h] = p;
    if (!valid) return;
    float pv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) pv[i] = lane_bcast(p, i) * (float)vr[i];        // (weight 0 past n)
    const size_t obase = (size_t)b * T * d + h * hd + e;
#pragma unroll 2
    for (int tt = 0; tt < T; ++tt) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NV; i += 4) {
            const float4 m4 = *reinterpret_cast<const float4*>(mt + tt * NV + i);
            acc = fmaf(m4.x, pv[i], acc); acc = fmaf(m4.y, pv[i + 1], acc); acc = fmaf(m4.z, pv[i + 2], acc); acc = fmaf(m4.w, pv[i + 3], acc);
        }
        const size_t o = obase + (size_t)tt * d;
        if (ctx) ctx[o] = acc;
        if (ctx_h) ctx_h[o] = (bf16_t)acc;
    }
}

constexpr int MT = 4;   // backward keeps the dropout scales of up to MT*64 forecast steps in registers

// Backward, part 1.  grid (B, H, ceil(hd/64)), 256 threads: every workgroup owns 64 head columns of one window,
// its 4 waves take the window's notes round-robin, one column per lane:
//   g[c]      = sum_t m[t,i] * dctx[b,t,h,c]            (m = dropout scale of the attention weight, 1 if p = 0)
//   dv[i,c]   = p[i] * g[c]
//   dp[i]    += sum_c g[c] * v[i,c]                     (fp32 atomic: hd/64 partial sums per note)
template <typename KT>
__global__ __launch_bounds__(256) void ragged_attn_bwd_dv_kernel(RaggedAttnDims dm, const int* __restrict__ offsets,
                                                                  const int* __restrict__ rowmap,
                                                                  const KT* __restrict__ KVp, const float* __restrict__ P,
                                                                  const float* __restrict__ dctx, float* __restrict__ dKVp,
                                                                  float* __restrict__ dp_buf, DropCfg drop, uint64_t site,
                                                                  bf16_t* __restrict__ dKVp_h, int maxch, size_t dp_stride) {
    // maxch > 1: grid.x = B * maxch, workgroup (b, ch) takes notes [ch * RAGGED_CH, +RAGGED_CH) of window b
    // dp_stride != 0: every 64-column slice writes its partial dp into its own slab (slab z at dp_buf + z * dp_stride; exactly one
    // writer per note and slice, part 2 adds the slabs): no zero-fill in front of this kernel and no atomics
    const int b = blockIdx.x / maxch, ch = blockIdx.x % maxch;
    const int h = blockIdx.y, c = blockIdx.z * 64 + (threadIdx.x & 63);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hd = dm.hd, d = dm.H * hd, ld = 2 * d, T = dm.T;
    const int ob = offsets[b], nfull = offsets[b + 1] - ob;
    const int i_lo = maxch > 1 ? ch * RAGGED_CH : 0;
    if (i_lo >= nfull) return;
    const int o0 = ob + i_lo, n = maxch > 1 ? min(RAGGED_CH, nfull - i_lo) : nfull;
    const bool valid = c < hd;
    const float* dc = dctx + (size_t)b * T * d + h * hd + c;   // row t at dc + t*d
    float gsum = 0.f;
    if (drop.p <= 0.f && valid)
        for (int t = 0; t < T; ++t) gsum += dc[(size_t)t * d];
    // the upstream gradient column is the same for every note of the window: up to DCR forecast steps stay in registers
    // (re-reading them per note made every note wait on T dependent loads)
    constexpr int DCR = 32;
    float dcv[DCR];
#pragma unroll
    for (int t = 0; t < DCR; ++t) dcv[t] = (valid && drop.p > 0.f && t < T) ? dc[(size_t)t * d] : 0.f;
    for (int i = wave; i < n; i += 4) {
        float g = gsum;
        if (drop.p > 0.f) {
            const int n_orig = rowmap[o0 + i] - b * dm.N;
            float mreg[MT];
#pragma unroll
            for (int k = 0; k < MT; ++k) {
                const int t = k * 64 + lane;
                mreg[k] = 0.f;
                if (t < T) {
                    const uint64_t idx = ((uint64_t)(b * T + t) * dm.H + h) * dm.N + n_orig;
                    mreg[k] = dropout_scale(drop, site, idx);
                }
            }
            g = 0.f;
#pragma unroll
            for (int tt = 0; tt < DCR; ++tt) g = fmaf(__shfl(mreg[0], tt, 64), dcv[tt], g);      // steps 0..DCR-1 (dcv is 0 past T)
#pragma unroll
            for (int k = 0; k < MT; ++k) {
                const int tcnt = min(64, T - k * 64);
                for (int tt = (k == 0 ? DCR : 0); tt < tcnt; ++tt) {
                    const float mt = __shfl(mreg[k], tt, 64);
                    if (valid) g = fmaf(mt, dc[(size_t)(k * 64 + tt) * d], g);
                }
            }
        }
        float a = 0.f;
        if (valid) {
            const size_t off = (size_t)(o0 + i) * ld + d + h * hd + c;
            a = g * (float)KVp[off];
            const float dvv = P[(size_t)(o0 + i) * dm.H + h] * g;
            if (dKVp) dKVp[off] = dvv;
            if (dKVp_h) dKVp_h[off] = (bf16_t)dvv;
        }
        a = wave_sum(a);
        if (lane == 0) {
            if (dp_stride) dp_buf[(size_t)blockIdx.z * dp_stride + (size_t)(o0 + i) * dm.H + h] = a;
            else atomicAdd(dp_buf + (size_t)(o0 + i) * dm.H + h, a);
        }
    }
}

// Backward, part 1 for long windows (chunked path).  grid (B * maxch, H, ceil(hd/256)), 256 threads, one column per thread,
// the chunk's RAGGED_CH notes walked by every thread.  The dropout scales m[t, i] of the chunk are generated ONCE per
// workgroup into LDS (four consecutive notes share one Philox call when their original positions are consecutive) instead
// of once per wave and 64-column slice, g[i, c] = sum_t m[t, i] dctx[t, c] reads them as wave-wide broadcasts against the
// T upstream values the thread keeps in registers (T <= 32; beyond that they are re-read through L1), and the per-note
// dp partial sums of the four waves meet in LDS: one atomic per note and workgroup.
// LDS: mt[RAGGED_CH][Tp] (dropout only; note-major so that a note's T scales are a few 16-byte broadcast reads; Tp = T
// rounded up to 4, 32 when T <= 32, padding zero) | dpw[4][RAGGED_CH]
template <bool REG, typename KT>
__global__ __launch_bounds__(256) void ragged_attn_bwd_dv_long_kernel(RaggedAttnDims dm, const int* __restrict__ offsets,
                                                                       const int* __restrict__ rowmap,
                                                                       const KT* __restrict__ KVp, const float* __restrict__ P,
                                                                       const float* __restrict__ dctx, float* __restrict__ dKVp,
                                                                       float* __restrict__ dp_buf, DropCfg drop, uint64_t site,
                                                                       bf16_t* __restrict__ dKVp_h, int maxch) {
    extern __shared__ float lds[];
    const int T = dm.T;
    const bool dropping = drop.p > 0.f;
    const int Tp = REG ? 32 : ((T + 3) & ~3);
    float* mt = lds;
    float* dpw = lds + (dropping ? (size_t)Tp * RAGGED_CH : 0);
    const int b = blockIdx.x / maxch, ch = blockIdx.x % maxch, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = blockIdx.z * 256 + tid;
    const int hd = dm.hd, d = dm.H * hd, ld = 2 * d;
    const int ob = offsets[b], nfull = offsets[b + 1] - ob, i_lo = ch * RAGGED_CH;
    if (i_lo >= nfull) return;
    const int o0 = ob + i_lo, n = min(RAGGED_CH, nfull - i_lo);
    const bool valid = c < hd;
    if (dropping) {
        const int n4 = (n + 3) >> 2;
        for (int x = tid; x < n * (Tp - T); x += 256) {      // zero padding t in [T, Tp)
            const int i = x / (Tp - T), t = T + (x - i * (Tp - T));
            mt[i * Tp + t] = 0.f;
        }
        for (int x = tid; x < T * n4; x += 256) {
            const int t = x / n4, i = (x - t * n4) * 4;
            const int r0 = rowmap[o0 + i] - b * dm.N;
            const uint64_t base = ((uint64_t)(b * T + t) * dm.H + h) * dm.N;
            float sc4[4];
            const bool run = i + 3 < n && (((base + r0) & 3) == 0) && rowmap[o0 + i + 3] - b * dm.N == r0 + 3;
            if (run) {
                dropout_scale4(drop, site, base + r0, sc4);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    sc4[u] = i + u < n ? dropout_scale(drop, site, base + (uint64_t)(rowmap[o0 + i + u] - b * dm.N)) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u < n) mt[(i + u) * Tp + t] = sc4[u];
        }
    }
    const float* dc = dctx + (size_t)b * T * d + h * hd + c;   // row t at dc + t*d
    constexpr int DCR = 32;
    float dcv[DCR];
    float gsum = 0.f;
#pragma unroll
    for (int t = 0; t < DCR; ++t) {
        dcv[t] = (valid && t < T) ? dc[(size_t)t * d] : 0.f;
        gsum += dcv[t];
    }
    if (!REG && valid)
        for (int t = DCR; t < T; ++t) gsum += dc[(size_t)t * d];
    __syncthreads();
    const size_t voff = (size_t)o0 * ld + d + h * hd + c;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
        float g = gsum;
        if (dropping) {
            g = 0.f;
            if (REG) {
                const float4* m4 = reinterpret_cast<const float4*>(mt + i * 32);
#pragma unroll
                for (int t4 = 0; t4 < DCR / 4; ++t4) {
                    const float4 m = m4[t4];
                    g = fmaf(m.x, dcv[4 * t4], fmaf(m.y, dcv[4 * t4 + 1], fmaf(m.z, dcv[4 * t4 + 2], fmaf(m.w, dcv[4 * t4 + 3], g))));
                }
            } else {
                for (int t = 0; t < T; ++t) g = fmaf(mt[i * Tp + t], valid ? dc[(size_t)t * d] : 0.f, g);
            }
        }
        float a = 0.f;
        if (valid) {
            const size_t off = voff + (size_t)i * ld;
            a = g * (float)KVp[off];
            const float dvv = P[(size_t)(o0 + i) * dm.H + h] * g;
            if (dKVp) dKVp[off] = dvv;
            if (dKVp_h) dKVp_h[off] = (bf16_t)dvv;
        }
        a = wave_sum(a);
        if (lane == 0) dpw[wave * RAGGED_CH + i] = a;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256)
        atomicAdd(dp_buf + (size_t)(o0 + i) * dm.H + h, dpw[i] + dpw[RAGGED_CH + i] + dpw[2 * RAGGED_CH + i] + dpw[3 * RAGGED_CH + i]);
}

// Backward, part 2.  grid (B, H, ceil(hd/256)), 256 threads (every workgroup recomputes the window's ds -- n values -- and
// owns a 256-column slice of dk / dqs_part; one workgroup per window made 64 of them walk all hd columns); LDS: ds[N] | red[16].
//   ds[i] = p[i] (dp[i] - sum_j p[j] dp[j]);  dk[i,:] = ds[i] * qs_h;  dqs_part[b, h, :] = sum_i ds[i] k[i,:]
template <typename KT>
__global__ __launch_bounds__(256) void ragged_attn_bwd_ds_kernel(RaggedAttnDims dm, const int* __restrict__ offsets,
                                                                  const KT* __restrict__ KVp, const float* __restrict__ qs,
                                                                  const float* __restrict__ P, const float* __restrict__ dp_buf,
                                                                  float* __restrict__ dKVp, float* __restrict__ dqs_part,
                                                                  bf16_t* __restrict__ dKVp_h, int maxch, int dp_slabs, size_t dp_stride) {
    // dp_slabs > 1: dp arrives as that many partial slabs (see part 1)
    // maxch > 1: grid.x = B * maxch; workgroup (b, ch) recomputes the window's sum_j p_j dp_j (n values), owns the chunk's
    // ds / dk rows and adds its share of dqs_part (zero-filled by the launcher) with one atomic per column
    extern __shared__ float lds[];
    float* ds = lds;
    float* red = lds + (maxch > 1 ? RAGGED_CH : dm.N);
    const int b = blockIdx.x / maxch, ch = blockIdx.x % maxch;
    const int h = blockIdx.y, c0 = blockIdx.z * 256;
    const int tid = threadIdx.x;
    const int hd = dm.hd, d = dm.H * hd, ld = 2 * d;
    const int ob = offsets[b], nfull = offsets[b + 1] - ob;
    const int c = c0 + tid;
    if (nfull == 0) {
        if (c < hd && maxch == 1) dqs_part[(size_t)b * d + h * hd + c] = 0.f;
        return;
    }
    const int i_lo = maxch > 1 ? ch * RAGGED_CH : 0;
    if (i_lo >= nfull) return;
    const int o0 = ob + i_lo, n = maxch > 1 ? min(RAGGED_CH, nfull - i_lo) : nfull;
    float part = 0.f;
    for (int i = tid; i < nfull; i += 256) {
        const float p = P[(size_t)(ob + i) * dm.H + h];
        float g = dp_buf[(size_t)(ob + i) * dm.H + h];
        for (int z = 1; z < dp_slabs; ++z) g += dp_buf[(size_t)z * dp_stride + (size_t)(ob + i) * dm.H + h];
        if (i >= i_lo && i < i_lo + n) ds[i - i_lo] = g;
        part = fmaf(p, g, part);
    }
    const float dot = block_sum(part, red);
    __syncthreads();
    for (int i = tid; i < n; i += 256) ds[i] = P[(size_t)(o0 + i) * dm.H + h] * (ds[i] - dot);
    __syncthreads();
    if (c >= hd) return;
    const float q = qs[h * hd + c];
    const KT* kc = KVp + (size_t)o0 * ld + h * hd + c;
    const size_t dk0 = (size_t)o0 * ld + h * hd + c;
    float a = 0.f;
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
        a = fmaf(ds[i], (float)kc[(size_t)i * ld], a);
        const float dkv = ds[i] * q;
        if (dKVp) dKVp[dk0 + (size_t)i * ld] = dkv;
        if (dKVp_h) dKVp_h[dk0 + (size_t)i * ld] = (bf16_t)dkv;
    }
    if (maxch > 1) atomicAdd(dqs_part + (size_t)b * d + h * hd + c, a);
    else dqs_part[(size_t)b * d + h * hd + c] = a;
}

// ---- dense attention rows: one wave per (b,h,l) row of length S ---------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(float* __restrict__ sc, float* __restrict__ A, int rows, int HL,
                                                                int S, const unsigned char* __restrict__ live, DropCfg drop,
                                                                uint64_t site, int causal_L) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = sc + (size_t)row * S;
    float* a = A + (size_t)row * S;
    if (live && !live[row / HL]) {
        for (int i = lane; i < S; i += 64) { p[i] = 0.f; a[i] = 0.f; }
        return;
    }
    // causal_L > 0: TriangularCausalMask (utils/masking.py): query l = row % L may only see keys s <= l
    const int Sv = causal_L > 0 ? min(S, (row % causal_L) + 1) : S;
    float m = -INFINITY;
    for (int i = lane; i < Sv; i += 64) m = fmaxf(m, p[i]);
    m = wave_max(m);
    float sum = 0.f;
    for (int i = lane; i < Sv; i += 64) sum += expf(p[i] - m);
    const float inv = 1.f / wave_sum(sum);
    for (int i = lane; i < S; i += 64) {
        const float v = (i < Sv) ? expf(p[i] - m) * inv : 0.f;
        p[i] = v;
        a[i] = v * dropout_scale(drop, site, (uint64_t)row * S + i);
    }
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(float* __restrict__ dA, const float* __restrict__ P, int rows,
                                                                int S, DropCfg drop, uint64_t site) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* g = dA + (size_t)row * S;
    const float* p = P + (size_t)row * S;
    float dot = 0.f;
    for (int i = lane; i < S; i += 64) {
        const float dpv = g[i] * dropout_scale(drop, site, (uint64_t)row * S + i);
        g[i] = dpv;
        dot = fmaf(p[i], dpv, dot);
    }
    dot = wave_sum(dot);
    for (int i = lane; i < S; i += 64) g[i] = p[i] * (g[i] - dot);
}

// ---- short-sequence self-attention on the packed in-projection output qkv (B, L, 3, H, E), L <= SHORT_L ----------
// tPatchGNN attends over the M patches of one variable: M = 2 at the benchmark configuration, i.e. 512 sequences of
// length 2.  As batched GEMMs + a row softmax that is 3 launches forward and 5 backward of ~8 us each for 2x2 score
// matrices; here one thread owns one (sequence, head, position), everything in registers, one launch per direction.
// Same arithmetic and the same dropout indexing (site, ((b*H+h)*L + l)*L + s) as the GEMM + softmax_rows path.
constexpr int SHORT_L = 8;       // longest sequence
constexpr int SHORT_EV = 16;     // float4 chunks per row: head dim <= 64, multiple of 4
// The kernels are instantiated for (SL, SEV) = bounds on (L, E / 4): every per-thread array and unrolled loop is sized by
// them, and the (8, 16) instance alone needs 2.9 KB of scratch per lane in the backward (25 us at tPatchGNN's L = 2, E = 32).

struct ShortDims { int B, L, H, E; };

// one row of E floats as (predicated) float4 chunks: every chunk is an independent 16-byte load, issued back to back
template <int SEV>
__device__ __forceinline__ void short_load(const float* __restrict__ p, int ev, float4 (&r)[SEV]) {
#pragma unroll
    for (int c = 0; c < SEV; ++c) r[c] = c < ev ? reinterpret_cast<const float4*>(p)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int SEV>
__device__ __forceinline__ float short_dot(const float4 (&a)[SEV], const float4 (&b)[SEV]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < SEV; ++c) {
        s = fmaf(a[c].x, b[c].x, s); s = fmaf(a[c].y, b[c].y, s); s = fmaf(a[c].z, b[c].z, s); s = fmaf(a[c].w, b[c].w, s);
    }
    return s;
}
template <int SEV>
__device__ __forceinline__ void short_axpy(float w, const float4 (&x)[SEV], float4 (&acc)[SEV]) {
#pragma unroll
    for (int c = 0; c < SEV; ++c) {
        acc[c].x = fmaf(w, x[c].x, acc[c].x); acc[c].y = fmaf(w, x[c].y, acc[c].y);
        acc[c].z = fmaf(w, x[c].z, acc[c].z); acc[c].w = fmaf(w, x[c].w, acc[c].w);
    }
}
template <int SEV>
__device__ __forceinline__ void short_store(float* __restrict__ p, int ev, float scale, const float4 (&r)[SEV]) {
#pragma unroll
    for (int c = 0; c < SEV; ++c)
        if (c < ev) reinterpret_cast<float4*>(p)[c] = make_float4(scale * r[c].x, scale * r[c].y, scale * r[c].z, scale * r[c].w);
}

// element offsets inside the packed (B, L, 3, H, E) tensor
__device__ __forceinline__ size_t short_off(const ShortDims& d, int b, int l, int which, int h) {
    return (((size_t)b * d.L + l) * 3 + which) * d.H * d.E + (size_t)h * d.E;
}

// scores of query row l of (b, h): p[s] = softmax_s(scale q[l].k[s]) (causal: s <= l), a[s] = p[s] * dropscale
// (b indexes `qkv`, which may be a staged copy of a few sequences; bg is the sequence's global index, for the dropout row)
template <int SL, int SEV>
__device__ __forceinline__ void short_row(const ShortDims& d, const float* __restrict__ qkv, int b, int bg, int h, int l, float scale,
                                          int causal, const DropCfg& drop, uint64_t site, float (&p)[SL], float (&a)[SL]) {
    const int ev = d.E >> 2, Sv = causal ? l + 1 : d.L;
    float4 q[SEV], k[SEV];
    short_load(qkv + short_off(d, b, l, 0, h), ev, q);
    float m = -INFINITY;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        float acc = 0.f;
        if (s < Sv) {
            short_load(qkv + short_off(d, b, s, 1, h), ev, k);
            acc = scale * short_dot(q, k);
            m = fmaxf(m, acc);
        }
        p[s] = acc;
    }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        p[s] = s < Sv ? expf(p[s] - m) : 0.f;
        sum += p[s];
    }
    const float inv = 1.f / sum;
    const uint64_t row = ((uint64_t)bg * d.H + h) * d.L + l;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        p[s] *= inv;
        a[s] = s < d.L ? p[s] * dropout_scale(drop, site, row * d.L + s) : 0.f;
    }
}

template <int SL, int SEV>
__device__ __forceinline__ void attn_short_fwd_body(const ShortDims& d, const float* __restrict__ qkv, int b, int bg, int h, int l,
                                                    float scale, int causal, const DropCfg& drop, uint64_t site,
                                                    float* __restrict__ out) {
    const int ev = d.E >> 2;
    float p[SL], a[SL];
    short_row<SL, SEV>(d, qkv, b, bg, h, l, scale, causal, drop, site, p, a);
    float4 acc[SEV], v[SEV];
#pragma unroll
    for (int c = 0; c < SEV; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < SL; ++s)
        if (s < d.L) {
            short_load(qkv + short_off(d, b, s, 2, h), ev, v);
            short_axpy(a[s], v, acc);
        }
    short_store(out + (((size_t)bg * d.L + l) * d.H + h) * d.E, ev, 1.f, acc);
}

template <int SL, int SEV>
__global__ __launch_bounds__(256) void attn_short_fwd_kernel(ShortDims d, const float* __restrict__ qkv, float scale, int causal,
                                                              DropCfg drop, uint64_t site, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.B * d.H * d.L) return;
    const int l = i % d.L, h = (i / d.L) % d.H, b = i / (d.L * d.H);
    attn_short_fwd_body<SL, SEV>(d, qkv, b, b, h, l, scale, causal, drop, site, out);
}

// Staged form: a thread's work is a chain of ~(3 L + 2) dependent row loads, and with one thread per (sequence, head,
// position) the whole launch is a handful of waves -- pure load latency (12 us forward, 25 us backward at 512 sequences of
// 2).  Here a workgroup first copies its SB sequences' qkv (and dout) rows -- contiguous in memory -- into LDS with one
// round of coalesced 16-byte loads, and the per-thread chains then run against LDS.  grid ceil(B / SB).
template <int SL, int SEV>
__global__ __launch_bounds__(256) void attn_short_fwd_staged_kernel(ShortDims d, const float* __restrict__ qkv, float scale, int causal,
                                                                     DropCfg drop, uint64_t site, float* __restrict__ out, int SB) {
    extern __shared__ __attribute__((aligned(16))) float st[];
    const int b0 = blockIdx.x * SB, nb = min(SB, d.B - b0), per = d.L * 3 * d.H * d.E;
    const float4* src = reinterpret_cast<const float4*>(qkv + (size_t)b0 * per);
    for (int x = threadIdx.x; x < nb * per / 4; x += 256) reinterpret_cast<float4*>(st)[x] = src[x];
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= nb * d.H * d.L) return;
    const int l = t % d.L, h = (t / d.L) % d.H, bl = t / (d.L * d.H);
    attn_short_fwd_body<SL, SEV>(d, st, bl, b0 + bl, h, l, scale, causal, drop, site, out);
}

// thread (b, h, i): as query row i -> dq[i]; as key / value row i -> dk[i], dv[i] (recomputing every query row's softmax)
template <int SL, int SEV>
__device__ __forceinline__ void attn_short_bwd_body(const ShortDims& d, const float* __restrict__ qkv, const float* __restrict__ dout,
                                                    int b, int bg, int h, int i, float scale, int causal, const DropCfg& drop,
                                                    uint64_t site, float* __restrict__ dqkv) {
    const int ev = d.E >> 2;
    float wk[SL], wv[SL], wq[SL];      // dS[l][i] (l = 0..), A[l][i], dS[i][s]
#pragma unroll
    for (int l = 0; l < SL; ++l) { wk[l] = 0.f; wv[l] = 0.f; wq[l] = 0.f; }
#pragma unroll
    for (int l = 0; l < SL; ++l) {
        if (l < d.L) {
            float p[SL], a[SL], dp[SL];
            short_row<SL, SEV>(d, qkv, b, bg, h, l, scale, causal, drop, site, p, a);
            float4 g[SEV], v[SEV];
            short_load(dout + (((size_t)b * d.L + l) * d.H + h) * d.E, ev, g);
            const uint64_t row = ((uint64_t)bg * d.H + h) * d.L + l;
            float dot = 0.f;
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                float acc = 0.f;
                if (s < d.L) {
                    short_load(qkv + short_off(d, b, s, 2, h), ev, v);
                    acc = short_dot(g, v) * dropout_scale(drop, site, row * d.L + s);
                }
                dp[s] = acc;
                dot = fmaf(p[s], acc, dot);
            }
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const float ds = p[s] * (dp[s] - dot);
                if (s == i) { wk[l] = ds; wv[l] = a[s]; }
                if (l == i) wq[s] = ds;
            }
        }
    }
    float4 acc[SEV], x[SEV];
    for (int which = 0; which < 3; ++which) {      // dq[i] = scale sum_j dS[i][j] k[j]; dk[i] = scale sum_j dS[j][i] q[j]; dv[i] = sum_j A[j][i] dout[j]
#pragma unroll
        for (int c = 0; c < SEV; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < SL; ++j)
            if (j < d.L) {
                const float* src = which == 0 ? qkv + short_off(d, b, j, 1, h)
                                 : which == 1 ? qkv + short_off(d, b, j, 0, h) : dout + (((size_t)b * d.L + j) * d.H + h) * d.E;
                short_load(src, ev, x);
                short_axpy(which == 0 ? wq[j] : which == 1 ? wk[j] : wv[j], x, acc);
            }
        short_store(dqkv + short_off(d, bg, i, which, h), ev, which == 2 ? 1.f : scale, acc);
    }
}

template <int SL, int SEV>
__global__ __launch_bounds__(256) void attn_short_bwd_kernel(ShortDims d, const float* __restrict__ qkv, const float* __restrict__ dout,
                                                              float scale, int causal, DropCfg drop, uint64_t site,
                                                              float* __restrict__ dqkv) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= d.B * d.H * d.L) return;
    const int i = t % d.L, h = (t / d.L) % d.H, b = t / (d.L * d.H);
    attn_short_bwd_body<SL, SEV>(d, qkv, dout, b, b, h, i, scale, causal, drop, site, dqkv);
}

template <int SL, int SEV>
__global__ __launch_bounds__(256) void attn_short_bwd_staged_kernel(ShortDims d, const float* __restrict__ qkv,
                                                                     const float* __restrict__ dout, float scale, int causal, DropCfg drop,
                                                                     uint64_t site, float* __restrict__ dqkv, int SB) {
    extern __shared__ __attribute__((aligned(16))) float st[];
    const int b0 = blockIdx.x * SB, nb = min(SB, d.B - b0), per = d.L * 3 * d.H * d.E, pero = d.L * d.H * d.E;
    float* sq = st;
    float* so = st + (size_t)SB * per;
    const float4* src = reinterpret_cast<const float4*>(qkv + (size_t)b0 * per);
    for (int x = threadIdx.x; x < nb * per / 4; x += 256) reinterpret_cast<float4*>(sq)[x] = src[x];
    const float4* srco = reinterpret_cast<const float4*>(dout + (size_t)b0 * pero);
    for (int x = threadIdx.x; x < nb * pero / 4; x += 256) reinterpret_cast<float4*>(so)[x] = srco[x];
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= nb * d.H * d.L) return;
    const int i = t % d.L, h = (t / d.L) % d.H, bl = t / (d.L * d.H);
    attn_short_bwd_body<SL, SEV>(d, sq, so, bl, b0 + bl, h, i, scale, causal, drop, site, dqkv);
}


// ---- dense T x T cross-attention of MMF_XAttn_Add over the prediction steps of a window (T <= 32), bf16 mode ------------
// As batched GEMMs + a row softmax this is 3 launches forward and 5 backward of 32 x 32 (x hd) problems per (window, head)
// on the serial section between the forward and the backward of the step: ~10 us apiece for 0.1 GFLOP in total.  Here one
// workgroup per (window, head, 256-column chunk of the head dimension) does a whole direction for its chunk of the outputs: the two k-contiguous products (Q K^T, dO V^T) as MFMA tiles straight
// from global fp32 rows (a 16 x 16 tile and half of the head dimension per wave, eight waves), the softmax / its backward on the 32 x 32 tile in LDS (8 lanes per
// row), and the products against V / K / dO / Q from bf16 LDS images of 256-column chunks, read with the hardware
// transpose where the reduction index is the row index of the image.  Operand rounding as in the GEMM path (bf16 operands,
// fp32 accumulation); same dropout indexing as softmax_rows_{fwd,bwd}: (site, ((b*H + h)*T + i)*T + j).
constexpr int XS_T = 32;         // padded tile: rows / columns beyond T are zero
constexpr int XS_PT = 40;        // pitch of the 32 x 32 bf16 tiles (80-byte rows)
constexpr int XS_EC = 256;       // columns of the head dimension staged per pass
constexpr int XS_PC = XS_EC + 8; // pitch of the staged chunk images
constexpr int XS_PS = 33;        // pitch of the fp32 score tile
constexpr int XS_GC = 16;        // widest C-column input of a generated operand (XsGen)
struct XSmallDims { int B, T, H, hd, d; };     // d = H * hd: pitch of Q / O / dO rows (K | V rows: 2d)
typedef short xs_s16x4 __attribute__((ext_vector_type(4)));
typedef short xs_s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 xs_load8(const float* __restrict__ src) {     // 8 consecutive fp32 -> bf16x8 (RNE)
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    const bf16x8 r = {(bf16_t)a.x, (bf16_t)a.y, (bf16_t)a.z, (bf16_t)a.w, (bf16_t)b.x, (bf16_t)b.y, (bf16_t)b.z, (bf16_t)b.w};
    return r;
}
// fragment of a row-major bf16 LDS tile: row = row0 + (lane & 15), k = k0 + (lane >> 4) * 8 ..
__device__ __forceinline__ bf16x8 xs_frag_row(const bf16_t* tile, int pitch, int row0, int k0, int fr, int fq) {
    return *reinterpret_cast<const bf16x8*>(tile + (row0 + fr) * pitch + k0 + fq * 8);
}
// hardware-transpose read of a 16 (row) x 8 (k) fragment from a [k][row] LDS image (see gemm.hip)
__device__ __forceinline__ bf16x8 xs_frag_kmajor(const bf16_t* tile, int pitch, int rbase, int kbase, int fr, int fq) {
    typedef __attribute__((address_space(3))) xs_s16x4 lds_s16x4;
    const int q = fr >> 2, pp = fr & 3;
    const bf16_t* a0 = tile + (kbase + fq * 8 + q) * pitch + rbase + 4 * pp;
    const xs_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    const xs_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * pitch));
    const xs_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x4 xs_mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// Sf / Sf2[i][j] = alpha * (partial) sum_e X[i][e] Y[j][e] on the padded 32 x 32 tile: wave w owns tile ((w & 3) >> 1, w & 1)
// and half (w >> 2) of the k-steps (Sf: first half, Sf2: second half; the callers add them); U k-steps of loads are in
// flight per wave -- the rows come from HBM / MALL, and one workgroup per window has nothing else to hide that latency with.
// Rows beyond T read row T - 1 (their results are masked by the callers).
template <int U>
__device__ __forceinline__ void xs_scores(int T, int hd, const float* __restrict__ X, size_t ldx, const float* __restrict__ Y, size_t ldy,
                                          float alpha, float* Sf, float* Sf2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4, mt = (wave & 3) >> 1, nt = wave & 1;
    const int half = wave >> 2, ksteps = hd >> 5, ks0 = (ksteps + 1) >> 1;
    const float* xr = X + (size_t)min(mt * 16 + fr, T - 1) * ldx + fq * 8;
    const float* yr = Y + (size_t)min(nt * 16 + fr, T - 1) * ldy + fq * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int k = (half ? ks0 : 0) * 32;
    const int kend = (half ? ksteps : ks0) * 32;
    for (; k + 32 * U <= kend; k += 32 * U) {
        bf16x8 av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { av[u] = xs_load8(xr + k + 32 * u); bv[u] = xs_load8(yr + k + 32 * u); }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = xs_mfma(av[u], bv[u], acc);
    }
    for (; k + 32 <= kend; k += 32) acc = xs_mfma(xs_load8(xr + k), xs_load8(yr + k), acc);
    if (half && (hd & 31)) {                   // hd % 32 == 16: the upper half of the last k-step is zero
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        bf16x8 at = z, bt = z;
        if (fq < 2) { at = xs_load8(xr + k); bt = xs_load8(yr + k); }
        acc = xs_mfma(at, bt, acc);
    }
    float* out = half ? Sf2 : Sf;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(mt * 16 + fq * 4 + r) * XS_PS + nt * 16 + fr] = alpha * acc[r];
}
// The same tile from bf16 LDS images of the two operands (hd <= XS_HD_STAGED, hd % 32 == 0): read straight from global memory
// every load instruction of a fragment touches 16 rows that are a row pitch (3 - 6 KB) apart, and the product ran at a fraction
// of the rate of the same bytes read row by row; staged, a wave's load covers 1 KB of one row.
constexpr int XS_HD_STAGED = 768;
__host__ __device__ inline bool xs_staged(int hd) { return hd <= XS_HD_STAGED && (hd & 31) == 0; }
// images imX, imY [32][pf] of X, Y (rows beyond T: zero): wave w owns rows w, w + 8, w + 16, w + 24 and its lanes stride the
// 16-byte chunks of a row (hd <= 768: three per lane) -- no index arithmetic beyond adds (a first version mapped a flat index to
// (row, chunk) with two run-time divisions per element: 7 us of the kernel for 98 KB), all loads of an operand in flight at once
constexpr int XS_SU = XS_HD_STAGED / 4 / 64;       // chunks per lane and row
__device__ __forceinline__ void xs_stage_rows(int T, int hd, const float* __restrict__ X, size_t ldx, float4 (&v)[4][XS_SU]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n4 = hd >> 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = wave + 8 * k;
#pragma unroll
        for (int u = 0; u < XS_SU; ++u) {
            const int c = lane + 64 * u;
            v[k][u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < T && c < n4) v[k][u] = reinterpret_cast<const float4*>(X + (size_t)r * ldx)[c];
        }
    }
}
__device__ __forceinline__ void xs_store_rows(int hd, const float4 (&v)[4][XS_SU], bf16_t* im, int pf) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n4 = hd >> 2;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < XS_SU; ++u) {
            const int c = lane + 64 * u;
            if (c < n4) {
                const bf16x4 hx = {(bf16_t)v[k][u].x, (bf16_t)v[k][u].y, (bf16_t)v[k][u].z, (bf16_t)v[k][u].w};
                *reinterpret_cast<bf16x4*>(im + (wave + 8 * k) * pf + 4 * c) = hx;
            }
        }
}
__device__ __forceinline__ void xs_stage_pair(int T, int hd, const float* __restrict__ X, size_t ldx, const float* __restrict__ Y, size_t ldy,
                                              bf16_t* imX, bf16_t* imY, int pf) {
    float4 vx[4][XS_SU], vy[4][XS_SU];
    xs_stage_rows(T, hd, X, ldx, vx);
    xs_stage_rows(T, hd, Y, ldy, vy);
    xs_store_rows(hd, vx, imX, pf);
    xs_store_rows(hd, vy, imY, pf);
}
__device__ __forceinline__ void xs_stage_one(int T, int hd, const float* __restrict__ X, size_t ldx, bf16_t* imX, int pf) {
    float4 vx[4][XS_SU];
    xs_stage_rows(T, hd, X, ldx, vx);
    xs_store_rows(hd, vx, imX, pf);
}
__device__ __forceinline__ float dot4(const float4 a, const float4 b, float acc) {
    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc);
    return fmaf(a.w, b.w, acc);
}
__device__ __forceinline__ void axpy4(float w, const float4 x, float4& acc) {
    acc.x = fmaf(w, x.x, acc.x); acc.y = fmaf(w, x.y, acc.y); acc.z = fmaf(w, x.z, acc.z); acc.w = fmaf(w, x.w, acc.w);
}
// bf16 image img[r][e] (pitch XS_PC) of rows r < 32 of X, columns e0 .. e0 + ec (rows beyond T: zero), in two steps so that a
// chunk's loads can be in flight while something else runs: wave w owns rows w, w + 8, w + 16, w + 24, lane = 16-byte chunk
constexpr int XS_NPRE = XS_T / 8;
static_assert(XS_EC / 4 <= 64, "one lane per 16-byte chunk of a staged row");
struct XsPre { float4 v[XS_NPRE]; };
__device__ __forceinline__ void xs_stage_load(int T, int e0, int ec, const float* __restrict__ X, size_t ldx, XsPre& p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int u = 0; u < XS_NPRE; ++u) {
        const int r = wave + 8 * u;
        p.v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < T && 4 * lane < ec) p.v[u] = reinterpret_cast<const float4*>(X + (size_t)r * ldx + e0)[lane];
    }
}
__device__ __forceinline__ void xs_stage_store(int ec, const XsPre& p, bf16_t* img) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (4 * lane >= ec) return;
#pragma unroll
    for (int u = 0; u < XS_NPRE; ++u) {
        const bf16x4 hv = {(bf16_t)p.v[u].x, (bf16_t)p.v[u].y, (bf16_t)p.v[u].z, (bf16_t)p.v[u].w};
        *reinterpret_cast<bf16x4*>(img + (wave + 8 * u) * XS_PC + 4 * lane) = hv;
    }
}
// ---- operands that are themselves a C-column product (C <= 16): the projected queries Q = Y W_Q^T + b_q and the context
// gradient dO = ddelta W_O of MMF_XAttn_Add.  As launches of their own they are 6.3 MB written and read back three times on the
// serial section of the step; here the 32 x C input tile sits in LDS and a thread that owns four columns forms them for a group
// of rows straight into the bf16 image (fp32 FMAs, one rounding: the values the separate launch + staging would give).
struct XsGen {
    const float *Y, *WQ, *bq;      // Y (B*T, C), WQ (d, C), bq (d)
    const float *dd, *WO;          // backward: ddelta (B*T, C), WO (C, d)
    int C;                         // 0: Q / dO are read from memory
};
// img[r][icol0 + e] (pitch) for e < ec, r < 32:  WROWS: bias[ecol0 + e] + sum_k tile[r][k] W[(ecol0 + e) * C + k];
// else sum_k tile[r][k] W[k * ldw + ecol0 + e].  A lane owns one 4-column chunk (its W values live in registers), the waves
// that share a 64-chunk column group split the 32 rows between them (ec <= 768: 1, 2 or 3 groups -> 8, 4 or 2 waves each).
// In two steps, so that W's loads are issued together with every other operand's (a first version went through three
// dependent global round trips -- input tile, staged operand, W: 9 of the forward's 13 us).
template <int C4> struct XsGenRegs { float4 w[4 * C4]; float4 bv; int c, r0, nrow; bool on; };
template <int C4, bool WROWS>
__device__ __forceinline__ void xs_gen_load(const float* __restrict__ W, size_t ldw, const float* __restrict__ bias, int ecol0, int ec,
                                            XsGenRegs<C4>& g) {
    constexpr int C = 4 * C4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n4 = ec >> 2, ncw = (n4 + 63) >> 6;
    const int sh = ncw == 1 ? 3 : ncw == 2 ? 2 : 1;         // log2 of the waves per column group
    const int cw = wave >> sh;
    g.c = cw * 64 + lane;
    g.nrow = XS_T >> sh;
    g.r0 = (wave & ((1 << sh) - 1)) * g.nrow;
    g.on = cw < ncw && g.c < n4;
    const int e = ecol0 + 4 * (g.on ? g.c : 0);
#pragma unroll
    for (int q = 0; q < C; ++q)          // WROWS: w[j * C4 + k4] = W[e + j][4 k4 ..]; else w[k] = W[k][e ..]
        g.w[q] = WROWS ? reinterpret_cast<const float4*>(W + (size_t)(e + q / C4) * C)[q % C4] : *reinterpret_cast<const float4*>(W + (size_t)q * ldw + e);
    g.bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) g.bv = *reinterpret_cast<const float4*>(bias + e);
}
template <int C4, bool WROWS>
__device__ __forceinline__ void xs_gen_compute(const XsGenRegs<C4>& g, const float* tile, bf16_t* img, int pitch, int icol0) {
    constexpr int C = 4 * C4;
    if (!g.on) return;
#pragma unroll 4
    for (int r = g.r0; r < g.r0 + g.nrow; ++r) {
        float4 y[C4];
#pragma unroll
        for (int k4 = 0; k4 < C4; ++k4) y[k4] = *reinterpret_cast<const float4*>(tile + r * C + 4 * k4);
        float4 o = g.bv;
        if (WROWS) {
#pragma unroll
            for (int k4 = 0; k4 < C4; ++k4) {
                o.x = dot4(y[k4], g.w[0 * C4 + k4], o.x); o.y = dot4(y[k4], g.w[1 * C4 + k4], o.y);
                o.z = dot4(y[k4], g.w[2 * C4 + k4], o.z); o.w = dot4(y[k4], g.w[3 * C4 + k4], o.w);
            }
        } else {
#pragma unroll
            for (int k4 = 0; k4 < C4; ++k4) {
                axpy4(y[k4].x, g.w[4 * k4], o); axpy4(y[k4].y, g.w[4 * k4 + 1], o); axpy4(y[k4].z, g.w[4 * k4 + 2], o);
                axpy4(y[k4].w, g.w[4 * k4 + 3], o);
            }
        }
        const bf16x4 hv = {(bf16_t)o.x, (bf16_t)o.y, (bf16_t)o.z, (bf16_t)o.w};
        *reinterpret_cast<bf16x4*>(img + r * pitch + icol0 + 4 * g.c) = hv;
    }
}
// the forward's operand stage with a generated Q: every global load is issued before the first use (one round trip)
template <int C4>
__device__ __forceinline__ void xs_fwd_stage_gen(int T, int hd, int b, int hcol0, const XsGen& gn, const float* __restrict__ Krows, size_t ldk,
                                                 const XsPre& pre, int ec, float* ytile, bf16_t* imX, bf16_t* imY, int pf, bf16_t* img) {
    const int C = 4 * C4, x = threadIdx.x;
    const float yv = x < T * C ? gn.Y[(size_t)b * T * C + x] : 0.f;          // 32 * C <= 512 threads
    float4 vk[4][XS_SU];
    xs_stage_rows(T, hd, Krows, ldk, vk);
    XsGenRegs<C4> gq;
    xs_gen_load<C4, true>(gn.WQ, 0, gn.bq, hcol0, hd, gq);
    if (x < XS_T * C) ytile[x] = yv;
    xs_store_rows(hd, vk, imY, pf);
    xs_stage_store(ec, pre, img);
    __syncthreads();
    xs_gen_compute<C4, true>(gq, ytile, imX, pf, 0);
}
// the backward's: dO (whole) and Q's chunk generated, V staged, K's chunk from its prefetch registers
template <int C4>
__device__ __forceinline__ void xs_bwd_stage_gen(int T, int hd, int b, int hcol0, int e0, int ec, const XsGen& gn, size_t ldo,
                                                 const float* __restrict__ Vrows, size_t ldk, const XsPre& pK, float* ytile, float* dtile,
                                                 bf16_t* imG, bf16_t* imV, int pf, bf16_t* imK, bf16_t* imQ) {
    const int C = 4 * C4, x = threadIdx.x;
    const float yv = x < T * C ? gn.Y[(size_t)b * T * C + x] : 0.f;
    const float dv = x < T * C ? gn.dd[(size_t)b * T * C + x] : 0.f;
    float4 vv[4][XS_SU];
    xs_stage_rows(T, hd, Vrows, ldk, vv);
    XsGenRegs<C4> go, gq;
    xs_gen_load<C4, false>(gn.WO, ldo, nullptr, hcol0, hd, go);
    xs_gen_load<C4, true>(gn.WQ, 0, gn.bq, hcol0 + e0, ec, gq);
    if (x < XS_T * C) { ytile[x] = yv; dtile[x] = dv; }
    xs_store_rows(hd, vv, imV, pf);
    xs_stage_store(ec, pK, imK);
    __syncthreads();
    xs_gen_compute<C4, false>(go, dtile, imG, pf, 0);
    xs_gen_compute<C4, true>(gq, ytile, imQ, XS_PC, 0);
}
__device__ __forceinline__ void xs_scores_lds(int hd, const bf16_t* imX, const bf16_t* imY, int pf, float alpha, float* Sf, float* Sf2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4, mt = (wave & 3) >> 1, nt = wave & 1;
    const int half = wave >> 2, ksteps = hd >> 5, ks0 = (ksteps + 1) >> 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int kend = (half ? ksteps : ks0) * 32;
#pragma unroll 4
    for (int k = (half ? ks0 : 0) * 32; k < kend; k += 32)
        acc = xs_mfma(xs_frag_row(imX, pf, mt * 16, k, fr, fq), xs_frag_row(imY, pf, nt * 16, k, fr, fq), acc);
    float* out = half ? Sf2 : Sf;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(mt * 16 + fq * 4 + r) * XS_PS + nt * 16 + fr] = alpha * acc[r];
}
// Z[m][e0 + n] = sum_k W(m, k) img[k][n] for the staged chunk: 2 x ec/16 tiles dealt to the eight waves.  WT: W is read through
// the hardware transpose (W(m, k) = tile[k][m]) instead of row-major (tile[m][k]).  Rows m >= T are not written.
// The product is formed transposed (image fragment as the first operand): a lane then holds four consecutive columns of one
// row and stores 16 bytes (8 as bf16) instead of four scattered scalars.
template <bool WT>
__device__ __forceinline__ void xs_mix(int T, int e0, int ec, const bf16_t* Wt, const bf16_t* img, int pimg, float* __restrict__ Z,
                                       size_t ldz, bf16_t* __restrict__ Zh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const bf16x8 w0 = WT ? xs_frag_kmajor(Wt, XS_PT, 0, 0, fr, fq) : xs_frag_row(Wt, XS_PT, 0, 0, fr, fq);
    const bf16x8 w1 = WT ? xs_frag_kmajor(Wt, XS_PT, 16, 0, fr, fq) : xs_frag_row(Wt, XS_PT, 16, 0, fr, fq);
    for (int nt = wave; nt < (ec >> 4); nt += 8) {
        const bf16x8 bfrag = xs_frag_kmajor(img, pimg, nt * 16, 0, fr, fq);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 c0 = xs_mfma(bfrag, w0, z), c1 = xs_mfma(bfrag, w1, z);      // [column 4 fq + r][row fr (+ 16)]
        const size_t col = (size_t)e0 + nt * 16 + fq * 4;
        if (fr < T) {
            *reinterpret_cast<float4*>(Z + (size_t)fr * ldz + col) = make_float4(c0[0], c0[1], c0[2], c0[3]);
            if (Zh) *reinterpret_cast<bf16x4*>(Zh + (size_t)fr * ldz + col) = bf16x4{(bf16_t)c0[0], (bf16_t)c0[1], (bf16_t)c0[2], (bf16_t)c0[3]};
        }
        if (16 + fr < T) {
            *reinterpret_cast<float4*>(Z + (size_t)(16 + fr) * ldz + col) = make_float4(c1[0], c1[1], c1[2], c1[3]);
            if (Zh) *reinterpret_cast<bf16x4*>(Zh + (size_t)(16 + fr) * ldz + col) = bf16x4{(bf16_t)c1[0], (bf16_t)c1[1], (bf16_t)c1[2], (bf16_t)c1[3]};
        }
    }
}
// a further chunk of the backward (a workgroup that walks the chunks): K's chunk from its prefetch registers, Q's chunk generated
template <int C4>
__device__ __forceinline__ void xs_bwd_next_gen(int hcol, int ec, const XsGen& gn, const XsPre& nK, const float* ytile, bf16_t* imK, bf16_t* imQ) {
    XsGenRegs<C4> gq;
    xs_gen_load<C4, true>(gn.WQ, 0, gn.bq, hcol, ec, gq);
    __syncthreads();        // the previous chunk's images have been read
    xs_stage_store(ec, nK, imK);
    xs_gen_compute<C4, true>(gq, ytile, imQ, XS_PC, 0);
}
__device__ __forceinline__ float xs_sum8(float v) {       // over the 8 lanes that share a row
    return group_sum(v, 8);
}
__device__ __forceinline__ float xs_max8(float v) {
    return group8_max(v);
}

__global__ __launch_bounds__(512) void xattn_tile_fwd_kernel(XSmallDims dm, const float* __restrict__ Q, const float* __restrict__ KV,
                                                              const unsigned char* __restrict__ live, float scale, DropCfg drop,
                                                              uint64_t site, float* __restrict__ Pm, float* __restrict__ Am,
                                                              float* __restrict__ O, XsGen gn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xs_smem[];
    float* Sf = reinterpret_cast<float*>(xs_smem);
    float* Sf2 = Sf + XS_T * XS_PS;
    float* ytile = Sf2 + XS_T * XS_PS;           // [32][C] input tile of a generated operand (XsGen)
    bf16_t* Ab = reinterpret_cast<bf16_t*>(ytile + XS_T * XS_GC);
    bf16_t* img = Ab + XS_T * XS_PT;
    bf16_t* imX = img + XS_T * XS_PC;           // staged score operands (xs_staged(hd) only)
    const int pf = dm.hd + 8;
    bf16_t* imY = imX + XS_T * pf;
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, T = dm.T;
    const size_t q0 = (size_t)b * T * dm.d + (size_t)h * dm.hd, k0 = (size_t)b * T * 2 * dm.d + (size_t)h * dm.hd;
    const size_t ldk = (size_t)2 * dm.d, pg = (size_t)(b * dm.H + h) * T * T;
    if (live && !live[b]) {        // window without text: zero attention rows, zero context
        if (blockIdx.z == 0)
            for (int x = tid; x < T * T; x += 512) { Pm[pg + x] = 0.f; Am[pg + x] = 0.f; }
        for (int z0 = blockIdx.z * XS_EC; z0 < dm.hd; z0 += gridDim.z * XS_EC) {
            const int n4 = min(XS_EC, dm.hd - z0) >> 2;
            for (int x = tid; x < T * n4; x += 512)
                reinterpret_cast<float4*>(O + q0 + (size_t)(x / n4) * dm.d + z0)[x % n4] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    // blockIdx.z, + gridDim.z, ...: the 256-column chunks of the head dimension whose part of O this workgroup produces.  Few windows:
    // one chunk per workgroup (every chunk's workgroup forms the full score tile: its operands are L2 hits for all but the first, and
    // the chunks run on different CUs).  Many windows (gridDim.z = 1): the workgroup forms the score tile once and walks the chunks.
    int e0 = blockIdx.z * XS_EC, ec = min(XS_EC, dm.hd - e0);
    XsPre pre;
    xs_stage_load(T, e0, ec, KV + k0 + dm.d, ldk, pre);        // V's chunk travels beside the score operands
    if (xs_staged(dm.hd)) {
        if (gn.C) {       // Q is formed here from the C-column input (host: only with the staged path)
            switch (gn.C >> 2) {
            case 1: xs_fwd_stage_gen<1>(T, dm.hd, b, h * dm.hd, gn, KV + k0, ldk, pre, ec, ytile, imX, imY, pf, img); break;
            case 2: xs_fwd_stage_gen<2>(T, dm.hd, b, h * dm.hd, gn, KV + k0, ldk, pre, ec, ytile, imX, imY, pf, img); break;
            case 3: xs_fwd_stage_gen<3>(T, dm.hd, b, h * dm.hd, gn, KV + k0, ldk, pre, ec, ytile, imX, imY, pf, img); break;
            default: xs_fwd_stage_gen<4>(T, dm.hd, b, h * dm.hd, gn, KV + k0, ldk, pre, ec, ytile, imX, imY, pf, img); break;
            }
        } else {
            xs_stage_pair(T, dm.hd, Q + q0, dm.d, KV + k0, ldk, imX, imY, pf);
            xs_stage_store(ec, pre, img);
        }
        __syncthreads();
        xs_scores_lds(dm.hd, imX, imY, pf, scale, Sf, Sf2);
    } else {
        xs_scores<6>(T, dm.hd, Q + q0, dm.d, KV + k0, ldk, scale, Sf, Sf2);
        xs_stage_store(ec, pre, img);
    }
    __syncthreads();
    if (tid < 256) {   // softmax + dropout: 8 lanes per row, 4 columns each
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        float sv[4], m = -INFINITY, sum = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sv[q] = (i < T && j0 + q < T) ? Sf[i * XS_PS + j0 + q] + Sf2[i * XS_PS + j0 + q] : -INFINITY;
            m = fmaxf(m, sv[q]);
        }
        m = xs_max8(m);
#pragma unroll
        for (int q = 0; q < 4; ++q) { sv[q] = (i < T && j0 + q < T) ? expf(sv[q] - m) : 0.f; sum += sv[q]; }
        sum = xs_sum8(sum);
        const float inv = 1.f / sum;
        const uint64_t row = ((uint64_t)b * dm.H + h) * T + i;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float av = 0.f;
            if (i < T && j0 + q < T) {
                const float pv = sv[q] * inv;
                av = pv * dropout_scale(drop, site, row * T + j0 + q);
                if (blockIdx.z == 0) {
                    Pm[pg + (size_t)i * T + j0 + q] = pv;
                    Am[pg + (size_t)i * T + j0 + q] = av;
                }
            }
            Ab[i * XS_PT + j0 + q] = (bf16_t)av;
        }
    }
    __syncthreads();
    xs_mix<false>(T, e0, ec, Ab, img, XS_PC, O + q0, dm.d, nullptr);       // O = A V
    for (e0 += gridDim.z * XS_EC; e0 < dm.hd; e0 += gridDim.z * XS_EC) {
        ec = min(XS_EC, dm.hd - e0);
        XsPre nx;
        xs_stage_load(T, e0, ec, KV + k0 + dm.d, ldk, nx);
        __syncthreads();        // the previous chunk's image has been read
        xs_stage_store(ec, nx, img);
        __syncthreads();
        xs_mix<false>(T, e0, ec, Ab, img, XS_PC, O + q0, dm.d, nullptr);
    }
}

// dO -> dQ, (dK | dV) (+ optional bf16 image)
__global__ __launch_bounds__(512) void xattn_tile_bwd_kernel(XSmallDims dm, const float* __restrict__ Q, const float* __restrict__ KV,
                                                              const float* __restrict__ dO, const float* __restrict__ Pm,
                                                              const float* __restrict__ Am, const unsigned char* __restrict__ live,
                                                              float scale, DropCfg drop, uint64_t site, float* __restrict__ dQ,
                                                              float* __restrict__ dKV, bf16_t* __restrict__ dKV_h, XsGen gn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xs_smem[];
    float* Sf = reinterpret_cast<float*>(xs_smem);
    float* Sf2 = Sf + XS_T * XS_PS;
    float* ytile = Sf2 + XS_T * XS_PS;           // [32][C] input tiles of the generated operands (XsGen): Y, ddelta
    float* dtile = ytile + XS_T * XS_GC;
    bf16_t* Ab = reinterpret_cast<bf16_t*>(dtile + XS_T * XS_GC);
    bf16_t* dSb = Ab + XS_T * XS_PT;
    bf16_t* imK = dSb + XS_T * XS_PT;
    bf16_t* imQ = imK + XS_T * XS_PC;
    bf16_t* imG = imQ + XS_T * XS_PC;           // dO: its chunk (pitch XS_PC), or staged whole beside V (pitch hd + 8: xs_staged(hd))
    const bool staged = xs_staged(dm.hd);
    const int pf = dm.hd + 8, pg_img = staged ? pf : XS_PC;
    bf16_t* imV = imG + XS_T * pf;
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, T = dm.T;
    const size_t q0 = (size_t)b * T * dm.d + (size_t)h * dm.hd, k0 = (size_t)b * T * 2 * dm.d + (size_t)h * dm.hd;
    const size_t ldk = (size_t)2 * dm.d, pg = (size_t)(b * dm.H + h) * T * T;
    if (live && !live[b]) {        // dO, A and P are zero there: so is every gradient
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const bf16x4 zh = {0, 0, 0, 0};
        for (int z0 = blockIdx.z * XS_EC; z0 < dm.hd; z0 += gridDim.z * XS_EC) {
        const int n4 = min(XS_EC, dm.hd - z0) >> 2;
        for (int x = tid; x < T * n4; x += 512) {
            const int i = x / n4, c = x - i * n4;
            reinterpret_cast<float4*>(dQ + q0 + (size_t)i * dm.d + z0)[c] = z4;
            reinterpret_cast<float4*>(dKV + k0 + (size_t)i * ldk + z0)[c] = z4;
            reinterpret_cast<float4*>(dKV + k0 + dm.d + (size_t)i * ldk + z0)[c] = z4;
            if (dKV_h) {
                reinterpret_cast<bf16x4*>(dKV_h + k0 + (size_t)i * ldk + z0)[c] = zh;
                reinterpret_cast<bf16x4*>(dKV_h + k0 + dm.d + (size_t)i * ldk + z0)[c] = zh;
            }
        }
        }
        return;
    }
    int e0 = blockIdx.z * XS_EC, ec = min(XS_EC, dm.hd - e0);       // this workgroup's chunks of dQ, dK, dV: e0, e0 + gridDim.z * 256, ... (see the forward)
    XsPre pK, pG, pQ;
    xs_stage_load(T, e0, ec, KV + k0, ldk, pK);           // the chunk's images travel beside dA's operands
    if (staged && gn.C) {      // dO and Q's chunk are formed here from their C-column inputs (host: only with the staged path)
        switch (gn.C >> 2) {
        case 1: xs_bwd_stage_gen<1>(T, dm.hd, b, h * dm.hd, e0, ec, gn, dm.d, KV + k0 + dm.d, ldk, pK, ytile, dtile, imG, imV, pf, imK, imQ); break;
        case 2: xs_bwd_stage_gen<2>(T, dm.hd, b, h * dm.hd, e0, ec, gn, dm.d, KV + k0 + dm.d, ldk, pK, ytile, dtile, imG, imV, pf, imK, imQ); break;
        case 3: xs_bwd_stage_gen<3>(T, dm.hd, b, h * dm.hd, e0, ec, gn, dm.d, KV + k0 + dm.d, ldk, pK, ytile, dtile, imG, imV, pf, imK, imQ); break;
        default: xs_bwd_stage_gen<4>(T, dm.hd, b, h * dm.hd, e0, ec, gn, dm.d, KV + k0 + dm.d, ldk, pK, ytile, dtile, imG, imV, pf, imK, imQ); break;
        }
        __syncthreads();
        xs_scores_lds(dm.hd, imG, imV, pf, 1.f, Sf, Sf2);                            // dA = dO V^T
    } else if (staged) {
        xs_stage_load(T, e0, ec, Q + q0, dm.d, pQ);
        xs_stage_pair(T, dm.hd, dO + q0, dm.d, KV + k0 + dm.d, ldk, imG, imV, pf);
        xs_stage_store(ec, pK, imK);
        xs_stage_store(ec, pQ, imQ);
        __syncthreads();
        xs_scores_lds(dm.hd, imG, imV, pf, 1.f, Sf, Sf2);                            // dA = dO V^T
    } else {
        xs_stage_load(T, e0, ec, Q + q0, dm.d, pQ);
        xs_stage_load(T, e0, ec, dO + q0, dm.d, pG);
        xs_scores<6>(T, dm.hd, dO + q0, dm.d, KV + k0 + dm.d, ldk, 1.f, Sf, Sf2);
        xs_stage_store(ec, pK, imK);
        xs_stage_store(ec, pQ, imQ);
        xs_stage_store(ec, pG, imG);
    }
    __syncthreads();
    if (tid < 256) {   // dS = scale P (dA dropscale - sum_j P dA dropscale): 8 lanes per row, 4 columns each
        const int i = tid >> 3, j0 = (tid & 7) * 4;
        const uint64_t row = ((uint64_t)b * dm.H + h) * T + i;
        float pv[4], dp[4], av[4], dot = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = i < T && j0 + q < T;
            pv[q] = in ? Pm[pg + (size_t)i * T + j0 + q] : 0.f;
            av[q] = in ? Am[pg + (size_t)i * T + j0 + q] : 0.f;
            dp[q] = in ? (Sf[i * XS_PS + j0 + q] + Sf2[i * XS_PS + j0 + q]) * dropout_scale(drop, site, row * T + j0 + q) : 0.f;
            dot = fmaf(pv[q], dp[q], dot);
        }
        dot = xs_sum8(dot);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dSb[i * XS_PT + j0 + q] = (bf16_t)(scale * pv[q] * (dp[q] - dot));      // scale folded in: dQ and dK both carry it
            Ab[i * XS_PT + j0 + q] = (bf16_t)av[q];
        }
    }
    __syncthreads();
    xs_mix<false>(T, e0, ec, dSb, imK, XS_PC, dQ + q0, dm.d, nullptr);                                                      // dQ = dS K
    xs_mix<true>(T, e0, ec, Ab, staged ? imG + e0 : imG, pg_img, dKV + k0 + dm.d, ldk, dKV_h ? dKV_h + k0 + dm.d : nullptr);   // dV = A^T dO
    xs_mix<true>(T, e0, ec, dSb, imQ, XS_PC, dKV + k0, ldk, dKV_h ? dKV_h + k0 : nullptr);                                // dK = dS^T Q
    for (e0 += gridDim.z * XS_EC; e0 < dm.hd; e0 += gridDim.z * XS_EC) {       // many windows: dA / dS were formed once, the chunks follow
        ec = min(XS_EC, dm.hd - e0);
        XsPre nK, nQ, nG;
        xs_stage_load(T, e0, ec, KV + k0, ldk, nK);
        if (staged && gn.C) {
            switch (gn.C >> 2) {
            case 1: xs_bwd_next_gen<1>(h * dm.hd + e0, ec, gn, nK, ytile, imK, imQ); break;
            case 2: xs_bwd_next_gen<2>(h * dm.hd + e0, ec, gn, nK, ytile, imK, imQ); break;
            case 3: xs_bwd_next_gen<3>(h * dm.hd + e0, ec, gn, nK, ytile, imK, imQ); break;
            default: xs_bwd_next_gen<4>(h * dm.hd + e0, ec, gn, nK, ytile, imK, imQ); break;
            }
        } else {
            xs_stage_load(T, e0, ec, Q + q0, dm.d, nQ);
            if (!staged) xs_stage_load(T, e0, ec, dO + q0, dm.d, nG);
            __syncthreads();
            xs_stage_store(ec, nK, imK);
            xs_stage_store(ec, nQ, imQ);
            if (!staged) xs_stage_store(ec, nG, imG);
        }
        __syncthreads();
        xs_mix<false>(T, e0, ec, dSb, imK, XS_PC, dQ + q0, dm.d, nullptr);
        xs_mix<true>(T, e0, ec, Ab, staged ? imG + e0 : imG, pg_img, dKV + k0 + dm.d, ldk, dKV_h ? dKV_h + k0 + dm.d : nullptr);
        xs_mix<true>(T, e0, ec, dSb, imQ, XS_PC, dKV + k0, ldk, dKV_h ? dKV_h + k0 : nullptr);
    }
}
inline size_t xs_fwd_lds(int hd) {
    return (size_t)2 * XS_T * XS_PS * 4 + (size_t)XS_T * XS_GC * 4 + (size_t)XS_T * XS_PT * 2 + (size_t)XS_T * XS_PC * 2 +
           (xs_staged(hd) ? (size_t)2 * XS_T * (hd + 8) * 2 : 0);
}
inline size_t xs_bwd_lds(int hd) {
    return (size_t)2 * XS_T * XS_PS * 4 + (size_t)2 * XS_T * XS_GC * 4 + (size_t)2 * XS_T * XS_PT * 2 + (size_t)2 * XS_T * XS_PC * 2 +
           (xs_staged(hd) ? (size_t)2 * XS_T * (hd + 8) * 2 : (size_t)XS_T * XS_PC * 2);
}

}  // namespace

size_t ragged_attn_dp_floats(int B, int N, int H, int hd) {
    return (size_t)B * N * H * (N > RAGGED_SPLIT_N ? 1 : cdiv(hd, 64));
}
size_t ragged_attn_part_floats(int B, int T, int d, int N) {
    return N > RAGGED_SPLIT_N ? (size_t)B * cdiv(N, RAGGED_CH) * T * d : 0;
}

template <typename KT>
static int ragged_attn_fwd_impl(RaggedAttnDims dm, const int* offsets, const int* rowmap, const KT* KVp, const float* qs,
                                float* P, float* ctx, DropCfg drop, uint64_t site, hipStream_t s, void* ctx_h, float* part) {
    if (dm.B <= 0) return IMMTSF_OK;
    bf16_t* ch = static_cast<bf16_t*>(ctx_h);
    if (dm.N > RAGGED_SPLIT_N) {      // long windows: chunked (see the comment above the kernels)
        if (!part) return IMMTSF_EWORKSPACE;
        const int maxch = cdiv(dm.N, RAGGED_CH);
        const size_t lds = (size_t)(RAGGED_CH + TT * 64 + 16) * sizeof(float);
        hipLaunchKernelGGL((ragged_scores_kernel<KT>), dim3(cdiv(dm.B * dm.N, 4), dm.H), dim3(256), 0, s, dm, offsets, KVp, qs, P);
        IMMTSF_LAUNCH_CHECK();
        hipLaunchKernelGGL(ragged_softmax_kernel, dim3(dm.B, dm.H), dim3(256), 0, s, dm, offsets, P);
        IMMTSF_LAUNCH_CHECK();
        hipLaunchKernelGGL((ragged_attn_fwd_kernel<true, KT>), dim3(dm.B * maxch, dm.H, cdiv(dm.hd, 256)), dim3(256), lds, s, dm, offsets, rowmap,
                           KVp, qs, P, ctx, drop, site, ch, part, maxch);
        IMMTSF_LAUNCH_CHECK();
        hipLaunchKernelGGL(ragged_ctx_reduce_kernel, dim3(dm.B, cdiv(dm.T * dm.H * dm.hd, 256)), dim3(256), 0, s, dm, offsets, part, maxch,
                           ctx, ch);
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    {   // the fusion's own regime: short windows, one round of loads (ragged_attn_fwd_short_kernel); IMMTSF_RAGGED_SHORT=0: off
        constexpr bool on = true;
        constexpr int W = KVec<KT>::W;
        if (on && dm.N <= 64 && dm.T <= TT && (dm.hd % W) == 0 && dm.hd / W <= 64 * RS_PCS && (reinterpret_cast<uintptr_t>(KVp) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(qs) & 15) == 0) {
            const dim3 grid(dm.B, dm.H, cdiv(dm.hd, 256));
            if (dm.N <= 32)
                hipLaunchKernelGGL((ragged_attn_fwd_short_kernel<KT, 32>), grid, dim3(256), 0, s, dm, offsets, rowmap, KVp, qs, P, ctx, drop, site, ch);
            else
                hipLaunchKernelGGL((ragged_attn_fwd_short_kernel<KT, 64>), grid, dim3(256), 0, s, dm, offsets, rowmap, KVp, qs, P, ctx, drop, site, ch);
            IMMTSF_LAUNCH_CHECK();
            return IMMTSF_OK;
        }
    }
    const size_t lds = (size_t)(dm.N + TT * 64 + 16) * sizeof(float);
    if (lds > 160 * 1024) return IMMTSF_EUNSUPPORTED;
    hipLaunchKernelGGL((ragged_attn_fwd_kernel<false, KT>), dim3(dm.B, dm.H, cdiv(dm.hd, 256)), dim3(256), lds, s, dm, offsets, rowmap, KVp, qs,
                       P, ctx, drop, site, ch, nullptr, 1);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

template <typename KT>
static int ragged_attn_bwd_impl(RaggedAttnDims dm, const int* offsets, const int* rowmap, const KT* KVp, const float* qs,
                                const float* P, const float* dctx, float* dKVp, float* dqs_part, float* dp_buf, DropCfg drop,
                                uint64_t site, hipStream_t s, void* dKVp_h) {
    if (dm.B <= 0) return IMMTSF_OK;
    if (drop.p > 0.f && dm.T > MT * 64) return IMMTSF_EUNSUPPORTED;
    const int maxch = dm.N > RAGGED_SPLIT_N ? cdiv(dm.N, RAGGED_CH) : 1;
    const size_t lds = (size_t)((maxch > 1 ? RAGGED_CH : dm.N) + 16) * sizeof(float);
    if (lds > 160 * 1024) return IMMTSF_EUNSUPPORTED;
    // short windows: dp as one slab per 64-column slice (dp_buf holds ragged_attn_dp_floats): no zero-fill, no atomics
    const size_t dp_stride = maxch > 1 ? 0 : (size_t)dm.B * dm.N * dm.H;
    const int dp_slabs = maxch > 1 ? 1 : cdiv(dm.hd, 64);
    if (maxch > 1) {
        // (fill kernels, not hipMemsetAsync: a memset node captured into a hipGraph did not zero the buffer again on the second and later
        // replays on ROCm 7.2 -- tests/test_gpu_train.py::test_split_k_gemm_zero_fill_survives_graph_replay; every zero-fill of this library is a kernel)
        if (int rc = launch_fill(dp_buf, 0.f, (size_t)dm.B * dm.N * dm.H, s)) return rc;
        if (int rc = launch_fill(dqs_part, 0.f, (size_t)dm.B * dm.H * dm.hd, s)) return rc;
    }
    const size_t lds_long = ((drop.p > 0.f ? (size_t)(dm.T <= 32 ? 32 : ((dm.T + 3) & ~3)) * RAGGED_CH : 0) + 4 * RAGGED_CH) * sizeof(float);
    if (maxch > 1 && lds_long <= 128 * 1024) {
        if (lds_long > 64 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ragged_attn_bwd_dv_long_kernel<true, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_long);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ragged_attn_bwd_dv_long_kernel<false, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_long);
        }
        const dim3 grid(dm.B * maxch, dm.H, cdiv(dm.hd, 256));
        if (dm.T <= 32)
            hipLaunchKernelGGL((ragged_attn_bwd_dv_long_kernel<true, KT>), grid, dim3(256), lds_long, s, dm, offsets, rowmap, KVp, P, dctx, dKVp,
                               dp_buf, drop, site, static_cast<bf16_t*>(dKVp_h), maxch);
        else
            hipLaunchKernelGGL((ragged_attn_bwd_dv_long_kernel<false, KT>), grid, dim3(256), lds_long, s, dm, offsets, rowmap, KVp, P, dctx, dKVp,
                               dp_buf, drop, site, static_cast<bf16_t*>(dKVp_h), maxch);
    } else {
        hipLaunchKernelGGL((ragged_attn_bwd_dv_kernel<KT>), dim3(dm.B * maxch, dm.H, cdiv(dm.hd, 64)), dim3(256), 0, s, dm, offsets, rowmap, KVp, P,
                           dctx, dKVp, dp_buf, drop, site, static_cast<bf16_t*>(dKVp_h), maxch, dp_stride);
    }
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL((ragged_attn_bwd_ds_kernel<KT>), dim3(dm.B * maxch, dm.H, cdiv(dm.hd, 256)), dim3(256), lds, s, dm, offsets, KVp, qs, P,
                       dp_buf, dKVp, dqs_part, static_cast<bf16_t*>(dKVp_h), maxch, dp_slabs, dp_stride);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// KVp_h != null: the K | V rows are read from that bf16 image (same layout) instead of KVp
int launch_ragged_attn_fwd(RaggedAttnDims dm, const int* offsets, const int* rowmap, const float* KVp, const float* qs,
                           float* P, float* ctx, DropCfg drop, uint64_t site, hipStream_t s, void* ctx_h, float* part, const void* KVp_h) {
    if (KVp_h) return ragged_attn_fwd_impl(dm, offsets, rowmap, static_cast<const bf16_t*>(KVp_h), qs, P, ctx, drop, site, s, ctx_h, part);
    return ragged_attn_fwd_impl(dm, offsets, rowmap, KVp, qs, P, ctx, drop, site, s, ctx_h, part);
}
int launch_ragged_attn_bwd(RaggedAttnDims dm, const int* offsets, const int* rowmap, const float* KVp, const float* qs,
                           const float* P, const float* dctx, float* dKVp, float* dqs_part, float* dp_buf, DropCfg drop,
                           uint64_t site, hipStream_t s, void* dKVp_h, const void* KVp_h) {
    if (KVp_h)
        return ragged_attn_bwd_impl(dm, offsets, rowmap, static_cast<const bf16_t*>(KVp_h), qs, P, dctx, dKVp, dqs_part, dp_buf, drop, site, s, dKVp_h);
    return ragged_attn_bwd_impl(dm, offsets, rowmap, KVp, qs, P, dctx, dKVp, dqs_part, dp_buf, drop, site, s, dKVp_h);
}

int launch_softmax_rows_fwd(float* sc, float* A, int B, int H, int L, int S, const unsigned char* live, DropCfg drop,
                            uint64_t site, int causal, hipStream_t s) {
    const int rows = B * H * L;
    if (rows <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, sc, A, rows, H * L, S, live, drop, site,
                       causal ? L : 0);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_softmax_rows_bwd(float* dA, const float* P, int B, int H, int L, int S, DropCfg drop, uint64_t site,
                            hipStream_t s) {
    const int rows = B * H * L;
    if (rows <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, dA, P, rows, S, drop, site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// sequences per workgroup of the staged kernels: one thread per (sequence, head, position) within 256 threads, rows_per
// (3 forward, 4 backward) x L x H x E floats per sequence within 48 KB of LDS; a small SB spreads the launch over more CUs
static int short_stage_seqs(int L, int H, int E, int rows_per) {
    int sb = 256 / (H * L);
    const int lds_cap = (48 * 1024) / (rows_per * L * H * E * (int)sizeof(float));
    if (sb > lds_cap) sb = lds_cap;
    if (sb > 16) sb = 16;
    return sb;
}

int launch_attn_short_fwd(const float* qkv, int B, int L, int H, int E, float scale, int causal, DropCfg drop, uint64_t site,
                          float* out, hipStream_t s) {
    if (L > SHORT_L || E > 4 * SHORT_EV || (E & 3) || (reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
        return IMMTSF_EUNSUPPORTED;
    const int n = B * H * L;
    if (n <= 0) return IMMTSF_OK;
    const int SB = short_stage_seqs(L, H, E, 3);
    if (SB >= 4) {
#define IMMTSF_SHORT_FWD_STAGED(SL, SEV)                                                                                        \
        hipLaunchKernelGGL((attn_short_fwd_staged_kernel<SL, SEV>), dim3(cdiv(B, SB)), dim3(256), (size_t)SB * L * 3 * H * E * sizeof(float), \
                           s, ShortDims{B, L, H, E}, qkv, scale, causal, drop, site, out, SB)
        if (L <= 2 && E <= 32) IMMTSF_SHORT_FWD_STAGED(2, 8);
        else if (L <= 4 && E <= 32) IMMTSF_SHORT_FWD_STAGED(4, 8);
        else IMMTSF_SHORT_FWD_STAGED(8, 16);
#undef IMMTSF_SHORT_FWD_STAGED
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    hipLaunchKernelGGL((attn_short_fwd_kernel<8, 16>), dim3(cdiv(n, 256)), dim3(256), 0, s, ShortDims{B, L, H, E}, qkv, scale, causal, drop, site, out);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_attn_short_bwd(const float* qkv, const float* dout, int B, int L, int H, int E, float scale, int causal, DropCfg drop,
                          uint64_t site, float* dqkv, hipStream_t s) {
    if (L > SHORT_L || E > 4 * SHORT_EV || (E & 3) || ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(dout) |
                                                        reinterpret_cast<uintptr_t>(dqkv)) & 15))
        return IMMTSF_EUNSUPPORTED;
    const int n = B * H * L;
    if (n <= 0) return IMMTSF_OK;
    const int SB = short_stage_seqs(L, H, E, 4);
    if (SB >= 4) {
#define IMMTSF_SHORT_BWD_STAGED(SL, SEV)                                                                                        \
        hipLaunchKernelGGL((attn_short_bwd_staged_kernel<SL, SEV>), dim3(cdiv(B, SB)), dim3(256), (size_t)SB * L * 4 * H * E * sizeof(float), \
                           s, ShortDims{B, L, H, E}, qkv, dout, scale, causal, drop, site, dqkv, SB)
        if (L <= 2 && E <= 32) IMMTSF_SHORT_BWD_STAGED(2, 8);
        else if (L <= 4 && E <= 32) IMMTSF_SHORT_BWD_STAGED(4, 8);
        else IMMTSF_SHORT_BWD_STAGED(8, 16);
#undef IMMTSF_SHORT_BWD_STAGED
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    hipLaunchKernelGGL((attn_short_bwd_kernel<8, 16>), dim3(cdiv(n, 256)), dim3(256), 0, s, ShortDims{B, L, H, E}, qkv, dout, scale, causal, drop,
                       site, dqkv);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

bool xattn_small_supported(int T, int H, int hd) { return T >= 1 && T <= XS_T && H >= 1 && hd >= 16 && (hd % 16) == 0; }
bool xattn_small_generates(int hd, int C) { return xs_staged(hd) && C >= 4 && C <= XS_GC && (C % 4) == 0; }
static bool xs_gen_ok(const XattnGen* g, int hd, bool bwd, XsGen* out) {
    *out = XsGen{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (!g || g->C == 0) return true;
    uintptr_t a = reinterpret_cast<uintptr_t>(g->Y) | reinterpret_cast<uintptr_t>(g->WQ) | reinterpret_cast<uintptr_t>(g->bq);
    if (bwd) a |= reinterpret_cast<uintptr_t>(g->dd) | reinterpret_cast<uintptr_t>(g->WO);
    if (!xattn_small_generates(hd, g->C) || !g->Y || !g->WQ || !g->bq || (bwd && (!g->dd || !g->WO)) || (a & 15)) return false;
    *out = XsGen{g->Y, g->WQ, g->bq, g->dd, g->WO, g->C};
    return true;
}

int launch_xattn_small_fwd(const float* Q, const float* KV, const unsigned char* live, int B, int T, int H, int hd, float scale, DropCfg drop,
                           uint64_t site, float* Pm, float* Am, float* O, hipStream_t s, const XattnGen* gen) {
    if (B <= 0) return IMMTSF_OK;
    XsGen gn;
    if (!xattn_small_supported(T, H, hd) || ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(KV) | reinterpret_cast<uintptr_t>(O)) & 15) ||
        !xs_gen_ok(gen, hd, false, &gn) || (!gn.C && !Q))
        return IMMTSF_EINVAL;
    const XSmallDims dm{B, T, H, hd, H * hd};
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_tile_fwd_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)xs_fwd_lds(XS_HD_STAGED));
    if (attr != hipSuccess) return (int)attr;
    // few windows: a workgroup per 256-column chunk (parallelism); many: one per (window, head) that forms the scores once
    const int nz = B * H >= 1024 ? 1 : cdiv(hd, XS_EC);
    hipLaunchKernelGGL(xattn_tile_fwd_kernel, dim3(B, H, nz), dim3(512), xs_fwd_lds(hd), s, dm, Q, KV, live, scale, drop, site, Pm,
                       Am, O, gn);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_xattn_small_bwd(const float* Q, const float* KV, const float* dO, const float* Pm, const float* Am, const unsigned char* live,
                           int B, int T, int H, int hd, float scale, DropCfg drop, uint64_t site, float* dQ, float* dKV, void* dKV_h,
                           hipStream_t s, const XattnGen* gen) {
    if (B <= 0) return IMMTSF_OK;
    XsGen gn;
    if (!xattn_small_supported(T, H, hd) ||
        ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(KV) | reinterpret_cast<uintptr_t>(dO) | reinterpret_cast<uintptr_t>(dQ) |
          reinterpret_cast<uintptr_t>(dKV)) & 15) || (reinterpret_cast<uintptr_t>(dKV_h) & 7) || !xs_gen_ok(gen, hd, true, &gn) ||
        (!gn.C && (!Q || !dO)))
        return IMMTSF_EINVAL;
    const XSmallDims dm{B, T, H, hd, H * hd};
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_tile_bwd_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)xs_bwd_lds(XS_HD_STAGED));
    if (attr != hipSuccess) return (int)attr;
    const int nz = B * H >= 1024 ? 1 : cdiv(hd, XS_EC);
    hipLaunchKernelGGL(xattn_tile_bwd_kernel, dim3(B, H, nz), dim3(512), xs_bwd_lds(hd), s, dm, Q, KV, dO, Pm, Am, live, scale, drop, site, dQ, dKV,
                       static_cast<bf16_t*>(dKV_h), gn);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
