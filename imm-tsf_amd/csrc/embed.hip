// Time-aware patch embedding and data embedding (a13) as ONE gather + skinny-product + positional
// table + dropout kernel per direction -- no padded copy, no unfolded (rows, P, patch_len) tensor, no [x(l-1);x(l);x(l+1)]
// concatenation, no separate add / dropout passes.
//
//   mode 0, PatchEmbedding (reference layers/Embed.py:165-190): x (R, L) rows (R = B * n_vars; PatchTST feeds the
//       (value, mask, time)-interleaved series, models/PatchTST.py:100-105) -> ReplicationPad1d((0, pad)) ->
//       unfold(size = K = patch_len, step = stride) -> Linear(K -> D, no bias) + pe[p] -> dropout:
//           out[r, p, :] = drop( sum_k W[:, k] * x[r, min(p * stride + k, L - 1)] + pe[p, :] )
//   mode 1, TokenEmbedding + PositionalEmbedding (reference :29-42, 109-126): x (R = B, L, c_in), circular 3-tap
//       Conv1d(c_in -> D, no bias), weight (D, c_in, 3) read as (D, K = 3 c_in) with k = c * 3 + t:
//           out[b, l, :] = drop( sum_{c,t} W[:, c, t] * x[b, (l + t - 1) mod L, c] + pe[l, :] )
//
// HBM-bound on the (R, P, D) output (PatchTST cfg3: 384 x 10 x 512 fp32 = 7.9 MB written once; the inputs are 36 KB of
// weights, 20 KB of table and 147 KB of series).  Exact fp32 FMAs in both precision modes (K <= 64: an MFMA tile would
// be mostly padding).  Forward: a workgroup per 16 (row, patch) pairs, one thread per output column with that column's K
// weights in registers, stores coalesced along D.  Backward: dW by workgroups that own 16 output columns and a slice of
// the pairs (register accumulators, LDS sum over the pair lanes, <= 8 atomic adds per element); dx (optional: the series
// are data in every configured backbone) by a wave per (row, patch).
#include "../../include/immtsf.h"
#include "common.hpp"
#include "rowops.hpp"

namespace {

constexpr int EMB_KMAX = 64;

struct EmbDims {
    int mode, R, L, c_in, P, K, stride, D;
};

__device__ __forceinline__ int emb_src(const EmbDims& e, int p, int k) {      // offset of gathered element (p, k) inside row r's data
    if (e.mode == 0) return min(p * e.stride + k, e.L - 1);
    const int c = k / 3, t = k - 3 * c;
    int l = p + t - 1;
    l = l < 0 ? l + e.L : (l >= e.L ? l - e.L : l);
    return l * e.c_in + c;
}

// grid ceil(R P / PB): a workgroup takes PB consecutive (row, patch) pairs; their gathered taps sit in LDS (g[PB][K], read as
// wave-wide broadcasts), every thread keeps its output column's K weights in registers and writes that column of the PB
// outputs, coalesced along D
// KB: compile-time bound on K (16 / 32 / 64), the tap rows in LDS and the weight registers are zero-padded to it: the K loop
// is KB/4 16-byte broadcast reads + KB FMAs with no branches (with a run-time K every FMA waited on its own 4-byte LDS read
// behind a branch: 58 us for PatchTST's 9.4 MB of output)
template <int PB, int KB>
__global__ __launch_bounds__(256) void embed_fwd_kernel(EmbDims e, const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ pe, float* __restrict__ out, DropCfg drop,
                                                         uint64_t site) {
    __shared__ __attribute__((aligned(16))) float g[PB * KB];
    const int tid = threadIdx.x;
    const long pairs = (long)e.R * e.P, q0 = (long)blockIdx.x * PB;
    const int nq = (int)min((long)PB, pairs - q0);
    const size_t row_elems = e.mode == 0 ? (size_t)e.L : (size_t)e.L * e.c_in;
    for (int i = tid; i < PB * KB; i += 256) {
        const int qq = i / KB, k = i - qq * KB;
        float v = 0.f;
        if (qq < nq && k < e.K) {
            const long q = q0 + qq;
            const int r = (int)(q / e.P), p = (int)(q - (long)r * e.P);
            v = x[(size_t)r * row_elems + emb_src(e, p, k)];
        }
        g[i] = v;
    }
    __syncthreads();
    for (int d = tid; d < e.D; d += 256) {
        float w[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) w[k] = k < e.K ? W[(size_t)d * e.K + k] : 0.f;
        for (int qq = 0; qq < nq; ++qq) {
            const long q = q0 + qq;
            const int p = (int)(q % e.P);
            float a = pe[(size_t)p * e.D + d];
            const float4* g4 = reinterpret_cast<const float4*>(g + qq * KB);
#pragma unroll
            for (int k4 = 0; k4 < KB / 4; ++k4) {
                const float4 t = g4[k4];
                a = fmaf(w[4 * k4], t.x, fmaf(w[4 * k4 + 1], t.y, fmaf(w[4 * k4 + 2], t.z, fmaf(w[4 * k4 + 3], t.w, a))));
            }
            const size_t o = (size_t)q * e.D + d;
            out[o] = a * dropout_scale(drop, site, (uint64_t)o);
        }
    }
}

// dW[d, k] = sum over (row, patch) pairs q of dout[q, d] * dropscale * g(q, k): a (D x pairs) x (pairs x K) product with a
// tiny K.  grid (ceil(D / 16), S): a workgroup owns 16 output columns d and one of S slices of the pairs; its 256 threads
// are 16 columns x 16 pair lanes, every thread keeps K accumulators, the 16 pair lanes are summed through LDS and the
// workgroup adds its 16 x K block with one atomic per element (S <= 32 adds per address; S = 1 for short inputs).
// (The first version gave every workgroup all D columns of 16 pairs: 240 workgroups x D K atomics on D K addresses
// serialised in L2 -- 306 us at PatchTST's 512 x 16; this one is ~10 us.)
template <int KB>
__global__ __launch_bounds__(256) void embed_bwd_w_kernel(EmbDims e, const float* __restrict__ x, const float* __restrict__ dout,
                                                           float* __restrict__ dW, DropCfg drop, uint64_t site) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* g = lds;                              // [16 pairs][KB], zero-padded
    float* red = lds + 16 * KB;                  // [16 pair lanes][16 columns][KB + 1]
    const int tid = threadIdx.x, dc = tid & 15, pl = tid >> 4;
    const int d = blockIdx.x * 16 + dc;
    const long pairs = (long)e.R * e.P;
    const long per = ((pairs + gridDim.y - 1) / gridDim.y + 15) & ~15L, q0 = blockIdx.y * per, q1 = min(pairs, q0 + per);
    const size_t row_elems = e.mode == 0 ? (size_t)e.L : (size_t)e.L * e.c_in;
    float acc[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) acc[k] = 0.f;
    for (long qb = q0; qb < q1; qb += 16) {
        __syncthreads();
        for (int i = tid; i < 16 * KB; i += 256) {
            const int qq = i / KB, k = i - qq * KB;
            const long q = qb + qq;
            float v = 0.f;
            if (q < q1 && k < e.K) {
                const int r = (int)(q / e.P), p = (int)(q - (long)r * e.P);
                v = x[(size_t)r * row_elems + emb_src(e, p, k)];
            }
            g[i] = v;
        }
        __syncthreads();
        const long q = qb + pl;
        if (q < q1 && d < e.D) {
            const size_t o = (size_t)q * e.D + d;
            const float dv = dout[o] * dropout_scale(drop, site, (uint64_t)o);
            const float4* g4 = reinterpret_cast<const float4*>(g + pl * KB);
#pragma unroll
            for (int k4 = 0; k4 < KB / 4; ++k4) {
                const float4 t = g4[k4];
                acc[4 * k4] = fmaf(dv, t.x, acc[4 * k4]);
                acc[4 * k4 + 1] = fmaf(dv, t.y, acc[4 * k4 + 1]);
                acc[4 * k4 + 2] = fmaf(dv, t.z, acc[4 * k4 + 2]);
                acc[4 * k4 + 3] = fmaf(dv, t.w, acc[4 * k4 + 3]);
            }
        }
    }
    __syncthreads();
    constexpr int pitch = KB + 1;
#pragma unroll
    for (int k = 0; k < KB; ++k) red[(pl * 16 + dc) * pitch + k] = acc[k];
    __syncthreads();
    for (int i = tid; i < 16 * e.K; i += 256) {
        const int c = i / e.K, k = i - c * e.K;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) a += red[(j * 16 + c) * pitch + k];
        const int dd = blockIdx.x * 16 + c;
        if (dd < e.D) {
            if (gridDim.y == 1) dW[(size_t)dd * e.K + k] = a;       // sole writer: dW need not be zero
            else atomicAdd(dW + (size_t)dd * e.K + k, a);
        }
    }
}

// dx: a wave per (row, patch): v[k] = sum_d dout[r,p,d] * dropscale * W[d,k], scattered onto the source positions
__global__ __launch_bounds__(256) void embed_bwd_x_kernel(EmbDims e, const float* __restrict__ W, const float* __restrict__ dout,
                                                           float* __restrict__ dx, DropCfg drop, uint64_t site) {
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= (long)e.R * e.P) return;
    const int r = (int)(q / e.P), p = (int)(q - (long)r * e.P);
    const size_t row_elems = e.mode == 0 ? (size_t)e.L : (size_t)e.L * e.c_in;
    for (int k = 0; k < e.K; ++k) {
        float a = 0.f;
        for (int d = lane; d < e.D; d += 64) {
            const size_t o = (size_t)q * e.D + d;
            a = fmaf(dout[o] * dropout_scale(drop, site, (uint64_t)o), W[(size_t)d * e.K + k], a);
        }
        a = wave_sum(a);
        if (lane == 0) atomicAdd(dx + (size_t)r * row_elems + emb_src(e, p, k), a);
    }
}

DropCfg mk_drop2(float p, uint64_t seed, const uint64_t* seed_dev) {
    DropCfg d;
    d.seed = seed;
    d.p = p > 0.f ? p : 0.f;
    d.inv_keep = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    d.seed_dev = seed_dev;
    return d;
}

bool emb_bad(const EmbDims& e) {
    if (e.R <= 0 || e.L <= 0 || e.P <= 0 || e.D <= 0 || e.K <= 0 || e.K > EMB_KMAX || e.D > 512) return true;
    if (e.mode == 0) return e.stride <= 0;
    if (e.mode == 1) return e.c_in <= 0 || e.K != 3 * e.c_in || e.P != e.L;
    return true;
}

}  // namespace

extern "C" {

int immtsf_embed_forward(int32_t mode, const float* x, int32_t R, int32_t L, int32_t c_in, int32_t P, int32_t K, int32_t stride,
                         int32_t D, const float* W, const float* pe, float* out, float p_drop, uint64_t seed, uint64_t site,
                         const uint64_t* seed_step_dev, immtsf_stream_t stream) {
    const EmbDims e{mode, R, L, c_in, P, K, stride, D};
    if (!x || !W || !pe || !out) return IMMTSF_EINVAL;
    if (emb_bad(e)) return IMMTSF_EUNSUPPORTED;
    constexpr int PB = 16;
    const long pairs = (long)R * P;
    const dim3 grid((unsigned)((pairs + PB - 1) / PB));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DropCfg drop = mk_drop2(p_drop, seed, seed_step_dev);
    if (K <= 16) hipLaunchKernelGGL((embed_fwd_kernel<PB, 16>), grid, dim3(256), 0, s, e, x, W, pe, out, drop, site);
    else if (K <= 32) hipLaunchKernelGGL((embed_fwd_kernel<PB, 32>), grid, dim3(256), 0, s, e, x, W, pe, out, drop, site);
    else hipLaunchKernelGGL((embed_fwd_kernel<PB, 64>), grid, dim3(256), 0, s, e, x, W, pe, out, drop, site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

/* dW (D, K) must read zero on entry unless dw_prezeroed == 0 (then it is zero-filled here); dx (same shape as x) may be NULL */
int immtsf_embed_backward(int32_t mode, const float* x, int32_t R, int32_t L, int32_t c_in, int32_t P, int32_t K, int32_t stride,
                          int32_t D, const float* W, const float* dout, float* dW, int32_t dw_prezeroed, float* dx, float p_drop,
                          uint64_t seed, uint64_t site, const uint64_t* seed_step_dev, immtsf_stream_t stream) {
    const EmbDims e{mode, R, L, c_in, P, K, stride, D};
    if (!x || !W || !dout || !dW) return IMMTSF_EINVAL;
    if (emb_bad(e)) return IMMTSF_EUNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DropCfg drop = mk_drop2(p_drop, seed, seed_step_dev);
    const long pairs = (long)R * P;
    const int cs = cdiv(D, 16);
    // >= 64 pairs per slice, S * cs workgroups.  (A slice is a dependent chain of 16-pair steps -- gather, barrier, one dout load, FMAs:
    // with 8 slices PatchTST's 3840 pairs were 30 steps = 62-74 us at the tail of cfg3's backward; 32 slices: 8 steps.)
    int S = (int)min((long)32, max((long)1, pairs / 64));
    while (S > 1 && S * cs > 2048) --S;
    if (S == 1 && !dw_prezeroed) { /* sole writer per element: no zero-fill needed */ }
    else if (!dw_prezeroed) {
        if (int rc = launch_fill(dW, 0.f, (size_t)D * K, s)) return rc;       // (a kernel: memset nodes misbehave under graph replay)
    }
    const int KB = K <= 16 ? 16 : (K <= 32 ? 32 : 64);
    const size_t lds = (16 * (size_t)KB + 256 * (size_t)(KB + 1)) * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(embed_bwd_w_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (KB == 16) hipLaunchKernelGGL(embed_bwd_w_kernel<16>, dim3(cs, S), dim3(256), lds, s, e, x, dout, dW, drop, site);
    else if (KB == 32) hipLaunchKernelGGL(embed_bwd_w_kernel<32>, dim3(cs, S), dim3(256), lds, s, e, x, dout, dW, drop, site);
    else hipLaunchKernelGGL(embed_bwd_w_kernel<64>, dim3(cs, S), dim3(256), lds, s, e, x, dout, dW, drop, site);
    IMMTSF_LAUNCH_CHECK();
    if (dx) {
        const size_t n = (size_t)R * (mode == 0 ? (size_t)L : (size_t)L * c_in);
        if (int rc = launch_fill(dx, 0.f, n, s)) return rc;
        hipLaunchKernelGGL(embed_bwd_x_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, e, W, dout, dx, drop, site);
        IMMTSF_LAUNCH_CHECK();
    }
    return IMMTSF_OK;
}

}  // extern "C"
