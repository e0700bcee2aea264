// Dense attention over short sequences with wide heads: softmax(scale Q K^T) V per (batch, head) for L, S <= 32 positions and head
// dimensions up to 256 -- PatchTST's FullAttention (layers/SelfAttention_Family.py:50-77) over the 10 - 12 patches of a variable
// (cfg3: 384 sequences x 2 heads, L = S = 10, E = 256).  As batched GEMMs + a row softmax these were 3 launches forward and 5
// backward, every (batch, head) a 128 x 128 MFMA tile around a 10 x 10 result: 52 - 65 us per launch, 350 us of the 1.6 ms step for
// 0.06 GFLOP.  Here a workgroup owns a (batch, head): Q, K, V (backward: + dO) rows in LDS, scores / dA as wave dot products over
// the head dimension, softmax + Philox dropout by one thread per row, the mixes by one thread per output column.  Exact fp32 in
// both precision modes; same probabilities layout (B, H, L, S) and the same dropout indexing (site, ((b H + h) L + l) S + s) as
// immtsf_softmax_rows_*, so the masks and the saved P are interchangeable with the GEMM path.
#include "../../include/immtsf.h"
#include "common.hpp"

namespace {

constexpr int AM_L = 32;      // most positions
constexpr int AM_E = 256;     // widest head

struct AmDims { int B, L, S, H, E, D; float scale; int causal; };

__device__ __forceinline__ void am_stage(float* dst, int pitch, const float* src, long row_stride, int rows, int width) {
    const int w4 = width >> 2;
    for (int i = threadIdx.x; i < rows * w4; i += 256) {
        const int r = i / w4, c = (i - r * w4) * 4;
        *reinterpret_cast<float4*>(dst + r * pitch + c) = *reinterpret_cast<const float4*>(src + (long)r * row_stride + c);
    }
}
// dot[l][s] = sum_e X[l][e] Y[s][e] for all (l, s): a wave takes rows l = wave, wave + 4, ...; lanes stride e in 16-byte steps
__device__ __forceinline__ void am_dots(const float* X, int px, int L, const float* Y, int py, int S, int E, float* out /* [L][AM_L] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e0 = lane * 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = wave; l < L; l += 4) {
        const float4 x = e0 < E ? *reinterpret_cast<const float4*>(X + l * px + e0) : z4;
        for (int s0 = 0; s0 < S; s0 += 8) {          // eight keys at a time: their LDS reads and lane sums are independent chains
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 y = (s0 + u < S && e0 < E) ? *reinterpret_cast<const float4*>(Y + (s0 + u) * py + e0) : z4;
                a[u] = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, x.w * y.w)));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = wave_sum(a[u]);
            if (lane < 8 && s0 + lane < S) {
                float v = a[0];
#pragma unroll
                for (int u = 1; u < 8; ++u) v = lane == u ? a[u] : v;
                out[l * AM_L + s0 + lane] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void attn_mid_fwd_kernel(AmDims d, const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ P, float* __restrict__ out,
                                                            DropCfg drop, uint64_t site) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float Sc[AM_L * AM_L], Ad[AM_L * AM_L];
    const int L = d.L, S = d.S, E = d.E, D = d.D, H = d.H, pe = E + 4, pd = D + 4;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    float* Qs = lds;
    float* Ks = Qs + L * pe;
    float* Vs = Ks + S * pe;
    am_stage(Qs, pe, q + ((long)b * L * H + h) * E, (long)H * E, L, E);
    am_stage(Ks, pe, k + ((long)b * S * H + h) * E, (long)H * E, S, E);
    am_stage(Vs, pd, v + ((long)b * S * H + h) * D, (long)H * D, S, D);
    __syncthreads();
    am_dots(Qs, pe, L, Ks, pe, S, E, Sc);
    __syncthreads();
    for (int i = threadIdx.x; i < L * S; i += 256) {          // softmax + dropout: a thread per (l, s), the row statistics recomputed per thread
        const int l = i / S, s = i - l * S;
        const int Sv = d.causal ? min(S, l + 1) : S;
        float m = -INFINITY;
        for (int t = 0; t < Sv; ++t) m = fmaxf(m, d.scale * Sc[l * AM_L + t]);
        float sum = 0.f;
        for (int t = 0; t < Sv; ++t) sum += expf(d.scale * Sc[l * AM_L + t] - m);
        const uint64_t row = ((uint64_t)b * H + h) * L + l;
        const float p = s < Sv ? expf(d.scale * Sc[l * AM_L + s] - m) / sum : 0.f;
        P[row * S + s] = p;
        Ad[l * AM_L + s] = p * dropout_scale(drop, site, row * S + s);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256)
        for (int l = 0; l < L; ++l) {
            float a = 0.f;
            for (int s = 0; s < S; ++s) a = fmaf(Ad[l * AM_L + s], Vs[s * pd + c], a);
            out[(((long)b * L + l) * H + h) * D + c] = a;
        }
}

__global__ __launch_bounds__(256) void attn_mid_bwd_kernel(AmDims d, const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ P,
                                                            const float* __restrict__ dout, float* __restrict__ dq, float* __restrict__ dk,
                                                            float* __restrict__ dv, DropCfg drop, uint64_t site) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float Sc[AM_L * AM_L], Ad[AM_L * AM_L], Pr[AM_L * AM_L];      // dA -> dS ; A = P x dropout ; P
    const int L = d.L, S = d.S, E = d.E, D = d.D, H = d.H, pe = E + 4, pd = D + 4;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    float* Qs = lds;
    float* Ks = Qs + L * pe;
    float* Vs = Ks + S * pe;
    float* Gs = Vs + S * pd;          // dO [L][D]
    am_stage(Qs, pe, q + ((long)b * L * H + h) * E, (long)H * E, L, E);
    am_stage(Ks, pe, k + ((long)b * S * H + h) * E, (long)H * E, S, E);
    am_stage(Vs, pd, v + ((long)b * S * H + h) * D, (long)H * D, S, D);
    am_stage(Gs, pd, dout + ((long)b * L * H + h) * D, (long)H * D, L, D);
    const uint64_t row0 = ((uint64_t)b * H + h) * L;
    for (int i = threadIdx.x; i < L * S; i += 256) {
        const int l = i / S, s = i - l * S;
        const float p = P[(row0 + l) * S + s];
        Pr[l * AM_L + s] = p;
        Ad[l * AM_L + s] = p * dropout_scale(drop, site, (row0 + l) * S + s);
    }
    __syncthreads();
    am_dots(Gs, pd, L, Vs, pd, S, D, Sc);          // dA = dO V^T
    // dV[s][c] = sum_l A[l][s] dO[l][c]
    for (int c = threadIdx.x; c < D; c += 256)
        for (int s = 0; s < S; ++s) {
            float a = 0.f;
            for (int l = 0; l < L; ++l) a = fmaf(Ad[l * AM_L + s], Gs[l * pd + c], a);
            dv[(((long)b * S + s) * H + h) * D + c] = a;
        }
    __syncthreads();
    float ds_mine[4];          // dS = P (dA x dropout - sum_s P dA x dropout), as immtsf_softmax_rows_backward: a thread per (l, s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = threadIdx.x + 256 * r;
        ds_mine[r] = 0.f;
        if (i < L * S) {
            const int l = i / S, s = i - l * S;
            float dot = 0.f;
            for (int t = 0; t < S; ++t) dot = fmaf(Sc[l * AM_L + t], Ad[l * AM_L + t], dot);        // sum_t dA[t] x (P[t] x dropout[t])
            const float p = Pr[l * AM_L + s];
            ds_mine[r] = d.scale * (Sc[l * AM_L + s] * Ad[l * AM_L + s] - p * dot);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = threadIdx.x + 256 * r;
        if (i < L * S) Sc[(i / S) * AM_L + (i - (i / S) * S)] = ds_mine[r];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256) {
        for (int l = 0; l < L; ++l) {          // dQ = scale dS K
            float a = 0.f;
            for (int s = 0; s < S; ++s) a = fmaf(Sc[l * AM_L + s], Ks[s * pe + c], a);
            dq[(((long)b * L + l) * H + h) * E + c] = a;
        }
        for (int s = 0; s < S; ++s) {          // dK = scale dS^T Q
            float a = 0.f;
            for (int l = 0; l < L; ++l) a = fmaf(Sc[l * AM_L + s], Qs[l * pe + c], a);
            dk[(((long)b * S + s) * H + h) * E + c] = a;
        }
    }
}

inline bool am_ok(int L, int S, int E, int D) {
    return L >= 1 && S >= 1 && L <= AM_L && S <= AM_L && E >= 4 && D >= 4 && E <= AM_E && D <= AM_E && (E & 3) == 0 && (D & 3) == 0;
}
inline DropCfg am_drop(float p, uint64_t seed, const uint64_t* seed_dev) {
    DropCfg d;
    d.seed = seed;
    d.p = p > 0.f ? p : 0.f;
    d.inv_keep = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    d.seed_dev = seed_dev;
    return d;
}

}  // namespace

extern "C" {

int32_t immtsf_attn_mid_supported(int32_t L, int32_t S, int32_t E, int32_t D) { return am_ok(L, S, E, D) ? 1 : 0; }

int immtsf_attn_mid_forward(const float* q, const float* k, const float* v, int32_t B, int32_t L, int32_t S, int32_t H, int32_t E, int32_t D,
                            float scale, int32_t causal, float p_drop, uint64_t seed, uint64_t site, const uint64_t* seed_step_dev, float* P,
                            float* out, immtsf_stream_t stream) {
    if (!q || !k || !v || !P || !out || B < 0 || H <= 0 || !am_ok(L, S, E, D) || p_drop < 0.f || p_drop >= 1.f) return IMMTSF_EINVAL;
    if (((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) != 0) return IMMTSF_EINVAL;
    if (B == 0) return IMMTSF_OK;
    const AmDims d{B, L, S, H, E, D, scale, causal};
    const size_t lds = ((size_t)(L + S) * (E + 4) + (size_t)S * (D + 4)) * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mid_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_mid_fwd_kernel, dim3(B * H), dim3(256), lds, static_cast<hipStream_t>(stream), d, q, k, v, P, out,
                       am_drop(p_drop, seed, seed_step_dev), site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int immtsf_attn_mid_backward(const float* q, const float* k, const float* v, const float* P, const float* dout, int32_t B, int32_t L, int32_t S,
                             int32_t H, int32_t E, int32_t D, float scale, float p_drop, uint64_t seed, uint64_t site,
                             const uint64_t* seed_step_dev, float* dq, float* dk, float* dv, immtsf_stream_t stream) {
    if (!q || !k || !v || !P || !dout || !dq || !dk || !dv || B < 0 || H <= 0 || !am_ok(L, S, E, D) || p_drop < 0.f || p_drop >= 1.f)
        return IMMTSF_EINVAL;
    if (((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(dout)) & 15) != 0)
        return IMMTSF_EINVAL;
    if (B == 0) return IMMTSF_OK;
    const AmDims d{B, L, S, H, E, D, scale, 0};
    const size_t lds = ((size_t)(L + S) * (E + 4) + (size_t)(S + L) * (D + 4)) * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mid_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_mid_bwd_kernel, dim3(B * H), dim3(256), lds, static_cast<hipStream_t>(stream), d, q, k, v, P, dout, dq, dk, dv,
                       am_drop(p_drop, seed, seed_step_dev), site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

}  // extern "C"
