// Output head of MMF_XAttn_Add, forward (fusions/MMF_XAttn_Add.py:88-102): residual_head Linear(d -> C), LayerNorm(C),
// dropout, zero the no-text windows, kappa blend with Y_ts in one kernel.
//
// C (the number of series variables) is small: as a GEMM the head is a 2048 x 8 x 768 product that needs split-K plus a
// zero fill, followed by the LayerNorm / blend row kernel (21 us of kernel time for 25 MFLOP, 14.5 us here).  A wave owns
// a row: the lanes stride over d with 16-byte loads, the C dot products are reduced across the wave, every lane then
// holds the row's C values and does the LayerNorm / blend redundantly.  Exact fp32 in both precision modes.
// (The backward stays on the GEMM path: a fused version -- row workgroups for dY / dU plus column-slice workgroups for
// d(res_w) -- was built and measured at 60 us against 35 us for ln_blend_bwd + colsum2 + the two skinny GEMMs.)
#include "tail.hpp"

namespace {

constexpr int HEAD_DV = 4;        // float4 chunks per lane: d <= 1024, d % 4 == 0

struct HeadDims { int BT, T, C, d; };

template <int CM>
__global__ __launch_bounds__(256) void xadd_head_fwd_kernel(HeadDims hd, const float* __restrict__ U, const float* __restrict__ W,
                                                             const float* __restrict__ bW, const float* __restrict__ bWdead,
                                                             const float* __restrict__ Y,
                                                             const unsigned char* __restrict__ mtxt, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float kappa, float* __restrict__ xhat,
                                                             float* __restrict__ rstd, float* __restrict__ Yout, DropCfg drop,
                                                             uint64_t site) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];        // [C][d]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, C = hd.C, d4 = hd.d >> 2;
    const int per = (hd.BT + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(hd.BT, r0 + per);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // the wave's first row travels beside the weights (one global round trip for the usual one row per wave, not two)
    float4 u[HEAD_DV];
#pragma unroll
    for (int j = 0; j < HEAD_DV; ++j) {
        const int q = lane + 64 * j;
        u[j] = (q < d4 && r0 + wave < r1) ? reinterpret_cast<const float4*>(U + (size_t)(r0 + wave) * hd.d)[q] : z4;
    }
    for (int i = threadIdx.x; i < C * d4; i += 256) reinterpret_cast<float4*>(Ws)[i] = reinterpret_cast<const float4*>(W)[i];
    __syncthreads();
    const float inv = 1.f / (1.f + kappa);
    for (int row = r0 + wave; row < r1; row += 4) {
        float acc[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) acc[c] = 0.f;
#pragma unroll
        for (int j = 0; j < HEAD_DV; ++j) {
            const int q = lane + 64 * j;
            if (q < d4) {
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) {
                        const float4 w = reinterpret_cast<const float4*>(Ws + c * hd.d)[q];
                        acc[c] = fmaf(u[j].x, w.x, fmaf(u[j].y, w.y, fmaf(u[j].z, w.z, fmaf(u[j].w, w.w, acc[c]))));
                    }
            }
        }
        if (row + 4 < r1) {
#pragma unroll
            for (int j = 0; j < HEAD_DV; ++j) {
                const int q = lane + 64 * j;
                u[j] = q < d4 ? reinterpret_cast<const float4*>(U + (size_t)(row + 4) * hd.d)[q] : z4;
            }
        }
        const bool live = mtxt[row / hd.T] != 0;
        const float* bias = live ? bW : bWdead;       // a window without text: U is zero there and the head reduces to its own bias
        float mu = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < C) { acc[c] = wave_sum(acc[c]) + bias[c]; mu += acc[c]; }
        mu /= (float)C;
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < C) { const float t = acc[c] - mu; var = fmaf(t, t, var); }
        const float rs = 1.0f / sqrtf(var / (float)C + 1e-5f);
        float mine = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c == lane) mine = acc[c];
        if (lane < C) {
            const float h = (mine - mu) * rs;
            const size_t o = (size_t)row * C + lane;
            xhat[o] = h;
            float y = fmaf(h, gamma[lane], beta[lane]) * dropout_scale(drop, site, (uint64_t)o);
            if (!live) y = 0.f;
            Yout[o] = (Y[o] + kappa * y) * inv;
        }
        if (lane == 0) rstd[row] = rs;
    }
}

}  // namespace

bool xadd_head_supported(int C, int d) { return C >= 1 && C <= 16 && d >= 4 && d <= 256 * HEAD_DV && (d & 3) == 0; }

int launch_xadd_head_fwd(const float* U, const float* W, const float* bW, const float* bWdead, const float* Y, const unsigned char* mtxt, int BT, int T,
                         int C, int d, const float* gamma, const float* beta, float kappa, float* xhat, float* rstd, float* Yout,
                         DropCfg drop, uint64_t site, hipStream_t s) {
    if (!xadd_head_supported(C, d)) return IMMTSF_EUNSUPPORTED;
    if (BT <= 0) return IMMTSF_OK;
    const HeadDims hd{BT, T, C, d};
    const int grid = cdiv(BT, 4);      // one row per wave: a single global-load round trip after the staged weights (2 rows per wave: 14.3 us)
    const size_t lds = (size_t)C * d * sizeof(float);
    if (C <= 8)
        hipLaunchKernelGGL(xadd_head_fwd_kernel<8>, dim3(grid), dim3(256), lds, s, hd, U, W, bW, bWdead, Y, mtxt, gamma, beta, kappa, xhat, rstd, Yout,
                           drop, site);
    else
        hipLaunchKernelGGL(xadd_head_fwd_kernel<16>, dim3(grid), dim3(256), lds, s, hd, U, W, bW, bWdead, Y, mtxt, gamma, beta, kappa, xhat, rstd, Yout,
                           drop, site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
