// TTF_RecAvg core (fusions/TTF_RecAvg.py:94-102): Gaussian recency-weighted average of a window's packed,
// projected note embeddings for every forecast step:
//   w[i,t] = exp(-(max(t_hat[b,t] - tau[i], 0)/sigma)^2),  E_raw[b,t,:] = sum_i w[i,t] Vp[i,:] / max(sum_i w[i,t], 1e-6)
// HBM/L2-bound: each window's n_b packed rows are streamed once per 32 forecast steps, weights are built in LDS.
#include "recavg.hpp"

namespace {

constexpr int TT = 32;

// grid (B, ceil(d/256)); LDS: wtile[TT*64] | den[TT]
__global__ __launch_bounds__(256) void recavg_fwd_kernel(int T, int d, int Npad, const int* __restrict__ offsets,
                                                          const int* __restrict__ rowmap, const float* __restrict__ tau_pad,
                                                          const float* __restrict__ t_hat, const float* __restrict__ log_sigma,
                                                          const float* __restrict__ Vp, float* __restrict__ Eraw,
                                                          float* __restrict__ denom) {
    __shared__ float wtile[TT * 64];
    __shared__ float den[TT];
    const int b = blockIdx.x, tid = threadIdx.x, e = blockIdx.y * 256 + tid;
    const bool valid = e < d;
    const int o0 = offsets[b], n = offsets[b + 1] - o0;
    const float inv_sigma = expf(-log_sigma[0]);
    (void)Npad;
    for (int t0 = 0; t0 < T; t0 += TT) {
        float acc[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[tt] = 0.f;
        if (tid < TT) den[tid] = 0.f;
        for (int i0 = 0; i0 < n; i0 += 64) {
            __syncthreads();
            for (int x = tid; x < TT * 64; x += 256) {
                const int tt = x >> 6, ii = x & 63, t = t0 + tt, i = i0 + ii;
                float wv = 0.f;
                if (t < T && i < n) {
                    const float dl = fmaxf(t_hat[(size_t)b * T + t] - tau_pad[rowmap[o0 + i]], 0.f) * inv_sigma;
                    wv = expf(-dl * dl);
                }
                wtile[x] = wv;
            }
            __syncthreads();
            if (tid < TT) {   // running denominator (fixed order: deterministic)
                float a = den[tid];
                const int cnt = min(64, n - i0);
                for (int ii = 0; ii < cnt; ++ii) a += wtile[tid * 64 + ii];
                den[tid] = a;
            }
            if (valid) {
                const int cnt = min(64, n - i0);
                const float* vb = Vp + (size_t)(o0 + i0) * d + e;
                for (int ii = 0; ii < cnt; ++ii) {
                    const float v = vb[(size_t)ii * d];
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) acc[tt] = fmaf(wtile[tt * 64 + ii], v, acc[tt]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int t = t0 + tt;
            if (t < T) {
                const float dn = fmaxf(den[tt], 1e-6f);
                if (valid) Eraw[(size_t)(b * T + t) * d + e] = acc[tt] / dn;
                if (blockIdx.y == 0 && tid == 0) denom[b * T + t] = den[tt];
            }
        }
        __syncthreads();
    }
}

// grid (B); 256 threads (4 waves, one note per wave at a time).
// LDS: dDn[T] | red[16].  dS[t,:] = dEraw[t,:]/Dn[t] is formed on the fly.
__global__ __launch_bounds__(256) void recavg_bwd_kernel(int T, int d, const int* __restrict__ offsets,
                                                          const int* __restrict__ rowmap, const float* __restrict__ tau_pad,
                                                          const float* __restrict__ t_hat, const float* __restrict__ log_sigma,
                                                          const float* __restrict__ Vp, const float* __restrict__ Eraw,
                                                          const float* __restrict__ denom, const float* __restrict__ dEraw,
                                                          float* __restrict__ dVp, float* __restrict__ dls_part) {
    extern __shared__ float lds[];
    float* dDn = lds;          // [T]  -(dEraw . Eraw)/Dn, zero where the clamp is active
    float* rDn = lds + T;      // [T]  1/Dn
    float* red = rDn + T;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o0 = offsets[b], n = offsets[b + 1] - o0;
    const float inv_sigma = expf(-log_sigma[0]);
    const float* dE = dEraw + (size_t)b * T * d;
    const float* Er = Eraw + (size_t)b * T * d;
    for (int t = wave; t < T; t += 4) {
        float a = 0.f;
        for (int c = lane; c < d; c += 64) a = fmaf(dE[(size_t)t * d + c], Er[(size_t)t * d + c], a);
        a = wave_sum(a);
        if (lane == 0) {
            const float raw = denom[b * T + t];
            const float dn = fmaxf(raw, 1e-6f);
            rDn[t] = 1.f / dn;
            dDn[t] = (raw >= 1e-6f) ? -a / dn : 0.f;
        }
    }
    __syncthreads();
    float ls_acc = 0.f;
    for (int i = wave; i < n; i += 4) {
        const float tau = tau_pad[rowmap[o0 + i]];
        const float* vr = Vp + (size_t)(o0 + i) * d;
        float* dvr = dVp + (size_t)(o0 + i) * d;
        // pass A: dw[i,t] = dS[t,:] . Vp[i,:] + dDn[t]  -> contributes to d log_sigma
        for (int t = 0; t < T; ++t) {
            float a = 0.f;
            for (int c = lane; c < d; c += 64) a = fmaf(dE[(size_t)t * d + c], vr[c], a);
            a = wave_sum(a) * rDn[t] + dDn[t];
            const float dl = fmaxf(t_hat[(size_t)b * T + t] - tau, 0.f) * inv_sigma;
            const float wv = expf(-dl * dl);
            ls_acc += a * wv * 2.f * dl * dl;   // dw * dw/dlog_sigma ; identical in every lane
        }
        // pass B: dVp[i,:] = sum_t w[i,t] dS[t,:]
        for (int c = lane; c < d; c += 64) {
            float g = 0.f;
            for (int t = 0; t < T; ++t) {
                const float dl = fmaxf(t_hat[(size_t)b * T + t] - tau, 0.f) * inv_sigma;
                g = fmaf(expf(-dl * dl) * rDn[t], dE[(size_t)t * d + c], g);
            }
            dvr[c] = g;
        }
    }
    // one value per wave (all lanes equal) -> block total
    float v = (lane == 0) ? ls_acc : 0.f;
    v = block_sum(v, red);
    if (tid == 0) dls_part[b] = v;
}

}  // namespace

int launch_recavg_fwd(int B, int T, int d, int N, const int* offsets, const int* rowmap, const float* tau_pad,
                      const float* t_hat, const float* log_sigma, const float* Vp, float* Eraw, float* denom,
                      hipStream_t s) {
    if (B <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(recavg_fwd_kernel, dim3(B, cdiv(d, 256)), dim3(256), 0, s, T, d, N, offsets, rowmap, tau_pad, t_hat,
                       log_sigma, Vp, Eraw, denom);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_recavg_bwd(int B, int T, int d, const int* offsets, const int* rowmap, const float* tau_pad, const float* t_hat,
                      const float* log_sigma, const float* Vp, const float* Eraw, const float* denom, const float* dEraw,
                      float* dVp, float* dls_part, hipStream_t s) {
    if (B <= 0) return IMMTSF_OK;
    const size_t lds = (size_t)(2 * T + 16) * sizeof(float);
    if (lds > 64 * 1024) return IMMTSF_EUNSUPPORTED;
    hipLaunchKernelGGL(recavg_bwd_kernel, dim3(B), dim3(256), lds, s, T, d, offsets, rowmap, tau_pad, t_hat, log_sigma, Vp, Eraw,
                       denom, dEraw, dVp, dls_part);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
