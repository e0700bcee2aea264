// TTF_RecAvg core (fusions/TTF_RecAvg.py:94-102): Gaussian recency-weighted average of a window's packed,
// projected note embeddings for every forecast step:
//   w[i,t] = exp(-(max(t_hat[b,t] - tau[i], 0)/sigma)^2),  E_raw[b,t,:] = sum_i w[i,t] Vp[i,:] / max(sum_i w[i,t], 1e-6)
// HBM/L2-bound: each window's n_b packed rows are streamed once per 32 forecast steps, weights are built in LDS.
#include "recavg.hpp"
#include <stdlib.h>

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


// ---- bf16 mode: the two contractions of the backward as MFMA tiles.  Per window: dS = dEraw / Dn (T <= 32 rows, bf16 in LDS,
// staged once), then per block of 32 notes  dw[i, t] = Vp[i, :] . dS[t, :] + dDn[t]  (32 x 32 over d: one 16 x 16 tile per wave)
// and  dVp[i, :] = sum_t w[i, t] dS[t, :]  (one K-step, formed transposed: a lane stores 16 bytes).  The fp32 kernel above spends
// its time in T wave reductions and d/64 x T exponentials per note: 818 us per step at cfg4 (64 windows, d = 768).
typedef __attribute__((address_space(3))) s16x4 ra_lds_s16x4;
__device__ __forceinline__ bf16x8 ra_frag_row(const bf16_t* tile, int pitch, int row0, int k0, int fr, int fq) {
    return *reinterpret_cast<const bf16x8*>(tile + (row0 + fr) * pitch + k0 + fq * 8);
}
__device__ __forceinline__ bf16x8 ra_frag_kmajor(const bf16_t* tile, int pitch, int rbase, int kbase, int fr, int fq) {
    const int q = fr >> 2, pp = fr & 3;
    const bf16_t* a0 = tile + (kbase + fq * 8 + q) * pitch + rbase + 4 * pp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ra_lds_s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ra_lds_s16x4*)(a0 + 4 * pitch));
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// grid B, 256 threads.  LDS: dSb [32][d + 8] | Vb [32][d + 8] | Wt [32][40] (bf16) | rDn, dDn, that [32] | red [16] (fp32)
__global__ __launch_bounds__(256) void recavg_bwd_mfma_kernel(int T, int d, const int* __restrict__ offsets,
                                                               const int* __restrict__ rowmap, const float* __restrict__ tau_pad,
                                                               const float* __restrict__ t_hat, const float* __restrict__ log_sigma,
                                                               const float* __restrict__ Vp, const float* __restrict__ Eraw,
                                                               const float* __restrict__ denom, const float* __restrict__ dEraw,
                                                               float* __restrict__ dVp, float* __restrict__ dls_part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ra_smem[];
    const int pd = d + 8;
    bf16_t* dSb = reinterpret_cast<bf16_t*>(ra_smem);
    bf16_t* Vb = dSb + 32 * pd;
    bf16_t* Wt = Vb + 32 * pd;
    float* rDn = reinterpret_cast<float*>(Wt + 32 * 40);
    float* dDn = rDn + 32;
    float* that = dDn + 32;
    float* red = that + 32;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int o0 = offsets[b], n = offsets[b + 1] - o0;
    const float inv_sigma = expf(-log_sigma[0]);
    const float* dE = dEraw + (size_t)b * T * d;
    const float* Er = Eraw + (size_t)b * T * d;
    // dS = dE / Dn as bf16; dDn[t] = -(dE[t] . Eraw[t]) / Dn[t] (zero where the clamp of the forward is active)
    for (int t = wave; t < 32; t += 4) {
        float a = 0.f, rd = 0.f, raw = 0.f;
        if (t < T) { raw = denom[b * T + t]; rd = 1.f / fmaxf(raw, 1e-6f); }
        for (int c = lane * 4; c < d; c += 256) {
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < T) {
                g = *reinterpret_cast<const float4*>(dE + (size_t)t * d + c);
                const float4 e = *reinterpret_cast<const float4*>(Er + (size_t)t * d + c);
                a = fmaf(g.x, e.x, a); a = fmaf(g.y, e.y, a); a = fmaf(g.z, e.z, a); a = fmaf(g.w, e.w, a);
            }
            const bf16x4 hv = {(bf16_t)(g.x * rd), (bf16_t)(g.y * rd), (bf16_t)(g.z * rd), (bf16_t)(g.w * rd)};
            *reinterpret_cast<bf16x4*>(dSb + t * pd + c) = hv;
        }
        a = wave_sum(a);
        if (lane == 0) {
            rDn[t] = rd;
            dDn[t] = (t < T && raw >= 1e-6f) ? -a * rd : 0.f;
            that[t] = t < T ? t_hat[(size_t)b * T + t] : 0.f;
        }
    }
    float ls_acc = 0.f;
    const int it = wave >> 1, tt = wave & 1;          // this wave's 16 x 16 tile of dw: notes 16 it .., steps 16 tt ..
    for (int i0 = 0; i0 < n; i0 += 32) {
        __syncthreads();                             // (first pass: dSb complete; later: the previous block's Vb / Wt have been read)
        for (int x = tid; x < 32 * (d >> 2); x += 256) {       // this block's notes as bf16 rows (zero past n)
            const int i = x / (d >> 2), c = (x - i * (d >> 2)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i0 + i < n) v = *reinterpret_cast<const float4*>(Vp + (size_t)(o0 + i0 + i) * d + c);
            const bf16x4 hv = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
            *reinterpret_cast<bf16x4*>(Vb + i * pd + c) = hv;
        }
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = 0; k < d; k += 32)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ra_frag_row(Vb, pd, it * 16, k, fr, fq), ra_frag_row(dSb, pd, tt * 16, k, fr, fq), acc, 0, 0, 0);
        {   // lane: step t = 16 tt + fr, notes 16 it + 4 fq + e
            const int t = tt * 16 + fr;
            const float th = that[t], dd = dDn[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = it * 16 + fq * 4 + e;
                float wv = 0.f;
                if (i0 + i < n && t < T) {
                    const float tau = tau_pad[rowmap[o0 + i0 + i]];
                    const float dl = fmaxf(th - tau, 0.f) * inv_sigma;
                    wv = expf(-dl * dl);
                    ls_acc += (acc[e] + dd) * wv * 2.f * dl * dl;
                }
                Wt[i * 40 + t] = (bf16_t)wv;
            }
        }
        __syncthreads();
        // dVp[i, c] = sum_t w[i, t] dS[t, c]: tiles (note tile, 16-column tile) dealt to the waves, transposed product
        for (int tile = wave; tile < 2 * (d >> 4); tile += 4) {
            const int nt = tile >> 1, mi = tile & 1;
            const bf16x8 wfrag = ra_frag_row(Wt, 40, mi * 16, 0, fr, fq);              // [note][t]
            const bf16x8 sfrag = ra_frag_kmajor(dSb, pd, nt * 16, 0, fr, fq);          // [column][t] through the transpose
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sfrag, wfrag, z, 0, 0, 0);      // [column 4 fq + e][note fr]
            const int i = i0 + mi * 16 + fr;
            if (i < n) *reinterpret_cast<float4*>(dVp + (size_t)(o0 + i) * d + nt * 16 + fq * 4) = make_float4(c4[0], c4[1], c4[2], c4[3]);
        }
    }
    __syncthreads();
    const float v = block_sum(ls_acc, red);
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
                      float* dVp, float* dls_part, hipStream_t s, int precision) {
    if (B <= 0) return IMMTSF_OK;
    constexpr bool mfma_on = true;
    const size_t lm = (size_t)2 * 32 * (d + 8) * 2 + 32 * 40 * 2 + (3 * 32 + 16) * sizeof(float);
    const uintptr_t al = reinterpret_cast<uintptr_t>(Vp) | reinterpret_cast<uintptr_t>(Eraw) | reinterpret_cast<uintptr_t>(dEraw) |
                         reinterpret_cast<uintptr_t>(dVp);
    if (precision == 1 && mfma_on && T <= 32 && (d % 32) == 0 && lm <= 150 * 1024 && (al & 15) == 0) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(recavg_bwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL(recavg_bwd_mfma_kernel, dim3(B), dim3(256), lm, s, T, d, offsets, rowmap, tau_pad, t_hat, log_sigma, Vp, Eraw,
                           denom, dEraw, dVp, dls_part);
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    const size_t lds = (size_t)(2 * T + 16) * sizeof(float);
    if (lds > 64 * 1024) return IMMTSF_EUNSUPPORTED;
    hipLaunchKernelGGL(recavg_bwd_kernel, dim3(B), dim3(256), lds, s, T, d, offsets, rowmap, tau_pad, t_hat, log_sigma, Vp, Eraw,
                       denom, dEraw, dVp, dls_part);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
