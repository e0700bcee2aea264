// TTCN, third filter layer + masked softmax + pooling in ONE kernel per direction (bf16 MFMA mode, ttcn_dim <= 32).
//
// The streaming formulation (ttcn.hip) writes the (P*L, F*K) filter tensor once and reads it back five times across
// forward and backward -- 46 MB per pass at the benchmark shape, ~150 us of HBM-bound kernels per step.  But one
// patch's filter tile is only L x F*K (32 x 341), its producer GEMM has K = 32 (ONE MFMA k-step) and the softmax runs
// over the tile's rows: so a workgroup can produce the tile with 2 x 22 MFMAs straight from h2 (L x 32), normalise it in
// the accumulator registers (the 16x16 C layout keeps a column's rows in 4 lane groups x 4 registers: two xor-shuffles
// finish a column reduction) and pool it, without the tile ever leaving the CU.  The backward RECOMPUTES the tile the
// same way, forms d(logits) in registers, parks it in LDS as bf16 and feeds two more MFMA products from there:
//   dz2 = dS W3   (M = L, N = 32, K = F*K)      and      dW3 += dS^T h2   (M = F*K, N = 32, K = L)
// with dW3 / db3 accumulated in registers across the patches of a persistent workgroup and added to HBM once.
// Reference: models/tPatchGNN.py:182-195 (TTCN), Filter_Generators.4 = W3.
#include "ttcn.hpp"
#include "../../include/immtsf.h"
#include "common.hpp"

namespace {

typedef short s16x8 __attribute__((ext_vector_type(8)));

struct FDims { int P, L, F, K, NC, Fp, Kp, NCp; };

__device__ __forceinline__ bf16x8 load8_bf16(const float* __restrict__ src) {     // 8 consecutive fp32 -> bf16x8 (RNE)
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    bf16x8 r;
    r[0] = (bf16_t)a.x; r[1] = (bf16_t)a.y; r[2] = (bf16_t)a.z; r[3] = (bf16_t)a.w;
    r[4] = (bf16_t)b.x; r[5] = (bf16_t)b.y; r[6] = (bf16_t)b.z; r[7] = (bf16_t)b.w;
    return r;
}
__device__ __forceinline__ float col_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// One 16-column tile of one patch: logits by MFMA, masked softmax over the patch's L rows, in registers.
// sm[rt][r] = softmax weight of row rt*16 + fq*4 + r, column ct*16 + fr   (0 for rows >= L)
template <int RT>
__device__ __forceinline__ void sm_tile(const FDims& d, const bf16x8 (&a)[RT], const bf16x8 b, const float bias,
                                        const float (&mk)[RT][4], int fq, float (&sm)[RT][4]) {
    float m = -INFINITY;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt], b, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + fq * 4 + r;
            float v = -INFINITY;
            if (row < d.L) v = (acc[r] + bias) * mk[rt][r] + (1.f - mk[rt][r]) * (-1e8f);
            sm[rt][r] = v;
            m = fmaxf(m, v);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float s = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sm[rt][r] = expf(sm[rt][r] - m);     // exp(-inf) = 0 for the padded rows
            s += sm[rt][r];
        }
    s = col_sum(s);
    const float inv = 1.f / s;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sm[rt][r] *= inv;
}

template <int RT>
__device__ __forceinline__ void load_patch_rows(const FDims& d, const float* __restrict__ h2, const float* __restrict__ mask, int p,
                                                int fr, int fq, bf16x8 (&a)[RT], float (&mk)[RT][4]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = rt * 16 + fr;      // A fragment: row = lane & 15, k = (lane >> 4) * 8 ..
        if (row < d.L) a[rt] = load8_bf16(h2 + ((size_t)p * d.L + row) * d.Kp + fq * 8);
        else a[rt] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cr = rt * 16 + fq * 4 + r;      // C layout rows of this lane
            mk[rt][r] = cr < d.L ? mask[(size_t)p * d.L + cr] : 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- forward
// grid P, 256 threads.  LDS: ctr[NC].
template <int RT>
__global__ __launch_bounds__(256) void ttcn3_fwd_kernel(FDims d, const float* __restrict__ h2, const float* __restrict__ W3p,
                                                         const float* __restrict__ b3p, const float* __restrict__ X,
                                                         const float* __restrict__ mask, const float* __restrict__ Tb,
                                                         float* __restrict__ ctr, float* __restrict__ out, int out_ld, int flag_col) {
    extern __shared__ float lds[];
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    bf16x8 a[RT];
    float mk[RT][4];
    load_patch_rows<RT>(d, h2, mask, p, fr, fq, a, mk);
    const int nct = d.NCp / 16;
    for (int ct = wave; ct < nct; ct += 4) {
        float sm[RT][4];
        const int c = ct * 16 + fr;
        sm_tile<RT>(d, a, load8_bf16(W3p + (size_t)c * d.Kp + fq * 8), b3p[c], mk, fq, sm);
        if (c < d.NC) {             // uniform over the 4 lane groups of a column: the shuffles below stay convergent
            const int fc = c % d.F;
            float acc = 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rt * 16 + fq * 4 + r;
                    if (row < d.L) acc = fmaf(sm[rt][r], X[((size_t)p * d.L + row) * d.Fp + fc], acc);
                }
            acc = col_sum(acc);
            if (fq == 0) {
                lds[c] = acc;
                ctr[(size_t)p * d.NC + c] = acc;
            }
        }
    }
    __syncthreads();
    if (tid < d.K) {
        float s = Tb[tid];
        for (int f = 0; f < d.F; ++f) s += lds[tid * d.F + f];
        out[(size_t)p * out_ld + tid] = fmaxf(s, 0.f);
    }
    if (flag_col >= 0 && tid == 64) {
        float any = 0.f;
        for (int l = 0; l < d.L; ++l) any += mask[(size_t)p * d.L + l];
        out[(size_t)p * out_ld + flag_col] = any > 0.f ? 1.f : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------- backward
constexpr int MAXCT = 6;      // column tiles per wave (NCp <= 384)

// hardware-transpose read of a 16(row) x 8(k) bf16 fragment from a [k][row] LDS image (see gemm.hip)
__device__ __forceinline__ bf16x8 frag_kmajor(const bf16_t* tile, int pitch, int rbase, int kbase, int fr, int fq) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int q = fr >> 2, pp = fr & 3;
    const bf16_t* a0 = tile + (kbase + fq * 8 + q) * pitch + rbase + 4 * pp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * pitch));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// persistent workgroups (grid <= P), 256 threads.
// LDS (bf16 unless noted): W3T[Kp][NCp+8] | dS[RT*16][NCp+8] | h2s[RT*16][Kp+8] | dp[Kp] f32 | Xs[RT*16][Fp] f32 |
// cts[NCp] f32 | smdS[RT*16][NCp+8]
template <int RT>
__global__ __launch_bounds__(256) void ttcn3_bwd_kernel(FDims d, const float* __restrict__ h2, const float* __restrict__ W3p,
                                                         const float* __restrict__ b3p, const float* __restrict__ X,
                                                         const float* __restrict__ mask, const float* __restrict__ ctr,
                                                         const float* __restrict__ out, const float* __restrict__ dout, int out_ld,
                                                         float* __restrict__ dX, float* __restrict__ dpool, float* __restrict__ dz2,
                                                         float* __restrict__ gW3p, float* __restrict__ gb3p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int pitchC = d.NCp + 8, pitchK = d.Kp + 8, ROWS = RT * 16;
    bf16_t* W3T = reinterpret_cast<bf16_t*>(smem);                 // [Kp][pitchC]   W3T[k'][c] = W3[c][k']
    bf16_t* dS = W3T + d.Kp * pitchC;                              // [ROWS][pitchC]
    bf16_t* h2s = dS + ROWS * pitchC;                              // [ROWS][pitchK]
    float* dp = reinterpret_cast<float*>(h2s + ROWS * pitchK);     // [Kp]
    float* Xs = dp + d.Kp;                                         // [ROWS][Fp]  the patch's X tile
    float* cts = Xs + ROWS * d.Fp;                                 // [NCp]       the patch's pooled contributions
    bf16_t* smdS = reinterpret_cast<bf16_t*>(cts + d.NCp);         // [ROWS][pitchC]  sm * dpool: summed over k for the X gradient
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int nct = d.NCp / 16;

    for (int i = tid; i < d.NCp * d.Kp; i += 256) {               // stage W3^T once per workgroup
        const int c = i / d.Kp, k = i - c * d.Kp;
        W3T[k * pitchC + c] = (bf16_t)W3p[i];
    }
    // this wave's B fragments of W3 (c-tiles wave, wave+4, ...) and biases never change: registers for the whole kernel
    bf16x8 bw[MAXCT];
    float bias[MAXCT];
#pragma unroll
    for (int j = 0; j < MAXCT; ++j) {
        const int c = (wave + 4 * j) * 16 + fr;
        if (wave + 4 * j < nct) { bw[j] = load8_bf16(W3p + (size_t)c * d.Kp + fq * 8); bias[j] = b3p[c]; }
        else { bw[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; bias[j] = 0.f; }
    }
    f32x4 accW[MAXCT][2];          // dW3 tiles of this wave: c-tiles wave, wave+4, ... x two 16-wide halves of k'
    float accB[MAXCT];             // db3 of the same c-tiles (lanes with fq == 0 hold column fr)
#pragma unroll
    for (int j = 0; j < MAXCT; ++j) {
        accW[j][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        accW[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        accB[j] = 0.f;
    }
    __syncthreads();

    for (int p = blockIdx.x; p < d.P; p += gridDim.x) {
        // ---- stage: dp, h2 tile (bf16), zero dXs
        if (tid < d.Kp) {
            float g = 0.f;
            if (tid < d.K) {
                g = out[(size_t)p * out_ld + tid] > 0.f ? dout[(size_t)p * out_ld + tid] : 0.f;
                dpool[(size_t)p * d.K + tid] = g;
            }
            dp[tid] = g;
        }
        for (int i = tid; i < ROWS * d.Kp; i += 256) {
            const int l = i / d.Kp, k = i - l * d.Kp;
            h2s[l * pitchK + k] = l < d.L ? (bf16_t)h2[((size_t)p * d.L + l) * d.Kp + k] : (bf16_t)0.f;
        }
        for (int i = tid; i < ROWS * d.Fp; i += 256) Xs[i] = i < d.L * d.Fp ? X[(size_t)p * d.L * d.Fp + i] : 0.f;
        for (int i = tid; i < d.NCp; i += 256) cts[i] = i < d.NC ? ctr[(size_t)p * d.NC + i] : 0.f;
        bf16x8 a[RT];
        float mk[RT][4];
        load_patch_rows<RT>(d, h2, mask, p, fr, fq, a, mk);
        __syncthreads();
        // ---- d(logits) tile by tile, in registers -> LDS (bf16); pooling-path dX; db3
#pragma unroll
        for (int j = 0; j < MAXCT; ++j) {
            const int ct = wave + 4 * j;
            if (ct < nct) {
                float sm[RT][4];
                sm_tile<RT>(d, a, bw[j], bias[j], mk, fq, sm);
                const int c = ct * 16 + fr;
                const bool real = c < d.NC;
                const int fc = real ? c % d.F : 0, kc = real ? c / d.F : 0;
                const float dpk = real ? dp[kc] : 0.f, ct_c = cts[c];
                float colsum = 0.f;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rt * 16 + fq * 4 + r;
                        float ds = 0.f;
                        if (real && row < d.L) {
                            const float smd = sm[rt][r] * dpk;
                            ds = smd * (Xs[row * d.Fp + fc] - ct_c) * mk[rt][r];
                            smdS[row * pitchC + c] = (bf16_t)smd;
                        }
                        dS[row * pitchC + c] = (bf16_t)ds;
                        colsum += ds;
                    }
                colsum = col_sum(colsum);
                accB[j] += colsum;
            }
        }
        __syncthreads();
        // ---- pooling-path gradient of X (all Fp columns: the layer-1 data-gradient GEMM accumulates onto it)
        // dX[l, f] = sum_k sm[l, k*F+f] dpool[k]  (LDS atomics here cost 40 us: 11k contended adds per patch)
        for (int i = tid; i < d.L * d.Fp; i += 256) {
            const int l = i / d.Fp, f = i - l * d.Fp;
            float acc = 0.f;
            if (f < d.F) for (int k = 0; k < d.K; ++k) acc += (float)smdS[l * pitchC + k * d.F + f];
            dX[(size_t)p * d.L * d.Fp + i] = acc;
        }
        // ---- dz2 = (dS W3) * [h2 > 0]:  M = ROWS, N = Kp (2 tiles), K = NCp; tile t -> (rt, nt)
        for (int t = wave; t < RT * 2; t += 4) {
            const int rt = t >> 1, nt = t & 1;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kk = 0; kk < d.NCp; kk += 32) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(dS + (rt * 16 + fr) * pitchC + kk + fq * 8);
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(W3T + (nt * 16 + fr) * pitchC + kk + fq * 8);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + fq * 4 + r, col = nt * 16 + fr;
                if (row < d.L) {
                    const size_t o = ((size_t)p * d.L + row) * d.Kp + col;
                    dz2[o] = h2[o] > 0.f ? acc[r] : 0.f;
                }
            }
        }
        // ---- dW3 += dS^T h2:  M = NCp (this wave's c-tiles), N = Kp, K = ROWS (k-major operands: transposing reads)
#pragma unroll
        for (int j = 0; j < MAXCT; ++j) {
            const int ct = wave + 4 * j;
            if (ct < nct) {
#pragma unroll
                for (int kk = 0; kk < RT * 16; kk += 32) {
                    const bf16x8 af = frag_kmajor(dS, pitchC, ct * 16, kk, fr, fq);
                    const bf16x8 b0 = frag_kmajor(h2s, pitchK, 0, kk, fr, fq), b1 = frag_kmajor(h2s, pitchK, 16, kk, fr, fq);
                    accW[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b0, accW[j][0], 0, 0, 0);
                    accW[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b1, accW[j][1], 0, 0, 0);
                }
            }
        }
        __syncthreads();     // dS / smdS / h2s / Xs / cts are rewritten by the next patch
    }
    // ---- one atomic add per element of dW3 / db3 per workgroup
#pragma unroll
    for (int j = 0; j < MAXCT; ++j) {
        const int ct = wave + 4 * j;
        if (ct < nct) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = ct * 16 + fq * 4 + r, k = nt * 16 + fr;
                    atomicAdd(gW3p + (size_t)c * d.Kp + k, accW[j][nt][r]);
                }
            if (fq == 0) atomicAdd(gb3p + ct * 16 + fr, accB[j]);
        }
    }
}

size_t bwd_lds_bytes(const FDims& d, int RT) {
    const size_t pitchC = d.NCp + 8, pitchK = d.Kp + 8, ROWS = RT * 16;
    return (d.Kp * pitchC + 2 * ROWS * pitchC + ROWS * pitchK) * 2 + (d.Kp + ROWS * d.Fp + d.NCp) * 4 + 64;
}

}  // namespace

bool ttcn_fused_supported(int precision, int L, int F, int K) {
    const int NCp = (F * K + 31) / 32 * 32;
    return precision == 1 && K <= 32 && L >= 1 && L <= 64 && NCp <= 16 * 4 * MAXCT && F <= 32;
}

int launch_ttcn3_fwd(int P, int L, int F, int K, const float* h2, const float* W3p, const float* b3p, const float* X,
                     const float* mask, const float* Tb, float* ctr, float* out, int out_ld, int flag_col, hipStream_t s) {
    FDims d{P, L, F, K, F * K, (F + 31) / 32 * 32, (K + 31) / 32 * 32, (F * K + 31) / 32 * 32};
    const size_t lds = (size_t)d.NC * sizeof(float);
    if (L <= 32) hipLaunchKernelGGL(ttcn3_fwd_kernel<2>, dim3(P), dim3(256), lds, s, d, h2, W3p, b3p, X, mask, Tb, ctr, out, out_ld, flag_col);
    else hipLaunchKernelGGL(ttcn3_fwd_kernel<4>, dim3(P), dim3(256), lds, s, d, h2, W3p, b3p, X, mask, Tb, ctr, out, out_ld, flag_col);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_ttcn3_bwd(int P, int L, int F, int K, const float* h2, const float* W3p, const float* b3p, const float* X,
                     const float* mask, const float* ctr, const float* out, const float* dout, int out_ld, float* dX,
                     float* dpool, float* dz2, float* gW3p, float* gb3p, hipStream_t s) {
    FDims d{P, L, F, K, F * K, (F + 31) / 32 * 32, (K + 31) / 32 * 32, (F * K + 31) / 32 * 32};
    const int RT = L <= 32 ? 2 : 4;
    const size_t lds = bwd_lds_bytes(d, RT);
    const int grid = P < 512 ? P : 512;           // two persistent workgroups per CU
    if (RT == 2) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ttcn3_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(ttcn3_bwd_kernel<2>, dim3(grid), dim3(256), lds, s, d, h2, W3p, b3p, X, mask, ctr, out, dout, out_ld, dX, dpool, dz2, gW3p, gb3p);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ttcn3_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(ttcn3_bwd_kernel<4>, dim3(grid), dim3(256), lds, s, d, h2, W3p, b3p, X, mask, ctr, out, dout, out_ld, dX, dpool, dz2, gW3p, gb3p);
    }
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
