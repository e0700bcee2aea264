// TTF_T2V_XAttn, MIX-FIRST form for long windows (fusions/TTF_T2V_XAttn.py:139-166; orchestration in fusion_blocks.hip).
//
// The folded form (t2v_fold.hip) maps every note through W_tot first (a sum-of-N x d x (d_m + d/2) product) and mixes the d-wide rows;
// that is the right order while a window has fewer notes than forecast steps.  cfg5's windows hold up to 4096 notes for 32 steps: here
// the order is the other one.  With x_n = [note embedding ; Time2Vec(tau_n)] (dmc = d_m + d/2 wide, bf16) and one head:
//     s[n] = u . x_n                       a = softmax over the window's notes          a~[t, n] = dropout(a)[t, n]
//     xbar[b, t, :] = sum_n a~[t, n] x_n   wbar[b, t] = sum_n a~[t, n]
//     E_attn[b, t]  = W_tot xbar[b, t] + c wbar[b, t] + b_o                              (ONE B T x d x dmc product)
// so the per-note work is the score (launch_t2v_scores) and the mix in the RAW space: every pass streams X once (HBM-bound: 2 x dmc
// bytes per note), the products with d in them see B T rows instead of sum-of-N.  Backward, with dxbar = dx W_tot and dwbar = dx . c:
//     da~[t, n] = x_n . dxbar[b, t] + dwbar[b, t]        g[n] = sum_t a~[t, n] da~[t, n]       ds[n] = g[n] - a[n] sum_m g[m]
//     du = sum_n ds[n] x_n        dX_tau[n] = sum_t a~[t, n] dxbar[b, t, d_m:] + ds[n] u[d_m:]        (notes carry no gradient)
// The attention weights leave the weights kernel as a bf16 matrix At (R, 32) -- the A operand of the mix, the weights of g -- so the
// Philox bits are drawn once per (step, note).  Limits: one head, T <= 32, bf16 mode (the operands are the bf16 X image).
#include "t2v_fold.hpp"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) s16x4 pm_lds_s16x4;
typedef short pm_s16x8 __attribute__((ext_vector_type(8)));
constexpr int PT = 32;        // forecast steps per tile (T <= PT)
constexpr int PSB = 64;       // notes per LDS sub-block
constexpr int PCS = 256;      // columns of X per workgroup (forward mix) / per K chunk (backward)

// fragment of a [k][column] bf16 image whose reduction index is the ROW: two hardware-transposed reads (cf. skinny_tn.hip)
__device__ __forceinline__ bf16x8 pm_frag_kmajor(const bf16_t* tile, int pitch, int cbase, int kbase, int fr, int fq) {
    const int q = fr >> 2, pp = fr & 3;
    const bf16_t* a0 = tile + (kbase + fq * 8 + q) * pitch + cbase + 4 * pp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pm_lds_s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pm_lds_s16x4*)(a0 + 4 * pitch));
    const pm_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pm_zero8() {
    return bf16x8{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
}

// ------------------------------------------------------------------------------------------------ weights
// grid (B, ceil(N / 256)), 256 threads, a note per thread: softmax over the window's scores (every workgroup of the window derives the
// same maximum and sum from the whole score row; P written), the dropped weights of every (step, note) as bf16 rows of At (steps past T:
// 0), and the chunk's share of wbar[b, t] = the sum of those bf16 values (wpart (B, chunks, 32); premix_finish adds the chunks up)
__global__ __launch_bounds__(256) void premix_weights_kernel(int T, int N, const int* __restrict__ offsets, const int* __restrict__ rowmap,
                                                              const float* __restrict__ S, float* __restrict__ P, bf16_t* __restrict__ At,
                                                              float* __restrict__ wpart, DropCfg drop, uint64_t site) {
    __shared__ float red[16];
    __shared__ float wsum[4][PT];
    const int b = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ob = offsets[b], n = offsets[b + 1] - ob;
    float* wp = wpart + ((size_t)b * gridDim.y + ch) * PT;
    if (ch * 256 >= n) {
        if (tid < PT) wp[tid] = 0.f;
        return;
    }
    float m = -INFINITY;
    for (int i = tid; i < n; i += 256) m = fmaxf(m, S[ob + i]);
    m = block_max(m, red);
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) sum += expf(S[ob + i] - m);
    sum = block_sum(sum, red);
    const float inv = 1.f / sum;
    const uint64_t seed = drop.seed + ((drop.p > 0.f && drop.seed_dev) ? *drop.seed_dev : 0ull);
    float wl[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) wl[t] = 0.f;
    const int i = ch * 256 + tid;
    if (i < n) {
        const float p = expf(S[ob + i] - m) * inv;
        P[ob + i] = p;
        const int n_orig = rowmap[ob + i] - b * N;
        bf16x8 a[PT / 8];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            float v = 0.f;
            if (t < T) {
                v = p;
                if (drop.p > 0.f) v *= dropout_scale(seed, site, (uint64_t)(b * T + t) * N + n_orig, drop.p, drop.inv_keep);
            }
            const bf16_t h = (bf16_t)v;
            a[t >> 3][t & 7] = h;
            wl[t] = (float)h;
        }
        bf16x8* dst = reinterpret_cast<bf16x8*>(At + (size_t)(ob + i) * PT);
#pragma unroll
        for (int u = 0; u < PT / 8; ++u) dst[u] = a[u];
    }
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const float v = wave_sum(wl[t]);
        if (lane == 0) wsum[wave][t] = v;
    }
    __syncthreads();
    if (tid < PT) wp[tid] = (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);
}

// ------------------------------------------------------------------------------------------------ mix, forward
// grid (B, ceil(dmc / PCS)), 256 threads: xbar[b, t, c0 .. c0 + PCS) = At_b^T X_b on the MFMA: the reduction runs over the window's notes,
// 64 per LDS sub-block ([note][column] images, fragments by transposed reads), the next sub-block in registers while this one is
// multiplied.  A wave owns 4 of the 16 column tiles and both step tiles.  LDS: X image 64 x (PCS + 8) | At image 64 x (PT + 8), bf16.
__global__ __launch_bounds__(256) void premix_fwd_kernel(int T, int dmc, const int* __restrict__ offsets, const bf16_t* __restrict__ X,
                                                          const bf16_t* __restrict__ At, bf16_t* __restrict__ xbar) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pm_smem[];
    constexpr int pX = PCS + 8, pA = PT + 8;
    bf16_t* imX = reinterpret_cast<bf16_t*>(pm_smem);
    bf16_t* imA = imX + PSB * pX;
    const int b = blockIdx.x, c0 = blockIdx.y * PCS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int ob = offsets[b], n = offsets[b + 1] - ob;
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xv[8], av;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256, row = i >> 5, col = c0 + (i & 31) * 8;
            const int r = min(k0 + row, n - 1), c = min(col, dmc - 8);
            xv[u] = *reinterpret_cast<const bf16x8*>(X + (size_t)(ob + r) * dmc + c);          // (masked at the store)
        }
        const int row = tid >> 2;
        av = *reinterpret_cast<const bf16x8*>(At + (size_t)(ob + min(k0 + row, n - 1)) * PT + (tid & 3) * 8);
    };
    if (n > 0) fetch(0);
    for (int k0 = 0; k0 < n; k0 += PSB) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256, row = i >> 5, pc = i & 31;
            const bool ok = k0 + row < n && c0 + pc * 8 < dmc;
            *reinterpret_cast<bf16x8*>(imX + row * pX + pc * 8) = ok ? xv[u] : pm_zero8();
        }
        {
            const int row = tid >> 2;
            *reinterpret_cast<bf16x8*>(imA + row * pA + (tid & 3) * 8) = k0 + row < n ? av : pm_zero8();
        }
        __syncthreads();
        if (k0 + PSB < n) fetch(k0 + PSB);
#pragma unroll
        for (int kk = 0; kk < PSB; kk += 32) {
            bf16x8 a[2], bb[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = pm_frag_kmajor(imA, pA, i * 16, kk, fr, fq);
#pragma unroll
            for (int j = 0; j < 4; ++j) bb[j] = pm_frag_kmajor(imX, pX, (wave * 4 + j) * 16, kk, fr, fq);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[step = 16 i + 4 fq + e][column = c0 + 16 (4 wave + j) + fr]   (a window without notes: zeros -- the product behind reads them)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c0 + (wave * 4 + j) * 16 + fr;
            if (col >= dmc) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = i * 16 + fq * 4 + e;
                if (t < T) xbar[(size_t)(b * T + t) * dmc + col] = (bf16_t)acc[i][j][e];
            }
        }
}

// x_pre[r, :] = window has notes ? acc[r, :] + b_o + c wbar[r] + q : q   (in place on the product's output); wbar[r] = the chunks' sum, written
__global__ __launch_bounds__(256) void premix_finish_kernel(int BT, int T, int d, int chunks, float* __restrict__ xpre, const float* __restrict__ b_o,
                                                             const float* __restrict__ cvec, const float* __restrict__ wpart, float* __restrict__ wbar,
                                                             const float* __restrict__ q_res, const unsigned char* __restrict__ mtxt) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)BT * d) return;
    const int r = (int)(idx / d), j = (int)(idx - (size_t)r * d), b = r / T, t = r - b * T;
    float w = 0.f;
    for (int ch = 0; ch < chunks; ++ch) w += wpart[((size_t)b * chunks + ch) * PT + t];
    if (j == 0) wbar[r] = w;
    const float q = q_res[j];
    xpre[idx] = mtxt[b] ? xpre[idx] + b_o[j] + cvec[j] * w + q : q;
}

// ------------------------------------------------------------------------------------------------ backward
// grid (B, ceil(N / 64)), 256 threads: da~[note, t] = x_note . dxbar[b, t] for 64 notes of a window (a wave: 16 of them, both step tiles),
// the reduction over dmc in chunks of PCS columns ([row][k] images: plain 16-byte fragment reads), then
// g[note] = sum_t At[note, t] (da~[note, t] + dwbar[b, t]).  LDS: X image 64 x (PCS + 8) | dxbar image 32 x (PCS + 8), bf16.
__global__ __launch_bounds__(256) void premix_bwd_da_kernel(int T, int dmc, const int* __restrict__ offsets, const bf16_t* __restrict__ X,
                                                             const bf16_t* __restrict__ At, const bf16_t* __restrict__ dxbar,
                                                             const float* __restrict__ dwbar, float* __restrict__ g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pm_smem[];
    constexpr int pK = PCS + 8;
    bf16_t* imX = reinterpret_cast<bf16_t*>(pm_smem);
    bf16_t* imD = imX + PSB * pK;
    const int b = blockIdx.x, i0 = blockIdx.y * PSB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int ob = offsets[b], n = offsets[b + 1] - ob;
    if (i0 >= n) return;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    bf16x8 xv[8], dv[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256, row = i >> 5, c = min(k0 + (i & 31) * 8, dmc - 8);
            xv[u] = *reinterpret_cast<const bf16x8*>(X + (size_t)(ob + min(i0 + row, n - 1)) * dmc + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * 256, t = i >> 5, c = min(k0 + (i & 31) * 8, dmc - 8);
            dv[u] = *reinterpret_cast<const bf16x8*>(dxbar + (size_t)(b * T + min(t, T - 1)) * dmc + c);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < dmc; k0 += PCS) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256, row = i >> 5, pc = i & 31;
            *reinterpret_cast<bf16x8*>(imX + row * pK + pc * 8) = (i0 + row < n && k0 + pc * 8 < dmc) ? xv[u] : pm_zero8();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * 256, t = i >> 5, pc = i & 31;
            *reinterpret_cast<bf16x8*>(imD + t * pK + pc * 8) = (t < T && k0 + pc * 8 < dmc) ? dv[u] : pm_zero8();
        }
        __syncthreads();
        if (k0 + PCS < dmc) fetch(k0 + PCS);
#pragma unroll
        for (int kk = 0; kk < PCS; kk += 32) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(imX + (wave * 16 + fr) * pK + kk + fq * 8);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(imD + fr * pK + kk + fq * 8);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(imD + (16 + fr) * pK + kk + fq * 8);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b1, acc[1], 0, 0, 0);
        }
        __syncthreads();
    }
    // D[note = i0 + 16 wave + 4 fq + e][t = 16 ct + fr]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int note = i0 + wave * 16 + fq * 4 + e;
        float v = 0.f;
        if (note < n) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int t = ct * 16 + fr;
                if (t < T) v += (float)At[(size_t)(ob + note) * PT + t] * (acc[ct][e] + dwbar[b * T + t]);
            }
        }
        v = row16_sum(v);
        if (fr == 0 && note < n) g[ob + note] = v;
    }
}

// grid (B, ceil(N / 64)), 256 threads: ds[note] = g[note] - P[note] sum_m g[m] (written), and the Time2Vec columns' gradient
// dXt[note, c] = sum_t At[note, t] dxbar[b, t, d_m + c] + ds[note] u[d_m + c]: a 64 x dt x 32 product per workgroup -- ONE k-step: the
// weights' fragment straight from memory, dxbar's Time2Vec columns as a [t][column] LDS image read transposed; a wave = 16 notes, every
// column tile.  LDS: 32 x (dtp + 8) bf16 (dtp = dt rounded up to 16).
__global__ __launch_bounds__(256) void premix_bwd_ds_kernel(int T, int dmc, int d_m, int dt, int dtp, const int* __restrict__ offsets,
                                                             const bf16_t* __restrict__ At, const float* __restrict__ P,
                                                             const float* __restrict__ g, const bf16_t* __restrict__ dxbar,
                                                             const float* __restrict__ u, float* __restrict__ ds, float* __restrict__ dXt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pm_smem[];
    __shared__ float red[16];
    __shared__ float ds_s[PSB];
    bf16_t* imD = reinterpret_cast<bf16_t*>(pm_smem);
    const int pD = dtp + 8;
    const int b = blockIdx.x, i0 = blockIdx.y * PSB, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int ob = offsets[b], n = offsets[b + 1] - ob;
    if (i0 >= n) return;
    float G = 0.f;
    for (int i = tid; i < n; i += 256) G += g[ob + i];
    G = block_sum(G, red);
    if (tid < PSB) {
        const int note = i0 + tid;
        float v = 0.f;
        if (note < n) {
            v = g[ob + note] - P[ob + note] * G;
            ds[ob + note] = v;
        }
        ds_s[tid] = v;
    }
    const int p8 = dtp >> 3;
    for (int x = tid; x < PT * p8; x += 256) {
        const int t = x / p8, c = (x - t * p8) * 8;
        bf16x8 v = pm_zero8();
        if (t < T && c < dt) v = *reinterpret_cast<const bf16x8*>(dxbar + (size_t)(b * T + t) * dmc + d_m + c);       // (dt % 8 == 0)
        *reinterpret_cast<bf16x8*>(imD + t * pD + c) = v;
    }
    const int note_a = i0 + wave * 16 + fr;
    bf16x8 a = pm_zero8();
    if (note_a < n) a = *reinterpret_cast<const bf16x8*>(At + (size_t)(ob + note_a) * PT + fq * 8);
    __syncthreads();
    for (int ct = 0; ct < (dtp >> 4); ++ct) {
        const bf16x8 bb = pm_frag_kmajor(imD, pD, ct * 16, 0, fr, fq);
        const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const int col = ct * 16 + fr;
        if (col >= dt) continue;
        const float uc = u[d_m + col];
#pragma unroll
        for (int e = 0; e < 4; ++e) {       // D[note = 16 wave + 4 fq + e][column = 16 ct + fr]
            const int rl = wave * 16 + fq * 4 + e;
            if (i0 + rl < n) dXt[(size_t)(ob + i0 + rl) * dt + col] = acc[e] + ds_s[rl] * uc;
        }
    }
}

// du = sum over the packed notes of ds[r] X[r, :]: grid (ceil(dmc / 2048), row blocks), 256 threads, a thread = 8 adjacent columns;
// partial sums per row block to a slab, one reduce launch (fixed order: deterministic)
__global__ __launch_bounds__(256) void premix_du_kernel(int dmc, const int* __restrict__ total, int rows_per_block, const bf16_t* __restrict__ X,
                                                         const float* __restrict__ ds, float* __restrict__ slab) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * 8;
    const int R = total[0], r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    if (col >= dmc) return;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
        bf16x8 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const bf16x8*>(X + (size_t)(r + u) * dmc + col);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float w = ds[r + u];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(w, (float)v[u][e], acc[e]);
        }
    }
    for (; r < r1; ++r) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(X + (size_t)r * dmc + col);
        const float w = ds[r];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(w, (float)v[e], acc[e]);
    }
    float* o = slab + (size_t)blockIdx.y * dmc + col;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = acc[e];
}
__global__ __launch_bounds__(64) void premix_du_reduce_kernel(int dmc, int nrb, const float* __restrict__ slab, float* __restrict__ du) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= dmc) return;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    int b = 0;
    for (; b + 8 <= nrb; b += 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += slab[(size_t)(b + e) * dmc + c];
    }
    for (; b < nrb; ++b) s[0] += slab[(size_t)b * dmc + c];
    du[c] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

constexpr int PM_DU_BLOCKS = 128;      // row blocks of the score vector's gradient (x ceil(dmc / 2048) column groups of workgroups)

}  // namespace

bool t2v_premix_shape_ok(int T, int d, int H, int d_m) {
    return H == 1 && T >= 1 && T <= PT && d >= 16 && d <= 1024 && (d % 16) == 0 && d_m > 0 && (d_m % 8) == 0;
}
size_t t2v_premix_du_scratch_floats(int dmc) { return (size_t)PM_DU_BLOCKS * dmc; }

int t2v_premix_chunks(int N) { return cdiv(N, 256); }
int launch_t2v_premix_weights(int B, int T, int N, const int* offsets, const int* rowmap, const float* S, float* P, void* At, float* wpart,
                              DropCfg drop, uint64_t site, hipStream_t s) {
    if (T > PT) return IMMTSF_EUNSUPPORTED;
    hipLaunchKernelGGL(premix_weights_kernel, dim3(B, cdiv(N, 256)), dim3(256), 0, s, T, N, offsets, rowmap, S, P, static_cast<bf16_t*>(At), wpart, drop,
                       site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int launch_t2v_premix_fwd(int B, int T, int dmc, const int* offsets, const void* X, const void* At, void* xbar, hipStream_t s) {
    if (T > PT || (dmc % 8)) return IMMTSF_EUNSUPPORTED;
    const size_t lds = ((size_t)PSB * (PCS + 8) + (size_t)PSB * (PT + 8)) * sizeof(bf16_t);
    hipLaunchKernelGGL(premix_fwd_kernel, dim3(B, cdiv(dmc, PCS)), dim3(256), lds, s, T, dmc, offsets, static_cast<const bf16_t*>(X),
                       static_cast<const bf16_t*>(At), static_cast<bf16_t*>(xbar));
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int launch_t2v_premix_finish(int BT, int T, int N, int d, float* xpre, const float* b_o, const float* cvec, const float* wpart, float* wbar,
                             const float* q_res, const unsigned char* mtxt, hipStream_t s) {
    hipLaunchKernelGGL(premix_finish_kernel, dim3((unsigned)(((size_t)BT * d + 255) / 256)), dim3(256), 0, s, BT, T, d, cdiv(N, 256), xpre, b_o, cvec,
                       wpart, wbar, q_res, mtxt);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int launch_t2v_premix_bwd(int B, int T, int N, int dmc, int d_m, const int* offsets, const int* total, const void* X, const void* At,
                          const float* P, const void* dxbar, const float* dwbar, const float* u, float* g, float* ds, float* dXt, float* du,
                          float* du_slab, hipStream_t s) {
    const int dt = dmc - d_m;
    if (T > PT || (dmc % 8) || dt > 512) return IMMTSF_EUNSUPPORTED;
    const dim3 grid(B, cdiv(N, PSB));
    const size_t lds = ((size_t)PSB * (PCS + 8) + (size_t)PT * (PCS + 8)) * sizeof(bf16_t);
    hipLaunchKernelGGL(premix_bwd_da_kernel, grid, dim3(256), lds, s, T, dmc, offsets, static_cast<const bf16_t*>(X), static_cast<const bf16_t*>(At),
                       static_cast<const bf16_t*>(dxbar), dwbar, g);
    const int dtp = (dt + 15) & ~15;
    hipLaunchKernelGGL(premix_bwd_ds_kernel, grid, dim3(256), (size_t)PT * (dtp + 8) * sizeof(bf16_t), s, T, dmc, d_m, dt, dtp, offsets,
                       static_cast<const bf16_t*>(At), P, g, static_cast<const bf16_t*>(dxbar), u, ds, dXt);
    const int rpb = cdiv(B * N, PM_DU_BLOCKS);
    hipLaunchKernelGGL(premix_du_kernel, dim3(cdiv(dmc, 2048), PM_DU_BLOCKS), dim3(256), 0, s, dmc, total, rpb, static_cast<const bf16_t*>(X), ds, du_slab);
    hipLaunchKernelGGL(premix_du_reduce_kernel, dim3(cdiv(dmc, 64)), dim3(64), 0, s, dmc, PM_DU_BLOCKS, du_slab, du);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
