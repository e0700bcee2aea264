// Weight gradients with a tiny output and a very long reduction: dW (M, N) = dY^T X over R rows, db (M) = column sums of dY, with
// M <= 192, N <= 64 and R in the tens of thousands -- the attention projections of tPatchGNN's encoder layer at many windows
// (models/tPatchGNN.py:118-121: d_model 32, so dW_in is 96 x 32 and dW_out 32 x 32 over B * N * M = 65 536 rows at 4096 windows).
// On the general TN kernel (64 x 64 tiles, the reduction split over 256 workgroups with atomics) these took 160 - 185 us for the
// 16 - 33 MB they read (profiles/r03_w4096_kernel_sequence.txt); they are HBM-bound products of 0.2 - 0.6 GFLOP.
//
// Here a workgroup owns a block of rows and streams it in 128-row sub-blocks: dY and X rows are contiguous (ld = width), so the
// loads are fully coalesced; both go to LDS as bf16 images [row][column] and every operand fragment -- the reduction index is the
// ROW of both images -- is one hardware-transposed read (ds_read_b64_tr_b16).  The M/16 x N/16 output tiles (and M/16 bias tiles
// against a fragment of ones) are dealt to the four waves and stay in registers for the whole block; partial tiles go to a slab
// in MFMA register order, one reduce launch adds the row blocks up.
#include "skinny_tn.hpp"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) s16x4 st_lds_s16x4;
typedef short st_s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 st_frag_kmajor(const bf16_t* tile, int pitch, int cbase, int kbase, int fr, int fq) {
    const int q = fr >> 2, pp = fr & 3;
    const bf16_t* a0 = tile + (kbase + fq * 8 + q) * pitch + cbase + 4 * pp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_s16x4*)(a0 + 4 * pitch));
    const st_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

constexpr int ST_SB = 128;        // rows per LDS sub-block
constexpr int ST_MAXT = 8;        // output tiles per wave (M/16 * N/16 <= 32) + bias tiles

struct STArgs {
    const float* dY; int ldy;
    const float* X; int ldx;
    float* slab;                  // [row block][tile][64 lanes][4]; tiles: (mt, nt) -> mt * NT + nt, then the MT bias tiles
    int R, M, N, RB;
};

// grid: row blocks; 256 threads
__global__ __launch_bounds__(256) void skinny_tn_kernel(const STArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char st_smem[];
    const int MT = a.M >> 4, NT = a.N >> 4, pY = a.M + 8, pX = a.N + 8;
    bf16_t* imY = reinterpret_cast<bf16_t*>(st_smem);
    bf16_t* imX = imY + ST_SB * pY;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int ntile = MT * NT, nall = ntile + MT;              // output tiles, then one bias tile per row tile of dW
    f32x4 acc[ST_MAXT];
#pragma unroll
    for (int j = 0; j < ST_MAXT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    const int row_lo = blockIdx.x * a.RB, row_hi = min(a.R, row_lo + a.RB);
    const int my4 = a.M >> 2, nx4 = a.N >> 2;
    for (int base = row_lo; base < row_hi; base += ST_SB) {
        __syncthreads();                        // the previous sub-block's images have been read
        for (int x = tid; x < ST_SB * my4; x += 256) {
            const int r = x / my4, c = (x - r * my4) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (base + r < row_hi) v = *reinterpret_cast<const float4*>(a.dY + (size_t)(base + r) * a.ldy + c);
            *reinterpret_cast<bf16x4*>(imY + r * pY + c) = bf16x4{(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
        }
        for (int x = tid; x < ST_SB * nx4; x += 256) {
            const int r = x / nx4, c = (x - r * nx4) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (base + r < row_hi) v = *reinterpret_cast<const float4*>(a.X + (size_t)(base + r) * a.ldx + c);
            *reinterpret_cast<bf16x4*>(imX + r * pX + c) = bf16x4{(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
        }
        __syncthreads();
        const int nk = min(ST_SB, row_hi - base + 31 & ~31);
        for (int k = 0; k < nk; k += 32) {
#pragma unroll
            for (int j = 0; j < ST_MAXT; ++j) {
                const int t = wave + 4 * j;
                if (t >= nall) break;
                if (t < ntile) {
                    const int mt = t / NT, nt = t - mt * NT;
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st_frag_kmajor(imY, pY, mt * 16, k, fr, fq),
                                                                     st_frag_kmajor(imX, pX, nt * 16, k, fr, fq), acc[j], 0, 0, 0);
                } else {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st_frag_kmajor(imY, pY, (t - ntile) * 16, k, fr, fq), ones, acc[j], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < ST_MAXT; ++j) {
        const int t = wave + 4 * j;
        if (t >= nall) break;
        *reinterpret_cast<f32x4*>(a.slab + (((size_t)blockIdx.x * nall + t) * 64 + lane) * 4) = acc[j];
    }
}

// one thread per (tile, lane) register quad: sums the row blocks; dW[m][n] (ld N) and db[m]
__global__ __launch_bounds__(256) void skinny_tn_reduce_kernel(const float* __restrict__ slab, int M, int N, int nrb, float* __restrict__ dW,
                                                                float* __restrict__ db) {
    const int MT = M >> 4, NT = N >> 4, ntile = MT * NT, nall = ntile + MT;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nall * 64) return;
    const int t = idx >> 6, l = idx & 63, fr = l & 15, fq = l >> 4;
    const float* p = slab + (size_t)idx * 4;
    const size_t stride = (size_t)nall * 256;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int b = 0; b < nrb; ++b) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)b * stride);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float o[4] = {s.x, s.y, s.z, s.w};
    if (t < ntile) {           // D[i = m = 16 mt + 4 fq + e][j = n = 16 nt + fr]
        const int mt = t / NT, nt = t - mt * NT;
#pragma unroll
        for (int e = 0; e < 4; ++e) dW[(size_t)(mt * 16 + fq * 4 + e) * N + nt * 16 + fr] = o[e];
    } else if (db && fr == 0) {
        const int mt = t - ntile;
#pragma unroll
        for (int e = 0; e < 4; ++e) db[mt * 16 + fq * 4 + e] = o[e];
    }
}

inline int st_row_block(int R) {       // ~256 row blocks of whole sub-blocks
    int rb = (R + 255) / 256;
    rb = (rb + ST_SB - 1) / ST_SB * ST_SB;
    return rb < ST_SB ? ST_SB : rb;
}

}  // namespace

bool skinny_tn_ok(int M, int N, int R, int ldy, int ldx, const void* dY, const void* X, const void* dW) {
    constexpr bool on = true;
    const uintptr_t al = reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X);
    return on && M >= 16 && M <= 192 && N >= 16 && N <= 64 && (M % 16) == 0 && (N % 16) == 0 && R >= 8192 && (ldy % 4) == 0 && (ldx % 4) == 0 &&
           (al & 15) == 0 && dW != nullptr && (M / 16) * (N / 16) + M / 16 <= 4 * ST_MAXT;
}
size_t skinny_tn_scratch_floats(int M, int N, int R) {
    const int rb = st_row_block(R), nrb = (R + rb - 1) / rb;
    return (size_t)nrb * ((M / 16) * (N / 16) + M / 16) * 256;
}
int launch_skinny_tn(const float* dY, int ldy, int M, const float* X, int ldx, int N, int R, float* dW, float* db, float* scratch,
                     hipStream_t s) {
    STArgs a;
    a.dY = dY; a.ldy = ldy; a.X = X; a.ldx = ldx; a.slab = scratch;
    a.R = R; a.M = M; a.N = N; a.RB = st_row_block(R);
    const int nrb = (R + a.RB - 1) / a.RB, nall = (M / 16) * (N / 16) + M / 16;
    const size_t lds = (size_t)ST_SB * (M + 8 + N + 8) * sizeof(bf16_t);
    if (lds > 64 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_tn_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (attr != hipSuccess) return (int)attr;
    }
    hipLaunchKernelGGL(skinny_tn_kernel, dim3(nrb), dim3(256), lds, s, a);
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL(skinny_tn_reduce_kernel, dim3((nall * 64 + 255) / 256), dim3(256), 0, s, scratch, M, N, nrb, dW, db);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
