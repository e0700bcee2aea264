// bf16-in-memory MFMA GEMM for gfx950: both operands are bf16 in HBM and travel global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no conversion, no ds_write), through a ring of S LDS stages with
// ONE raw s_barrier per 64-deep K step and a counted s_waitcnt vmcnt(N) that leaves S-2 later stages in flight
// across the barrier.  At the fusion shapes (2048 x 768 x 768 and relatives) the K loop is bound by what a CU can
// pull out of its XCD's L2 (r01: time linear in K, 11.8 ns per K unit with fp32 activations whatever the tiling);
// this path halves those bytes and keeps 2-3 K steps of them in flight instead of one.
//
// LDS-DMA writes lane-linearly (wave-uniform LDS base + lane * 16 B), so the LDS images are plain arrays of 16-byte
// chunks and every bank-conflict-avoiding permutation lives in the per-lane SOURCE address and in the fragment reads:
//   * R image (operand rows x 64 k, k contiguous in memory: A of NT/NN, B of NT): chunk (row, c) sits at position
//     row*8 + (c ^ ((row>>1)&7)); a fragment is one ds_read_b128 and the 16 lanes of a read group hit 16 different
//     16-byte slots of the 256-byte bank row.
//   * T image (64 k-lines x operand rows, the reduction index is the slow one in memory: B of NN, A and B of TN):
//     64-column blocks of [k][8 chunks], chunk (k, n8) at position k*8 + (n8 ^ sw(k)),
//     sw(k) = (((k>>1)&1)<<1) | (((k>>3)&1)<<2); fragments come out of ds_read_b64_tr_b16 (hardware transpose),
//     whose 32-lane groups then cover all 64 banks exactly once.  Global side: 8 lanes fetch one full 128-byte line.
// Out-of-range k-lines / columns (ragged K = note count, K not a multiple of 64, N edge of a T image) are fetched from
// a 16-byte zero page; out-of-range rows of an R image are clamped to the last row (their products only reach output
// elements that are never stored).  The steps past the end of the K range issue zero-page loads too, so every wave
// issues the same number of loads per step and the vmcnt arithmetic has no special cases.
#include "gemm.hpp"
#include "rowops.hpp"
#include <string.h>
#include <stdlib.h>

namespace {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_vp;
typedef const __attribute__((address_space(1))) void* glb_vp;

__device__ __attribute__((aligned(16))) const unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ float gelu_erf2(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf2_grad(float z) {      // Phi(z) + z phi(z)
    return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int BK2 = 64;

// WK > 1: WK groups of WM x WN waves share the output tile and split the K steps round-robin (group kg takes steps
// kt0 + kg, kt0 + kg + WK, ...), each with its own ring; their accumulators are summed through LDS in the epilogue.
// More waves and more LDS-DMA streams per CU without smaller per-wave tiles: one LDS-DMA stream per wave lands ~1 KiB
// per ~100 cycles whatever is asked of it, so a 4-wave workgroup cannot pull what the CU's L2 port offers.
// (the kernel's body as a device function: bx / by / bz = the workgroup's tile index, K split and problem; gx / gy = the extents
// of the tile and split dimensions -- gemm2_kernel passes blockIdx / gridDim, gemm2_group_kernel a sub-problem's own numbering)
template <bool TA, bool TB, int BM, int BN, int WM, int WN, int S, int WK = 1>
__device__ __forceinline__ void gemm2_body(const GemmArgs& g, const int bx, const int by, const int bz, const int gx, const int gy,
                                           const int tile_m_given = -1, const int tile_n_given = -1) {
    constexpr int NT = WM * WN * 64;          // threads of one K group
    constexpr int NTALL = NT * WK;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int CA = BM * 8 / NT, CB = BN * 8 / NT;       // 16-byte chunks per thread, stage and operand
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0 && CA >= 1 && CB >= 1, "whole chunks per thread");
    // T images come in blocks of 64 columns (8 chunks per k-line) or, when the tile width is only a multiple of 32, of
    // 32 columns (4 chunks per k-line, chunk (k, n8) at position k*4 + (n8 ^ (((k>>3)&1)<<1)): conflict-free as well,
    // 64-byte global segments)
    static_assert(!TA || BM % 32 == 0, "T images come in 32- or 64-column blocks");
    static_assert(!TB || BN % 32 == 0, "T images come in 32- or 64-column blocks");
    constexpr int LA = (BM % 64 == 0) ? 3 : 2, LB = (BN % 64 == 0) ? 3 : 2;     // log2(chunks per k-line of a block)
    static_assert((BM / WM) % 16 == 0 && (BN / WN) % 16 == 0, "wave tile = whole 16x16 MFMA tiles");
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, SB = A_BYTES + B_BYTES;
    constexpr int LPS = CA + CB;
    static_assert((S - 2) * LPS <= 63, "vmcnt immediate");
    constexpr int CP = BN + 4;                              // epilogue staging pitch (floats)
    constexpr int RING = S * SB;                            // bytes of one K group's ring
    // the epilogue stages the accumulators through the (then dead) ring in bands of BAND tile rows
    constexpr int BAND = (BM * CP * 4 <= RING) ? BM : (BM / 2 * CP * 4 <= RING) ? BM / 2 : BM / 4;
    static_assert(BAND * CP * 4 <= RING && BAND % 16 == 0, "epilogue band must fit the ring");

    __shared__ __attribute__((aligned(16))) unsigned char smem_all[WK * RING];

    GemmProblem P = g.p[g.zbatch ? 0 : bz];
    if (g.zbatch) {        // one problem, grid.z batches at fixed strides (see GemmArgs)
        P.Ah = reinterpret_cast<const bf16_t*>(P.Ah) + (size_t)bz * g.zsA;
        P.Bh = reinterpret_cast<const bf16_t*>(P.Bh) + (size_t)bz * g.zsB;
        if (P.C) P.C += (size_t)bz * g.zsC;
        if (P.Ch) P.Ch = reinterpret_cast<bf16_t*>(P.Ch) + (size_t)bz * g.zsC;
        if (P.Cpre) P.Cpre += (size_t)bz * g.zsC;
    }
    int M = g.M, K = g.K;
    const int N = g.N;
    if (g.dyn) {
        const int dv = g.dyn[g.zbatch ? bz * g.dyn_stride : 0];
        if (g.dyn_which == 0) M = dv; else K = dv;
    }
    int lin = bx, tile_m, tile_n;
    if (tile_m_given >= 0) {          // the caller has placed this workgroup (gemm2_group_kernel)
        tile_m = tile_m_given;
        tile_n = tile_n_given;
    } else if (g.g2_fast) {          // launcher-made reciprocals instead of run-time divisions (see GemmArgs)
        if (g.xcd_remap && g.xcd_gm > 0) {
            const int xcd = lin & 7, idx = lin >> 3;
            const int qi = g.g2_cx_magic ? (int)__umulhi((unsigned)idx, g.g2_cx_magic) : idx;
            tile_m = (xcd >> g.g2_xc_shift) * g.g2_rows_x + qi;
            tile_n = (xcd & ((1 << g.g2_xc_shift) - 1)) * g.g2_cols_x + (idx - qi * g.g2_cols_x);
        } else {
            if (g.xcd_remap) {
                const int total = gx, q = total >> 3, r = total & 7, xcd = lin & 7;
                lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
            }
            tile_m = g.g2_tn_magic ? (int)__umulhi((unsigned)lin, g.g2_tn_magic) : lin;
            tile_n = lin - tile_m * g.g2_tiles_n;
        }
    } else {
    const int tiles_n = (N + BN - 1) / BN;
    if (g.xcd_remap && g.xcd_gm > 0) {
        // Exact 2-D partition: workgroup i runs on XCD i % 8 (round-robin dispatch) and is that XCD's (i / 8)-th tile; the 8
        // XCDs own an XR x XC arrangement (XR = xcd_gm, XR * XC = 8) of equal rectangles of the tile grid, so an XCD's L2
        // fetches 1/XR of A and 1/XC of B instead of a sliver of A and ALL of B (TN 768x768x2048, 12 x 12 tiles, 4 x 2:
        // 29.5 -> 18.9 MB fetched).  The launcher picks XR to minimise that sum and only when the grid divides exactly.
        const int XR = g.xcd_gm, XC = 8 / XR, tiles_m = (g.M + BM - 1) / BM;
        const int rows_x = tiles_m / XR, cols_x = tiles_n / XC, xcd = lin & 7, idx = lin >> 3;
        tile_m = (xcd / XC) * rows_x + idx / cols_x;
        tile_n = (xcd % XC) * cols_x + idx % cols_x;
    } else {
        if (g.xcd_remap) {        // contiguous tile range per XCD (bijective for any grid size), see gemm.hip
            const int total = gx, q = total >> 3, r = total & 7, xcd = lin & 7;
            lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
        }
        tile_m = lin / tiles_n;
        tile_n = lin % tiles_n;
    }
    }
    const int row0 = tile_m * BM, col0 = tile_n * BN;
    if (row0 >= M) return;

    const int tid_all = threadIdx.x, kg = WK > 1 ? tid_all / NT : 0;
    const int tid = WK > 1 ? tid_all % NT : tid_all, lane = tid & 63, wave = tid >> 6;
    unsigned char* smem = smem_all + kg * RING;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(P.Ah);
    const bf16_t* __restrict__ Bm = reinterpret_cast<const bf16_t*>(P.Bh);
    const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page);

    // K range of this split, in 64-deep steps
    const int nk_total = (K + BK2 - 1) / BK2;
    const int splits = gy;
    const int per = (g.g2_fast && g.g2_per) ? g.g2_per : (nk_total + splits - 1) / splits;
    const int kt0 = by * per, kt1 = min(nk_total, kt0 + per);
    if (splits > 1 && kt0 >= kt1) return;
    const bool atom = splits > 1 || g.atomic_c;       // results leave by fp32 atomics into a zeroed C
    const int kend = min(K, kt1 * BK2);

    // ---- per-thread source pointers of its chunks at k = 0 (R image: pointer to (row, c*8); T image: pointer to
    // (k-line, column)), the k coordinate the validity test needs, and whether the chunk's columns exist at all
    const bf16_t* pa[CA];
    const bf16_t* pb[CB];
    int ka[CA], kb[CB];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        const int p = tid + i * NT;
        if (!TA) {
            const int r = p >> 3, c = (p & 7) ^ ((r >> 1) & 7);
            int grow = min(row0 + r, M - 1);
            if (g.a_rowmap) grow = g.a_rowmap[grow];
            pa[i] = A + (size_t)grow * g.lda + c * 8;
            ka[i] = c * 8;
        } else {
            const int b = p >> (6 + LA), pk = p & ((64 << LA) - 1), k = pk >> LA;
            const int n8 = (pk & ((1 << LA) - 1)) ^ (LA == 3 ? ((((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2)) : (((k >> 3) & 1) << 1));
            const int col = row0 + b * (8 << LA) + n8 * 8;
            pa[i] = A + (size_t)k * g.lda + col;
            ka[i] = (col + 8 <= M) ? k : (1 << 30);        // columns past M: always the zero page
        }
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const int p = tid + i * NT;
        if (!TB) {
            const int r = p >> 3, c = (p & 7) ^ ((r >> 1) & 7);
            const int gcol = min(col0 + r, N - 1);
            pb[i] = Bm + (size_t)gcol * g.ldb + c * 8;
            kb[i] = c * 8;
        } else {
            const int b = p >> (6 + LB), pk = p & ((64 << LB) - 1), k = pk >> LB;
            const int n8 = (pk & ((1 << LB) - 1)) ^ (LB == 3 ? ((((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2)) : (((k >> 3) & 1) << 1));
            const int col = col0 + b * (8 << LB) + n8 * 8;
            pb[i] = Bm + (size_t)k * g.ldb + col;
            kb[i] = (col + 8 <= N) ? k : (1 << 30);
        }
    }
    const size_t stepA = TA ? (size_t)BK2 * g.lda : (size_t)BK2;       // elements per K step
    const size_t stepB = TB ? (size_t)BK2 * g.ldb : (size_t)BK2;

    // issue the LDS-DMA loads of K step `t` into ring slot `slot`
    auto issue = [&](int t, int slot) {
        const int k0 = t * BK2;
        unsigned char* sa = smem + slot * SB + wave * 1024;
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const bf16_t* src = (k0 + ka[i] < kend) ? pa[i] + (size_t)(t) * stepA : zp;
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(sa + i * (NT * 16)), 16, 0, 0);
        }
        unsigned char* sb = smem + slot * SB + A_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const bf16_t* src = (k0 + kb[i] < kend) ? pb[i] + (size_t)(t) * stepB : zp;
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(sb + i * (NT * 16)), 16, 0, 0);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // TN + bias gradient: bias_grad[m] = sum_k A[k][m] as one more MFMA column against a fragment of ones, in the
    // workgroups of the first tile column and there in the waves of the first wave column
    const bool want_bsum = TA && TB && g.ones_col && P.bias_grad != nullptr && col0 == 0 && wn0 == 0;
    f32x4 accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

    const int fr = lane & 15, fq = lane >> 4;
    // fragment byte offsets inside a stage (constant over the K loop)
    int offA[TM], offB[TN];
    if (!TA) {
#pragma unroll
        for (int i = 0; i < TM; ++i) offA[i] = (wm0 + i * 16 + fr) * 128;
    } else {
        const int q = fr >> 2, p = fr & 3, sw = LA == 3 ? (((q >> 1) << 1) | ((fq & 1) << 2)) : ((fq & 1) << 1);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rb = wm0 + i * 16, n8 = 2 * ((rb & ((8 << LA) - 1)) >> 4) + (p >> 1);
            offA[i] = (rb >> (3 + LA)) * (1024 << LA) + (((fq * 8 + q) << LA) + (n8 ^ sw)) * 16 + (p & 1) * 8;
        }
    }
    if (!TB) {
#pragma unroll
        for (int j = 0; j < TN; ++j) offB[j] = A_BYTES + (wn0 + j * 16 + fr) * 128;
    } else {
        const int q = fr >> 2, p = fr & 3, sw = LB == 3 ? (((q >> 1) << 1) | ((fq & 1) << 2)) : ((fq & 1) << 1);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int rb = wn0 + j * 16, n8 = 2 * ((rb & ((8 << LB) - 1)) >> 4) + (p >> 1);
            offB[j] = A_BYTES + (rb >> (3 + LB)) * (1024 << LB) + (((fq * 8 + q) << LB) + (n8 ^ sw)) * 16 + (p & 1) * 8;
        }
    }
    const int rsw = fr >> 1;      // R image: chunk c of row r sits in slot c ^ ((r>>1)&7), and (r>>1)&7 == fr>>1 here

    auto frag = [&](const unsigned char* st, bool T, int L, int off, int kk) -> bf16x8 {
        if (!T) return *reinterpret_cast<const bf16x8*>(st + off + (((kk * 4 + fq) ^ rsw) << 4));
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        // k-lines kk*32 + fq*8 + q and + 4; a k-line of a block is 16 << L bytes
        const unsigned char* a0 = st + off + kk * (32 * (16 << L));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * (16 << L)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    // ---- the ring.  Before iteration t's wait, stages kt0 .. kt0+t+S-2 have been issued; stage t is complete once at
    // most (S-2)*LPS of this wave's loads are outstanding, and everybody's once all waves are past the barrier -- which
    // also says everyone is done reading slot (t-1)%S, the slot stage t+S-1 goes into.
#pragma unroll
    for (int s = 0; s < S - 1; ++s) issue(kt0 + kg + s * WK, s);
    int slot = 0, fill = S - 1;
    for (int t = kt0 + kg; t < kt1 + kg; t += WK) {     // same trip count in every K group (steps >= kt1 load zeros)
        wait_vmcnt<(S - 2) * LPS>();
        __builtin_amdgcn_s_barrier();
        issue(t + (S - 1) * WK, fill);
        const unsigned char* st = smem + slot * SB;
        bf16x8 a[2][TM], b[2][TN];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[kk][i] = frag(st, TA, LA, offA[i], kk);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[kk][j] = frag(st, TB, LB, offB[j], kk);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
            if (TA && TB && want_bsum) {
#pragma unroll
                for (int i = 0; i < TM; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], ones, accb[i], 0, 0, 0);
            }
        }
        slot = slot + 1 == S ? 0 : slot + 1;
        fill = fill + 1 == S ? 0 : fill + 1;
    }
    wait_vmcnt<0>();                      // the zero-page loads of the steps past kt1 still target the ring
    __builtin_amdgcn_s_barrier();

    // bias-gradient partial sums.  C/D map: col = lane & 15, row = (lane >> 4)*4 + reg; every column holds the sum
    if (TA && TB && want_bsum && fr == 0 && atom) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + wm0 + i * 16 + fq * 4 + r;
                if (row < M) atomicAdd(P.bias_grad + row, g.alpha * accb[i][r]);
            }
    }

    float* __restrict__ C = P.C;
    bf16_t* __restrict__ Ch = reinterpret_cast<bf16_t*>(P.Ch);
    const bool first = by == 0;
    // ---- epilogue through LDS: whole 16-byte pieces of C rows (fp32) and 8-byte pieces of the bf16 copy
    float* Ct = reinterpret_cast<float*>(smem);
    const int ldch = g.ldch ? g.ldch : g.ldc;
    const bool bsum_wg = TA && TB && g.ones_col && P.bias_grad != nullptr && col0 == 0;
#pragma unroll 1
    for (int band0 = 0; band0 < BM; band0 += BAND) {
        if (band0 > 0) __syncthreads();           // the previous band has been read
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rb = wm0 + i * 16 - band0;   // wave-uniform
            if (rb < 0 || rb >= BAND) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ct[(rb + fq * 4 + r) * CP + wn0 + j * 16 + fr] = acc[i][j][r];
            if (TA && TB && want_bsum && fr == 0) {       // the staging tile's first padding column carries the bias-gradient sums
#pragma unroll
                for (int r = 0; r < 4; ++r) Ct[(rb + fq * 4 + r) * CP + BN] = accb[i][r];
            }
        }
        __syncthreads();
        if (TA && TB && bsum_wg && !atom) {
            for (int rl = tid_all; rl < BAND; rl += NTALL) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < WK; ++w) sum += *reinterpret_cast<const float*>(smem_all + w * RING + (rl * CP + BN) * 4);
                if (row0 + band0 + rl < M) P.bias_grad[row0 + band0 + rl] = g.alpha * sum;
            }
        }
        constexpr int NCH = BAND * BN / 4;
        for (int q = tid_all; q < NCH; q += NTALL) {
            const int rl = q / (BN / 4), c4 = (q % (BN / 4)) * 4;
            const int row = row0 + band0 + rl, col = col0 + c4;
            if (row >= M || col >= N) continue;
            float4 a4 = *reinterpret_cast<const float4*>(smem_all + (rl * CP + c4) * 4);
#pragma unroll
            for (int w = 1; w < WK; ++w) {
                const float4 b4 = *reinterpret_cast<const float4*>(smem_all + w * RING + (rl * CP + c4) * 4);
                a4.x += b4.x; a4.y += b4.y; a4.z += b4.z; a4.w += b4.w;
            }
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
            const bool live = g.row_flag ? (g.row_flag[row / g.row_flag_div] != 0) : true;
            const int nv = min(4, N - col);
            float* dst = C ? C + (size_t)row * g.ldc + col : nullptr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e >= nv) break;
                float x = g.alpha * v[e];
                if (P.bias && first) x += P.bias[col + e];
                if (!live) x = 0.f;
                if (g.add_vec && first) x += g.add_vec[col + e];
                if (atom) {          // split-K / summed batches: the K groups are already summed; one fp32 atomic per element into a zeroed C
                    atomicAdd(dst + e, x);
                    continue;
                }
                if (P.Cpre) P.Cpre[(size_t)row * g.ldc + col + e] = x;        // the pre-activation (what a GELU backward needs)
                if (g.act == 1) x = fmaxf(x, 0.f);
                else if (g.act == 2) x = gelu_erf2(x);
                if (g.epi_drop.p > 0.f) x *= dropout_scale(g.epi_drop, g.epi_site, (uint64_t)row * N + col + e);
                if (g.relu_ref) {
                    const float rv = g.relu_ref[(size_t)row * g.ld_ref + col + e];
                    if (g.ref_kind == 2) x *= gelu_erf2_grad(rv);
                    else if (rv <= 0.f) x = 0.f;
                }
                if (g.accumulate) x += dst[e];
                v[e] = x;
            }
            if (atom) continue;
            if (dst) {
                if (nv == 4 && g.vecC) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                else for (int e = 0; e < nv; ++e) dst[e] = v[e];
            }
            if (Ch) {
                bf16_t* dh = Ch + (size_t)row * ldch + col;
                if (nv == 4 && g.vecB) {       // vecB doubles as "the bf16 result is 8-byte aligned" on this path
                    bf16x4 h;
                    h[0] = (bf16_t)v[0]; h[1] = (bf16_t)v[1]; h[2] = (bf16_t)v[2]; h[3] = (bf16_t)v[3];
                    *reinterpret_cast<bf16x4*>(dh) = h;
                } else {
                    for (int e = 0; e < nv; ++e) dh[e] = (bf16_t)v[e];
                }
            }
        }
    }
}

template <bool TA, bool TB, int BM, int BN, int WM, int WN, int S, int WK = 1>
__global__ __launch_bounds__(WM* WN* WK * 64) void gemm2_kernel(const GemmArgs g) {
    gemm2_body<TA, TB, BM, BN, WM, WN, S, WK>(g, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// Several products of different shapes in ONE launch (the text side's weight gradients at <= a few thousand rows: five TN
// products of 10 - 18 us each that wait on nothing but their dY -- a launch each was ~75 us of the backward's dependent chain).
// The grid is the concatenation of the sub-problems' tile lists; a workgroup looks its sub-problem up and runs the ordinary body.
constexpr int G2_GROUP_MAX = 6;
struct GemmGroupArgs {
    GemmArgs sub[G2_GROUP_MAX];
    int tile0[G2_GROUP_MAX + 1];      // first workgroup of each sub-problem
    int tiles_m[G2_GROUP_MAX], strip[G2_GROUP_MAX];      // tile rows; width (in tiles) of the column strips the tiles are walked in
    int n;
};
template <bool TA, bool TB, int BM, int BN, int WM, int WN, int S, int WK = 1>
__global__ __launch_bounds__(WM* WN* WK * 64) void gemm2_group_kernel(const GemmGroupArgs gg) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < G2_GROUP_MAX; ++i)
        if (i < gg.n && (int)blockIdx.x >= gg.tile0[i]) s = i;
    // XCD-aware placement.  Workgroup i runs on XCD i % 8 (round-robin dispatch), each XCD has its own L2, and a tile row / column of
    // a weight gradient is a K x 128 operand panel that every L2 touching it fetches again: in launch order the launch pulled 92 -
    // 105 MB through the fabric for 19 MB of operands.  So the member's workgroups that share an XCD take CONSECUTIVE tiles of a walk
    // that keeps neighbours close in both directions: column strips of `strip` tiles, row-major inside a strip (a run of q tiles
    // touches ~q / strip + strip panels instead of ~2 q): 60 MB (PMC FETCH_SIZE, cfg2 at 64 windows).
    const int t0 = gg.tile0[s], t1 = gg.tile0[s + 1], id = (int)blockIdx.x, x = id & 7;
    int start = 0, fx = 0;
#pragma unroll
    for (int xx = 0; xx < 8; ++xx) {
        const int f = t0 + ((xx - t0) & 7), c = f < t1 ? ((t1 - 1 - f) >> 3) + 1 : 0;
        if (xx < x) start += c;
        if (xx == x) fx = f;
    }
    const int L = start + ((id - fx) >> 3);
    const int w = gg.strip[s], tm = gg.tiles_m[s];
    const int st = L / (tm * w), r = L - st * tm * w, tile_m = r / w;
    gemm2_body<TA, TB, BM, BN, WM, WN, S, WK>(gg.sub[s], 0, 0, 0, t1 - t0, 1, tile_m, st * w + r - tile_m * w);
}

template <int BM, int BN, int WM, int WN, int S, int WK = 1>
int launch2(int layout, const GemmArgs& g_in, int Mmax, int splits, hipStream_t stream) {
    dim3 grid(cdiv(Mmax, BM) * cdiv(g_in.N, BN), splits, g_in.zbatch ? g_in.zbatch : g_in.nprob), block(WM * WN * WK * 64);
    if (grid.x == 0) return IMMTSF_OK;
    GemmArgs g = g_in;
    if (g.xcd_remap && g.xcd_gm) {        // exact 2-D XCD partition when the tile grid divides; else the 1-D contiguous ranges
        const int tiles_m = cdiv(Mmax, BM), tiles_n = cdiv(g.N, BN);
        int best = 0;
        long best_cost = 0;
        for (int XR = 1; XR <= 8; XR *= 2) {
            const int XC = 8 / XR;
            if (tiles_m % XR || tiles_n % XC) continue;
            const long cost = (long)(tiles_m / XR) * BM + (long)(tiles_n / XC) * BN;     // operand rows an XCD's L2 has to hold
            if (!best || cost < best_cost) { best = XR; best_cost = cost; }
        }
        g.xcd_gm = best;
    }
    {   // the kernel's tile-index arithmetic as shifts and multiplications (GemmArgs::g2_*)
        const int tiles_m = cdiv(Mmax, BM), tiles_n = cdiv(g.N, BN);
        auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
        g.g2_fast = grid.x < 65536 && tiles_n < 65536 && Mmax == g.M;
        g.g2_tiles_n = tiles_n;
        g.g2_tn_magic = magic(tiles_n);
        const int XR = g.xcd_gm > 0 ? g.xcd_gm : 1, XC = 8 / XR;
        g.g2_rows_x = tiles_m / XR;
        g.g2_cols_x = tiles_n / XC;
        g.g2_xc_shift = XC == 8 ? 3 : XC == 4 ? 2 : XC == 2 ? 1 : 0;
        g.g2_cx_magic = magic(g.g2_cols_x);
        g.g2_per = (g.dyn && g.dyn_which == 1) ? 0 : cdiv(cdiv(g.K, BK2), splits);
    }
    immtsf_gemm_note_grid((long)grid.x * grid.y * grid.z * block.x);
    switch (layout) {
        case GEMM_NT: hipLaunchKernelGGL((gemm2_kernel<false, false, BM, BN, WM, WN, S, WK>), grid, block, 0, stream, g); break;
        case GEMM_NN:
            if constexpr (BN % 32 == 0) hipLaunchKernelGGL((gemm2_kernel<false, true, BM, BN, WM, WN, S, WK>), grid, block, 0, stream, g);
            else return IMMTSF_EUNSUPPORTED;
            break;
        case GEMM_TN:
            if constexpr (BN % 32 == 0 && BM % 32 == 0) hipLaunchKernelGGL((gemm2_kernel<true, true, BM, BN, WM, WN, S, WK>), grid, block, 0, stream, g);
            else return IMMTSF_EUNSUPPORTED;
            break;
        default: return IMMTSF_EINVAL;
    }
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int g2_variant = 0, g2_splitk = 0, g2_xcd = -1;      // g2_xcd: -1 heuristic, 0 off, 1 contiguous ranges (1-D), 2 exact 2-D partition

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static int g2_group_tile = 0;       // tile edge of the grouped weight-gradient launch: 0 = by tile count, 64 | 128 forced (tool switch: variant 1064 / 1128 / 1000)
extern "C" int immtsf_debug_gemm2_config(int variant, int splitk, int xcd) {
    if (variant == 1000 || variant == 1064 || variant == 1128) { g2_group_tile = variant - 1000; return 0; }
    g2_variant = variant;
    g2_splitk = splitk;
    g2_xcd = xcd;
    return 0;
}

bool immtsf_gemm2_supported(int layout, const GemmArgs& g) {
    if (layout < 0 || layout > 2 || g.nprob < 1 || g.nprob > IMMTSF_GEMM_MAX_PROBLEMS || g.nbatch > 1) return false;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return false;
    if (g.zbatch < 0 || (g.zbatch && g.nprob != 1) || (g.atomic_c && (g.act || g.p[0].Ch || g.p[0].Cpre || !g.p[0].C))) return false;
    if (g.ref_kind && (g.ref_kind != 2 || !g.relu_ref)) return false;
    if ((g.epi_drop.p > 0.f || g.ref_kind) && (g.N & 3)) return false;       // (dropout / GELU' epilogues: round 3, for the FFN block)
    if ((g.lda % 8) || (g.ldb % 8)) return false;
    if (layout == GEMM_TN && (g.a_rowmap || g.b_rowmap)) return false;
    if (layout == GEMM_NN && g.b_rowmap) return false;
    // chunks are 8 elements along the contiguous dimension: the K tail of an R image and the column edge of a T image
    // are handled in whole chunks
    if (layout == GEMM_NT && (g.K % 8)) return false;
    if (layout == GEMM_NN && ((g.K % 8) || (g.N % 8))) return false;
    if (layout == GEMM_TN && ((g.M % 8) || (g.N % 8))) return false;
    if (g.dyn && g.dyn_which == 1 && layout != GEMM_TN) return false;      // a dynamic K must not cut a chunk
    if (g.dyn && g.dyn_which == 0 && layout == GEMM_TN) return false;
    for (int i = 0; i < g.nprob; ++i) {
        const GemmProblem& p = g.p[i];
        if (!p.Ah || !p.Bh || !al16(p.Ah) || !al16(p.Bh)) return false;
        if (!p.C && !p.Ch) return false;
        if (g.accumulate && !p.C) return false;
    }
    return true;
}

int immtsf_launch_gemm2(int layout, GemmArgs& g, hipStream_t stream) {
    if (!immtsf_gemm2_supported(layout, g)) return IMMTSF_EUNSUPPORTED;
    if (g.row_flag && g.row_flag_div <= 0) return IMMTSF_EINVAL;
    if (g.ones_col && layout != GEMM_TN) return IMMTSF_EINVAL;
    const int ldch = g.ldch ? g.ldch : g.ldc;
    bool vc = (g.ldc % 4) == 0, vh = (ldch % 4) == 0, any_h = false, all_c = true;
    for (int i = 0; i < g.nprob; ++i) {
        if (g.p[i].C) vc = vc && al16(g.p[i].C); else all_c = false;
        if (g.p[i].Ch) { any_h = true; vh = vh && ((reinterpret_cast<uintptr_t>(g.p[i].Ch) & 7) == 0); }
    }
    g.vecC = vc ? 1 : 0;
    g.vecB = vh ? 1 : 0;
    g.vecA = 1;
    const int Mmax = g.M;
    const int nz = g.zbatch ? g.zbatch : g.nprob;       // grid.z
    const long t64 = (long)cdiv(Mmax, 64) * cdiv(g.N, 64) * nz;

    // many rows (forward / data-gradient projections at >= 256 windows per GPU): the persistent ping-pong kernel of gemm3.hip,
    // 1.4-2x this file's tiles from ~190 row tiles of 128 on (profiles/r03_gemm_bigM.txt)
    constexpr int use_g3 = 1;
    if (use_g3 && g2_variant == 0 && g2_splitk <= 1 && layout != GEMM_TN && g.nprob == 1 && !g.zbatch && g.act == 0 && !g.relu_ref && !g.accumulate && g.epi_drop.p <= 0.f && !g.p[0].Cpre &&
        !g.a_rowmap && !g.b_rowmap && !g.ones_col && (!g.row_flag || g.row_flag32) && !(g.dyn && g.dyn_which != 0) &&
        (long)cdiv(Mmax, 128) * cdiv(g.N, 256) >= 192) {
        const GemmProblem& p = g.p[0];
        const int rc = immtsf_launch_gemm3(layout, p.Ah, g.lda, p.Bh, g.ldb, p.C, g.ldc, p.Ch, ldch, p.bias, g.add_vec,
                                           g.row_flag ? g.row_flag32 : nullptr, g.row_flag_div, Mmax, g.N, g.K, g.alpha, 0, g.dyn, stream);
        if (rc != IMMTSF_EUNSUPPORTED) return rc;
    }
    if (use_g3 && g2_variant == 0 && g2_splitk <= 1 && layout == GEMM_TN && g.nprob == 1 && !g.zbatch && g.ws && g.act == 0 && !g.relu_ref && !g.a_rowmap &&
        !g.b_rowmap && !(g.dyn && g.dyn_which != 1) && !g.row_flag && !g.add_vec && !g.p[0].bias && g.epi_drop.p <= 0.f && !g.p[0].Cpre) {
        const GemmProblem& p = g.p[0];
        const int rc = immtsf_launch_gemm3_tn(p.Ah, g.lda, p.Bh, g.ldb, p.C, g.ldc, p.Ch, ldch, g.ones_col ? p.bias_grad : nullptr, Mmax, g.N, g.K,
                                              g.alpha, g.accumulate, g.dyn, g.ws, g.ws_bytes, stream);
        if (rc != IMMTSF_EUNSUPPORTED) return rc;
    }

    // split-K over workgroups (fp32 atomics into a zeroed C) is only a tool option here: the K-group variants below split
    // the reduction INSIDE a workgroup and sum through LDS, which measured faster at every weight-gradient shape of the
    // fusion step (r02: 768x768x2048 13.1 us unsplit on 64x64 k4 vs 17.9 us as 2 atomic splits of the 4-wave tile)
    const bool can_split = !g.zbatch && all_c && !any_h && g.act == 0 && !g.relu_ref && g.epi_drop.p <= 0.f && !g.p[0].Cpre && (g.accumulate || g.ldc == g.N) &&
                           !(g.dyn && g.dyn_which == 0);
    int splits = 1;
    if (can_split && g2_splitk > 1) splits = g2_splitk;
    // ... except for LONG reductions on few tiles (weight gradients at >= 512 windows per GPU: 768x768x32768): there the 64x64
    // tiles are bound by L2 -> LDS bytes (1.2 GB for 100 MB of operands), 128x128 tiles halve that but leave 36 workgroups,
    // so the reduction is cut over ~256 / tiles workgroups: 151 -> 112 us (tools/gemm2_bench.py longk)
    bool long_k = false;
    if (can_split && g2_variant == 0 && g2_splitk <= 1 && layout == GEMM_TN && g.K >= 16384 && g.nprob == 1) {
        const long t128s = (long)cdiv(Mmax, 128) * cdiv(g.N, 128);
        if (t128s <= 256) {       // (768x4096x145k, cfg5's input-projection gradient, 192 tiles: 4249 us on 128x96 k2 -> 1770 us split 4)
            long_k = true;
            splits = t128s <= 72 ? (int)(256 / t128s) : 4;
            if (splits > 8) splits = 8;
            if (splits < 1) splits = 1;
        }
    }
    // ... and for SKINNY weight gradients (M <= 64 output rows, N >= 1024: TimesNet's merged convolution kernels, 16 / 32 x 3872 from ~4000
    // rows): one row of 64 x 64 tiles leaves three quarters of the chip idle for 32 us -- the reduction is cut four ways
    if (can_split && g2_variant == 0 && g2_splitk <= 1 && layout == GEMM_TN && g.nprob == 1 && Mmax <= 64 && g.N >= 1024 && g.K >= 2048 && !long_k)
        splits = 4;
    if ((splits > 1 || g.atomic_c) && !g.c_prezeroed && !g.accumulate) {
        for (int i = 0; i < g.nprob; ++i) {
            // (fill kernels, not memset nodes: see gemm.hip)
            if (int rc = launch_fill(g.p[i].C, 0.f, (size_t)Mmax * g.N, stream)) return rc;
            if (g.p[i].bias_grad)
                if (int rc = launch_fill(g.p[i].bias_grad, 0.f, (size_t)Mmax, stream)) return rc;
        }
    }
    {
        const bool b_fits_l2 = (size_t)g.N * g.K * 2 <= (size_t)3 << 20;
        // a device-side row count (ragged notes: about half of the allocation bound) leaves the XCDs that own the tail of the tile
        // rows idle under any contiguous / rectangular assignment: those launches keep the hardware's round-robin order
        const bool dyn_rows = g.dyn && g.dyn_which == 0;
        g.xcd_remap = (!dyn_rows && t64 >= 64 && (t64 < 2048 || b_fits_l2) && g.N >= 256) ? 8 : 0;
        g.xcd_gm = g.xcd_remap ? 1 : 0;       // launch2 turns the flag into the group height for its tile shape
        if (g2_xcd >= 0) { g.xcd_remap = g2_xcd ? 8 : 0; g.xcd_gm = g2_xcd >= 2 ? 1 : 0; }
    }
    int v = g2_variant;
    if (v == 0) {
        // Tile choice (r02 sweeps, profiles/r02_gemm2_sweep.txt).  The K-group variants (k4: 16 waves, 128-160 KB of LDS)
        // run ONE workgroup per CU, so they want a grid that fills the 256 CUs in a single round; when the rows are a
        // device-side count (ragged notes) the expected fill is about half the allocation bound.
        const int Meff = (g.dyn && g.dyn_which == 0) ? (Mmax * 9 + 15) / 16 : Mmax;
        const long n64 = (long)cdiv(Meff, 64) * cdiv(g.N, 64) * nz;
        const long n96 = (long)cdiv(Meff, 64) * cdiv(g.N, 96) * nz;
        const long t128 = (long)cdiv(Mmax, 128) * cdiv(g.N, 128) * nz;
        const bool n96_ok = g.N % 96 == 0 || g.N >= 960;
        if (long_k) v = 18;                                       // 128x128 k2 with the reduction split over workgroups
        else if (t128 >= 1024 && layout == GEMM_NT && g.N >= 1024) v = 13;     // 256x256, 8 waves, 2 stages (short-K, N = 768 and NN: 256x128 wins,
                                                                                // tools/gemm2_bench.py bigm)
        else if (t128 >= 512) v = 7;                              // 256x128
        else if (n64 <= 272) v = 17;                              // 64x64 k4
        else if (n96_ok && n96 <= 272) v = 16;                    // 64x96 k4
        // (128x96 k2 was the pick for 272 < tiles of 64x96 <= 544: it wins no shape of the r02 sweep -- 1117x1536x768 NT 13.3 us against
        // 9.7 on 64x64w8, 17.5 in the step with the device-side row count -- and is a tool-only variant now)
        else v = 9;                                               // 64x64, 8 waves, 4 stages, two workgroups per CU
    }
    switch (v) {
        case 1: return launch2<64, 64, 2, 2, 4>(layout, g, Mmax, splits, stream);
        case 2: return launch2<64, 96, 2, 2, 4>(layout, g, Mmax, splits, stream);
        case 3: return launch2<128, 64, 2, 2, 4>(layout, g, Mmax, splits, stream);
        case 4: return launch2<64, 128, 2, 2, 4>(layout, g, Mmax, splits, stream);
        case 5: return launch2<128, 128, 2, 2, 3>(layout, g, Mmax, splits, stream);
        case 6: return launch2<128, 128, 2, 4, 4>(layout, g, Mmax, splits, stream);
        case 7: return launch2<256, 128, 4, 2, 3>(layout, g, Mmax, splits, stream);
        case 8: return launch2<64, 64, 2, 2, 3>(layout, g, Mmax, splits, stream);
        case 9: return launch2<64, 64, 2, 4, 4>(layout, g, Mmax, splits, stream);
        case 10: return launch2<96, 64, 2, 2, 4>(layout, g, Mmax, splits, stream);
        case 11: return launch2<64, 96, 2, 2, 6>(layout, g, Mmax, splits, stream);
        case 12: return launch2<128, 96, 2, 2, 4>(layout, g, Mmax, splits, stream);
        case 13: return launch2<256, 256, 2, 4, 2>(layout, g, Mmax, splits, stream);
        case 14: return launch2<64, 96, 2, 2, 3, 2>(layout, g, Mmax, splits, stream);
        case 15: return launch2<64, 64, 2, 2, 3, 2>(layout, g, Mmax, splits, stream);
        case 16: return launch2<64, 96, 2, 2, 2, 4>(layout, g, Mmax, splits, stream);
        case 17: return launch2<64, 64, 2, 2, 2, 4>(layout, g, Mmax, splits, stream);
        case 18: return launch2<128, 128, 2, 2, 2, 2>(layout, g, Mmax, splits, stream);
        case 19: return launch2<96, 64, 2, 2, 3, 2>(layout, g, Mmax, splits, stream);
        case 20: return launch2<128, 64, 2, 2, 3, 2>(layout, g, Mmax, splits, stream);
        case 21: return launch2<64, 128, 2, 2, 3, 2>(layout, g, Mmax, splits, stream);
        case 22: return launch2<128, 96, 2, 2, 2, 2>(layout, g, Mmax, splits, stream);
        case 23: return launch2<64, 64, 2, 2, 4, 2>(layout, g, Mmax, splits, stream);
        case 24: return launch2<256, 128, 4, 2, 2>(layout, g, Mmax, splits, stream);
        case 25: return launch2<96, 96, 2, 2, 3, 2>(layout, g, Mmax, splits, stream);
        case 26: return launch2<64, 96, 2, 2, 2, 3>(layout, g, Mmax, splits, stream);
        case 27: return launch2<96, 64, 2, 2, 2, 4>(layout, g, Mmax, splits, stream);
        default: return IMMTSF_EINVAL;
    }
}

// On unless IMMTSF_GEMM_GROUP=0.  History: measured at 64 windows (cfg2) early in round 3 the five text-side weight gradients as one
// 936-workgroup launch at the end of the backward took 47 us alone on the critical stream, where the separate launches (10 - 18 us
// each) had overlapped with the backbone's backward: 0.731 vs 0.716 ms per step, and the switch stayed off.  At the end of the round
// the step is bound by the chip time its two branches share, and one launch that fills every CU once beats five that each
// hold 144 - 216 of them: 0.567 - 0.571 vs 0.581 - 0.589 ms.
bool immtsf_gemm_group_enabled() {
    constexpr bool on = true;
    return on;
}
// n (2 .. 6) TN products C_i = alpha A_i^T B_i (+ bias gradients) of different shapes as ONE launch of the 64 x 64 K-group
// tiles: every product must be one the single launch would run on those tiles without splitting (bf16 operands, fp32 result, no
// activation / row maps / row counts, K < 8192).  IMMTSF_EUNSUPPORTED otherwise: the caller launches them one by one.
int immtsf_launch_gemm2_group_tn(GemmArgs* list, int n, hipStream_t stream) {
    if (!immtsf_gemm_group_enabled() || n < 2 || n > G2_GROUP_MAX || g2_variant != 0 || g2_splitk > 1) return IMMTSF_EUNSUPPORTED;
    GemmGroupArgs gg;
    memset(&gg, 0, sizeof(gg));
    int tiles = 0;
    // 128 x 128 tiles (two K-groups of 2 x 2 waves) when they still give every CU most of a tile (>= 192 of them): a quarter of the
    // workgroups and half the operand traffic of the 64 x 64 K-group tiles (cfg2's five text-side weight gradients: 234 tiles in
    // one round instead of 936 in four; 0.546 vs 0.549 ms per step, HBM-side traffic per launch halved)
    int gt = g2_group_tile;
    if (gt == 0) {
        long t128 = 0;
        for (int i = 0; i < n; ++i) t128 += (long)cdiv(list[i].M, 128) * cdiv(list[i].N, 128);
        gt = t128 >= 192 ? 128 : 64;
    }
    // (measured and dropped, tools/group_bench.py at the cfg2 shapes: deeper rings -- 128 x 128 with 4 stages and one K group 50 - 64 us,
    // 64 x 64 with 8 waves and 4 stages 38 - 48 -- against 41 - 43 for the two tiles below; splitting the reduction of the longest
    // member, 2048 rows beside three ~1100-note ones, over two workgroups per tile: 34 us alone, but 234 workgroups of 128 KB of LDS
    // instead of 198 leave the backbone's branch fewer CUs and the step 1.5 % slower)
    for (int i = 0; i < n; ++i) {
        GemmArgs g = list[i];
        if (!immtsf_gemm2_supported(GEMM_TN, g) || g.nprob != 1 || g.act != 0 || g.relu_ref || g.row_flag || g.add_vec || g.accumulate ||
            g.K >= 8192 || (g.dyn && g.dyn_which != 1) || g.p[0].Ch || !g.p[0].C || (g.ldc % 4) || !al16(g.p[0].C))
            return IMMTSF_EUNSUPPORTED;
        g.vecA = 1; g.vecB = 1; g.vecC = 1;
        g.xcd_remap = 0; g.xcd_gm = 0; g.g2_fast = 0;
        const int tm = cdiv(g.M, gt), tn = cdiv(g.N, gt), q = tm * tn / 8 > 0 ? tm * tn / 8 : 1;
        int w = 1;          // strip width: the divisor of the tile columns closest to sqrt(tiles per XCD)
        for (int c = 2; c <= 4; ++c)
            if (tn % c == 0 && c * c <= 2 * q) w = c;
        gg.sub[i] = g;
        gg.tile0[i] = tiles;
        gg.tiles_m[i] = tm;
        gg.strip[i] = w;
        tiles += tm * tn;
    }
    // (four rounds of one-workgroup-per-CU tiles at most: with a wide member -- 768 x 4096 at LLaMA-width embeddings, 768 tiles of its
    // own -- the separate launches are as good or better: cfg3 1.349 grouped vs 1.331 ms)
    if (tiles > 1024) return IMMTSF_EUNSUPPORTED;
    gg.tile0[n] = tiles;
    for (int i = n + 1; i <= G2_GROUP_MAX; ++i) gg.tile0[i] = tiles;
    gg.n = n;
    immtsf_gemm_note_grid((long)tiles * (gt == 128 ? 512 : 1024));
    if (gt == 128)
        hipLaunchKernelGGL((gemm2_group_kernel<true, true, 128, 128, 2, 2, 2, 2>), dim3(tiles), dim3(512), 0, stream, gg);
    else
        hipLaunchKernelGGL((gemm2_group_kernel<true, true, 64, 64, 2, 2, 2, 4>), dim3(tiles), dim3(1024), 0, stream, gg);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// n (2 .. 6) weight-gradient products of different shapes as one launch (bench.py re-times the step's grouped launch through this)
extern "C" int immtsf_gemm_bf16_group_tn(int32_t n, const void* const* A, const int32_t* lda, const void* const* B, const int32_t* ldb,
                                         float* const* C, const int32_t* ldc, const int32_t* M, const int32_t* N, const int32_t* K,
                                         void* stream) {
    if (n < 2 || n > G2_GROUP_MAX || !A || !B || !C || !lda || !ldb || !ldc || !M || !N || !K) return IMMTSF_EINVAL;
    GemmArgs list[G2_GROUP_MAX];
    for (int i = 0; i < n; ++i) {
        GemmArgs& g = list[i];
        memset(&g, 0, sizeof(g));
        g.nprob = 1;
        g.M = M[i]; g.N = N[i]; g.K = K[i];
        g.lda = lda[i]; g.ldb = ldb[i]; g.ldc = ldc[i];
        g.alpha = 1.f;
        g.row_flag_div = 1;
        g.nbatch = 1;
        g.batch_inner = 1;
        g.p[0].Ah = A[i];
        g.p[0].Bh = B[i];
        g.p[0].C = C[i];
    }
    return immtsf_launch_gemm2_group_tn(list, n, static_cast<hipStream_t>(stream));
}

// debug / test / tool entry (declared in include/immtsf.h)
extern "C" int immtsf_gemm_bf16(int32_t layout, const void* A, int32_t lda, const void* B, int32_t ldb, float* C, int32_t ldc,
                                void* Ch, int32_t ldch, const float* bias, float* bias_grad, int32_t M, int32_t N, int32_t K,
                                float alpha, int32_t accumulate, int32_t act, const int32_t* dyn, int32_t dyn_which,
                                const int32_t* a_rowmap, void* stream) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.nprob = 1;
    g.M = M; g.N = N; g.K = K;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldch = ldch;
    g.alpha = alpha;
    g.accumulate = accumulate;
    g.act = act;
    g.row_flag_div = 1;
    g.nbatch = 1;
    g.batch_inner = 1;
    g.dyn = dyn;
    g.dyn_which = dyn_which;
    g.a_rowmap = a_rowmap;
    g.p[0].Ah = A;
    g.p[0].Bh = B;
    g.p[0].C = C;
    g.p[0].Ch = Ch;
    g.p[0].bias = bias;
    g.p[0].bias_grad = bias_grad;
    if (bias_grad) g.ones_col = 1;
    return immtsf_launch_gemm2(layout, g, static_cast<hipStream_t>(stream));
}
