// LDS-tiled MFMA GEMM for gfx950.  One kernel template serves the three layouts of a linear layer's
// forward (NT), data gradient (NN) and weight gradient (TN), in two arithmetic modes:
//   precision 0: exact fp32 on v_mfma_f32_16x16x4_f32   (parity mode: k-ordered fmaf chain, no rounding of inputs)
//   precision 1: bf16 operands, fp32 accumulate on v_mfma_f32_16x16x32_bf16 (operands are rounded RNE while
//                they are staged into LDS; HBM tensors stay fp32)
//
// Structure (64-lane wavefronts, WM x WN waves per workgroup, each wave owning a (BM/WM)x(BN/WN) block of 16x16
// accumulators):
//   * tiles go global -> registers -> LDS; LDS is double-buffered so a K-step costs ONE barrier, and the global
//     loads of tile t+2 are in flight while tile t+1 is being computed;
//   * an operand whose reduction index is contiguous in memory is staged as [row][k] (k contiguous, padded by one
//     16-byte slot) and its MFMA fragment is one ds_read_b128;
//   * an operand whose reduction index is the SLOW one (TN/NN layouts) is staged untransposed as [k][row] with
//     8-byte LDS writes and its fragment is fetched with two ds_read_b64_tr_b16 (hardware transpose read) -- no
//     scattered 2-byte LDS writes;
//   * small outputs with a long reduction (weight gradients) are split along K over blockIdx.y and combined with
//     fp32 atomics into a zero-initialised C (the launcher zero-fills);
//   * arbitrary M/N/K by zero-filling tile edges; M or K may live in device memory (ragged note count), so no host
//     synchronisation is needed to size the launch.
#include "gemm.hpp"
#include "rowops.hpp"
#include <atomic>
#include <mutex>
#include <shared_mutex>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <bool BF16> struct LdsElem { typedef float T; };
template <> struct LdsElem<true> { typedef bf16_t T; };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
// d/dz of the above: Phi(z) + z phi(z)
__device__ __forceinline__ float gelu_erf_grad(float z) {
    return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
}

// LDS image of one operand tile: R = [rows][BK+pad] (k contiguous), KM = [BK][rows+8] (row contiguous, bf16 only)
template <bool BF16, bool TR, int ROWS, int BK> struct TileGeom {
    static constexpr bool kmajor = BF16 && TR;
    static constexpr int pad = BF16 ? 8 : 1;
    static constexpr int pitch = kmajor ? (ROWS + 8) : (BK + pad);
    static constexpr int elems = kmajor ? BK * pitch : ROWS * pitch;
};

// SPEC (wave specialisation): the workgroup has 2*WM*WN waves.  Waves [0, WM*WN) only read fragments and issue MFMAs;
// waves [WM*WN, 2*WM*WN) only move data (global -> registers -> bf16 -> LDS), two tiles ahead in two register sets.
// The phases of one K step (load wait, convert + LDS fill, LDS fragment reads + MFMA) then overlap across waves
// instead of running back to back in every wave between two barriers (r01f/r01g ablations: at 2048x768x768 the three
// phases cost 2.4 + 1.1 + 2.0 us and simply add up in the unspecialised kernel).
// BBF: the B operand is read from its bf16 twin (GemmProblem::Bh): half as many 16-byte load instructions through the
// L1 -- the resource that bounds the K loop at the fusion shapes (r01g: time is linear in K at 11.8 ns per K unit for
// 2048x768xK whatever the tile shape, wave count or overlap scheme) -- no conversion, one 16-byte LDS write per chunk.
template <bool BF16, bool TA, bool TB, int BM, int BN, int BK, int WM, int WN, bool SPEC = false, bool BBF = false>
__global__ __launch_bounds__((SPEC ? 2 : 1) * WM * WN * 64) void gemm_kernel(const GemmArgs g) {
    static_assert(!BBF || (BF16 && !SPEC), "bf16-twin operands: bf16 MFMA path, unspecialised pipeline");
    typedef typename LdsElem<BF16>::T T;
    typedef TileGeom<BF16, TA, BM, BK> GA;
    typedef TileGeom<BF16, TB, BN, BK> GB;
    constexpr int NTC = WM * WN * 64;                 // compute threads
    constexpr int NT = SPEC ? 2 * NTC : NTC;          // all threads
    constexpr int NL = NTC;                           // loader threads (SPEC: the upper half; else everybody)
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int CA = (BM * BK / 4) / NL, CB = (BN * BK / 4) / NL;
    constexpr int CBH = BBF ? (BN * BK / 8) / NL : 1;      // 8-element (16-byte) chunks of a bf16 operand per thread
    static_assert(!BBF || ((BN * BK / 8) % NL == 0 && CBH >= 1), "bf16 chunking must be exact");
    static_assert(CA >= 1 && CB >= 1, "tile too small for the thread count");
    static_assert((BM * BK / 4) % NL == 0 && (BN * BK / 4) % NL == 0, "chunking must be exact");
    constexpr int BUF = ((GA::elems + GB::elems) * (int)sizeof(T) + 15) / 16 * 16 / (int)sizeof(T);

    __shared__ __attribute__((aligned(16))) T smem[2 * BUF];

    const int nb = g.nbatch > 1 ? g.nbatch : 1;
    const GemmProblem P = g.p[blockIdx.z / nb];
    const int bi = blockIdx.z % nb;
    long offA = 0, offB = 0, offC = 0;
    if (g.nbatch > 1) {
        const int bo = bi / g.batch_inner, bin = bi % g.batch_inner;
        offA = bo * g.sA_o + bin * g.sA_i;
        offB = bo * g.sB_o + bin * g.sB_i;
        offC = bo * g.sC_o + bin * g.sC_i;
    }
    int M = g.M, K = g.K;
    const int N = g.N;
    const int Nreal = g.N;
    if (g.dyn) {
        const int dv = *g.dyn;
        if (g.dyn_which == 0) M = dv; else K = dv;
    }
    const int tiles_n = (N + BN - 1) / BN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (bid % 8 labels the L2 they share), so
    // give every XCD one CONTIGUOUS range of tile indices -- a few tile rows of A plus the whole B panel then fit its
    // 4 MiB L2 instead of all eight L2s streaming all of A and B from the Infinity Cache.  Bijective for any grid size.
    int lin = blockIdx.x;
    if (g.xcd_remap) {
        const int total = gridDim.x, q = total >> 3, r = total & 7, xcd = lin & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    }
    int tile_m = lin / tiles_n, tile_n = lin % tiles_n;
    if (g.xcd_remap && g.xcd_remap != 8) {
        // 2-D variant: the XCD's range of positions walks a RECTANGLE of tiles (xm x xn = 8 rectangles, enumerated one
        // after the other), so an L2 holds tiles_m/xm row panels of A and tiles_n/xn panels of B instead of a sliver of A
        // and ALL of B -- which is what overflows 4 MiB when the reduction is long (weight gradients: r01g PMC, 59 MB
        // fetched for 12.6 MB of operands at 768x768x2048).  Still a bijection for any grid.
        const int tiles_m = gridDim.x / tiles_n, xm = g.xcd_remap, xn = 8 / xm;
        int p = lin;
        for (int rct = 0; rct < 8; ++rct) {
            const int mi = rct / xn, ni = rct - mi * xn;
            const int m0 = tiles_m * mi / xm, m1 = tiles_m * (mi + 1) / xm, n0 = tiles_n * ni / xn, n1 = tiles_n * (ni + 1) / xn;
            const int wn = n1 - n0, cnt = (m1 - m0) * wn;
            if (p < cnt) { tile_m = m0 + p / wn; tile_n = n0 + p % wn; break; }
            p -= cnt;
        }
    }
    const int row0 = tile_m * BM, col0 = tile_n * BN;
    if (row0 >= M) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool is_loader = !SPEC || tid >= NTC, is_compute = !SPEC || tid < NTC;
    const int ltid = SPEC ? (is_loader ? tid - NTC : 0) : tid;        // loader-relative thread index
    const int wm0 = ((wave % (WM * WN)) / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const float* __restrict__ A = P.A + offA;
    const float* __restrict__ Bp = P.B + offB;
    const bf16_t* __restrict__ Bh = reinterpret_cast<const bf16_t*>(P.Bh) + offB;      // BBF only
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 rbh[CBH];
    bool b_h = false;          // the staged B registers currently hold bf16 chunks (rbh) rather than floats (rb)

    float4 ra[CA], rb[CB];

    auto load_slow = [&](int k0, float4 (&ra)[CA], float4 (&rb)[CB]) {
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const int c = ltid + i * NL;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!TA) {
                const int r = c / (BK / 4), kq = c % (BK / 4);
                const int grow = row0 + r, gk = k0 + kq * 4;
                if (grow < M && gk < K) {
                    const int srow = g.a_rowmap ? g.a_rowmap[grow] : grow;
                    const float* src = A + (size_t)srow * g.lda + gk;
                    if (g.vecA && gk + 3 < K) v = *reinterpret_cast<const float4*>(src);
                    else {
                        v.x = src[0];
                        if (gk + 1 < K) v.y = src[1];
                        if (gk + 2 < K) v.z = src[2];
                        if (gk + 3 < K) v.w = src[3];
                    }
                }
            } else {
                const int kk = c / (BM / 4), rq = c % (BM / 4);
                const int gk = k0 + kk, grow = row0 + rq * 4;
                if (gk < K && grow < M) {
                    const int sk = g.a_rowmap ? g.a_rowmap[gk] : gk;
                    const float* src = A + (size_t)sk * g.lda + grow;
                    if (g.vecA && grow + 3 < M) v = *reinterpret_cast<const float4*>(src);
                    else {
                        v.x = src[0];
                        if (grow + 1 < M) v.y = src[1];
                        if (grow + 2 < M) v.z = src[2];
                        if (grow + 3 < M) v.w = src[3];
                    }
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int c = ltid + i * NL;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!TB) {
                const int r = c / (BK / 4), kq = c % (BK / 4);
                const int gcol = col0 + r, gk = k0 + kq * 4;
                if (BBF && gcol < N && gk < K) {
                    const bf16_t* src = Bh + (size_t)gcol * g.ldb + gk;
                    v.x = (float)src[0];
                    if (gk + 1 < K) v.y = (float)src[1];
                    if (gk + 2 < K) v.z = (float)src[2];
                    if (gk + 3 < K) v.w = (float)src[3];
                } else if (gcol < N && gk < K) {
                    const float* src = Bp + (size_t)gcol * g.ldb + gk;
                    if (g.vecB && gk + 3 < K) v = *reinterpret_cast<const float4*>(src);
                    else {
                        v.x = src[0];
                        if (gk + 1 < K) v.y = src[1];
                        if (gk + 2 < K) v.z = src[2];
                        if (gk + 3 < K) v.w = src[3];
                    }
                }
            } else {
                const int kk = c / (BN / 4), rq = c % (BN / 4);
                const int gk = k0 + kk, gcol = col0 + rq * 4;
                if (BBF && gk < K && gcol < N) {
                    const int sk = g.b_rowmap ? g.b_rowmap[gk] : gk;
                    const bf16_t* src = Bh + (size_t)sk * g.ldb + gcol;
                    v.x = (float)src[0];
                    if (gcol + 1 < N) v.y = (float)src[1];
                    if (gcol + 2 < N) v.z = (float)src[2];
                    if (gcol + 3 < N) v.w = (float)src[3];
                } else if (gk < K && gcol < N) {
                    const int sk = g.b_rowmap ? g.b_rowmap[gk] : gk;
                    const float* src = Bp + (size_t)sk * g.ldb + gcol;
                    if (g.vecB && gcol + 3 < Nreal) v = *reinterpret_cast<const float4*>(src);
                    else {
                        v.x = (gcol < Nreal) ? src[0] : 1.f;
                        if (gcol + 1 < N) v.y = (gcol + 1 < Nreal) ? src[1] : 1.f;
                        if (gcol + 2 < N) v.z = (gcol + 2 < Nreal) ? src[2] : 1.f;
                        if (gcol + 3 < N) v.w = (gcol + 3 < Nreal) ? src[3] : 1.f;
                    }
                }
            }
            rb[i] = v;
        }
        b_h = false;
    };

    // ---- fast path: straight-line 16-byte loads, no bounds checks, addresses = uniform base + per-lane 32-bit
    // offset + uniform k advance.  Valid when both operands are 16-byte aligned, the contiguous dimension of every
    // chunk is fully inside the matrix and the K tile is full.  Rows past M / N of a row-major operand are clamped
    // to the last valid row (their products only feed output rows/cols that are never stored).
    bool fast = g.vecA && g.vecB && !(TA && g.a_rowmap) && !(TB && g.b_rowmap);
    if (TA) fast = fast && (row0 + BM <= M);
    if (TB) fast = fast && (col0 + BN <= Nreal);
    unsigned oa[CA], ob[CB];
    if (fast) {
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const int c = ltid + i * NL;
            if (!TA) {
                const int r = min(row0 + c / (BK / 4), M - 1), kq = c % (BK / 4);
                const int srow = g.a_rowmap ? g.a_rowmap[r] : r;
                oa[i] = (unsigned)srow * (unsigned)g.lda + kq * 4;
            } else {
                oa[i] = (unsigned)(c / (BM / 4)) * (unsigned)g.lda + row0 + (c % (BM / 4)) * 4;
            }
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int c = ltid + i * NL;
            if (!TB) {
                const int r = min(col0 + c / (BK / 4), Nreal - 1), kq = c % (BK / 4);
                ob[i] = (unsigned)r * (unsigned)g.ldb + kq * 4;
            } else {
                ob[i] = (unsigned)(c / (BN / 4)) * (unsigned)g.ldb + col0 + (c % (BN / 4)) * 4;
            }
        }
    }
    unsigned obh[CBH];
    if (BBF && fast) {
#pragma unroll
        for (int i = 0; i < CBH; ++i) {
            const int c = ltid + i * NL;
            if (!TB) {
                const int r = min(col0 + c / (BK / 8), Nreal - 1), kq = c % (BK / 8);
                obh[i] = (unsigned)r * (unsigned)g.ldb + kq * 8;
            } else {
                obh[i] = (unsigned)(c / (BN / 8)) * (unsigned)g.ldb + col0 + (c % (BN / 8)) * 8;
            }
        }
    }
    auto load_fast = [&](int k0, float4 (&ra)[CA], float4 (&rb)[CB]) {
        const unsigned ka = TA ? (unsigned)k0 * (unsigned)g.lda : (unsigned)k0;
        const unsigned kb = TB ? (unsigned)k0 * (unsigned)g.ldb : (unsigned)k0;
#pragma unroll
        for (int i = 0; i < CA; ++i) ra[i] = *reinterpret_cast<const float4*>(A + (oa[i] + ka));
        if (BBF) {
#pragma unroll
            for (int i = 0; i < CBH; ++i) rbh[i] = *reinterpret_cast<const u32x4*>(Bh + (obh[i] + kb));
            b_h = true;
        } else {
#pragma unroll
            for (int i = 0; i < CB; ++i) rb[i] = *reinterpret_cast<const float4*>(Bp + (ob[i] + kb));
        }
    };
    // write one 4-element chunk of a staged tile.  `transposed`: the 4 values run along the tile's ROW index.
    auto put4 = [&](T* base, bool transposed, bool kmajor, int pitch, int BR, int c, const float4& v) {
        if (!transposed) {
            const int r = c / (BK / 4), kq = c % (BK / 4);
            T* dst = base + r * pitch + kq * 4;
            if (BF16) {
                bf16x4 h;
                h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
                *reinterpret_cast<bf16x4*>(dst) = h;
            } else {
                dst[0] = (T)v.x; dst[1] = (T)v.y; dst[2] = (T)v.z; dst[3] = (T)v.w;
            }
        } else if (kmajor) {   // [k][row]: rows contiguous -> one 8-byte write
            const int kk = c / (BR / 4), rq = c % (BR / 4);
            bf16x4 h;
            h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
            *reinterpret_cast<bf16x4*>(base + kk * pitch + rq * 4) = h;
        } else {               // fp32 parity mode: scattered transposing writes into [row][k]
            const int kk = c / (BR / 4), rq = c % (BR / 4);
            T* dst = base + (rq * 4) * pitch + kk;
            dst[0] = (T)v.x; dst[pitch] = (T)v.y; dst[2 * pitch] = (T)v.z; dst[3 * pitch] = (T)v.w;
        }
    };
    // TN + bias gradient: bias_grad[m] = sum_k A[k][m] (the column sums of dY) is gathered from the A chunks on their way
    // to LDS by the workgroups of the FIRST tile column -- no extra tile column, no extra pass over dY
    const bool want_bsum = TA && TB && g.ones_col && P.bias_grad != nullptr && col0 == 0;
    constexpr int RCH = BM / 4;                                      // 4-row chunks per k-line of the A tile
    constexpr bool BS1 = (NL % RCH == 0) && (64 % RCH == 0);        // all chunks of a thread cover the same 4 rows
    constexpr int NBS = (TA && TB) ? (BS1 ? 1 : CA) : 1;
    float4 bsum[NBS];
#pragma unroll
    for (int i = 0; i < NBS; ++i) bsum[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto store_tile = [&](int buf, const float4 (&ra)[CA], const float4 (&rb)[CB]) {
        T* As = smem + buf * BUF;
        T* Bs = As + GA::elems;
        if (TA && TB && want_bsum) {
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                float4& b = bsum[BS1 ? 0 : i];
                b.x += ra[i].x; b.y += ra[i].y; b.z += ra[i].z; b.w += ra[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < CA; ++i) put4(As, TA, GA::kmajor, GA::pitch, BM, ltid + i * NL, ra[i]);
        if (BBF && b_h) {          // bf16 chunks go to LDS as they are: one 16-byte write each
#pragma unroll
            for (int i = 0; i < CBH; ++i) {
                const int c = ltid + i * NL;
                bf16_t* dst = reinterpret_cast<bf16_t*>(Bs) +
                              (TB ? (c / (BN / 8)) * GB::pitch + (c % (BN / 8)) * 8 : (c / (BK / 8)) * GB::pitch + (c % (BK / 8)) * 8);
                *reinterpret_cast<u32x4*>(dst) = rbh[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < CB; ++i) put4(Bs, TB, GB::kmajor, GB::pitch, BN, ltid + i * NL, rb[i]);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    // hardware-transpose read of a 16(row) x 8(k) bf16 fragment from a [k][row] image (T10 of the CDNA4 guide):
    // within each 16-lane group, lane 4q+p supplies the address of k-line q, rows 4p..4p+3; lane i receives row i.
    auto frag_kmajor = [&](const T* tile, int pitch, int rbase, int kbase) -> bf16x8 {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const int q = fr >> 2, p = fr & 3;
        const T* a0 = tile + (kbase + fq * 8 + q) * pitch + rbase + 4 * p;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * pitch));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };

    // fragments of one staged tile (all BK/32 k-groups) -> registers.  Split from the MFMAs so that, for the k-major
    // images, the hardware-transpose reads can be issued BEFORE the next tile's global loads: the compiler guards
    // ds_read_b64_tr_b16 with a conservative s_waitcnt vmcnt(0), which would otherwise drain those loads in front of
    // the MFMAs they are meant to overlap (seen in the r01f ISA of the NN/TN kernels).
    constexpr int KG = BF16 ? BK / 32 : 1;
    auto read_frags = [&](int buf, bf16x8 (&a)[KG][TM], bf16x8 (&b)[KG][TN]) {
        const T* As = smem + buf * BUF;
        const T* Bs = As + GA::elems;
#pragma unroll
        for (int kk = 0; kk < KG; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (GA::kmajor) a[kk][i] = frag_kmajor(As, GA::pitch, wm0 + i * 16, kk * 32);
                else a[kk][i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(As) + (wm0 + i * 16 + fr) * GA::pitch + kk * 32 + fq * 8);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (GB::kmajor) b[kk][j] = frag_kmajor(Bs, GB::pitch, wn0 + j * 16, kk * 32);
                else b[kk][j] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(Bs) + (wn0 + j * 16 + fr) * GB::pitch + kk * 32 + fq * 8);
            }
        }
    };
    auto mfma_frags = [&](const bf16x8 (&a)[KG][TM], const bf16x8 (&b)[KG][TN]) {
#pragma unroll
        for (int kk = 0; kk < KG; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
    };
    auto compute = [&](int buf) {
        const T* As = smem + buf * BUF;
        const T* Bs = As + GA::elems;
        if (BF16) {
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (GA::kmajor) a[i] = frag_kmajor(As, GA::pitch, wm0 + i * 16, kk * 32);
                    else a[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(As) + (wm0 + i * 16 + fr) * GA::pitch + kk * 32 + fq * 8);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (GB::kmajor) b[j] = frag_kmajor(Bs, GB::pitch, wn0 + j * 16, kk * 32);
                    else b[j] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(Bs) + (wn0 + j * 16 + fr) * GB::pitch + kk * 32 + fq * 8);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = reinterpret_cast<const float*>(As)[(wm0 + i * 16 + fr) * GA::pitch + kk * 4 + fq];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = reinterpret_cast<const float*>(Bs)[(wn0 + j * 16 + fr) * GB::pitch + kk * 4 + fq];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // K range of this split (in BK tiles).  The fast pipeline covers the FULL tiles; a ragged last tile (dynamic
    // K = note count) is handled after the loop with the checked loader, so the hot loop stays straight-line.
    const int nk_total = (K + BK - 1) / BK;
    const int splits = gridDim.y;
    const int per = (nk_total + splits - 1) / splits;
    const int kt0 = blockIdx.y * per, kt1 = min(nk_total, kt0 + per);
    if (splits > 1 && kt0 >= kt1) return;     // nothing to add

    auto pipeline = [&](int t0, int t1, auto&& loader) {
        if (t0 < t1) {
            loader(t0 * BK, ra, rb);
            store_tile(0, ra, rb);
        }
        __syncthreads();
        for (int t = t0; t < t1; ++t) {
            const int cur = (t - t0) & 1;
            if (BF16 && (GA::kmajor || GB::kmajor) && KG <= 2) {   // deeper K tiles: too many live fragments
                bf16x8 fa[KG][TM], fb[KG][TN];
                read_frags(cur, fa, fb);
                __builtin_amdgcn_sched_barrier(0);               // fragment reads stay in front of the global loads
                if (t + 1 < t1 && !(g.dbg & 2)) loader((t + 1) * BK, ra, rb);
                if (!(g.dbg & 1)) mfma_frags(fa, fb);
            } else {
                if (t + 1 < t1 && !(g.dbg & 2)) loader((t + 1) * BK, ra, rb);   // issue early: in flight under this tile's MFMAs
                if (!(g.dbg & 1)) compute(cur);
            }
            __builtin_amdgcn_sched_barrier(0);                // keep the LDS writes (and their vmcnt wait) behind the MFMAs
            if (t + 1 < t1 && !(g.dbg & 4)) store_tile(cur ^ 1, ra, rb);    // other buffer: last read one barrier ago
            __syncthreads();
        }
    };
    // wave-specialised pipeline over the full tiles [t0, t1): producers keep two tiles in flight (register sets ra/rb and
    // ra2/rb2), consumers only see LDS.  One barrier per K step; buffer (t+1)&1 is filled while buffer t&1 is read.
    auto pipeline_spec = [&](int t0, int t1) {
        float4 ra2[CA], rb2[CB];
        if (is_loader) {
            if (t0 < t1) {
                load_fast(t0 * BK, ra, rb);
                store_tile(0, ra, rb);
            }
            if (t0 + 1 < t1) load_fast((t0 + 1) * BK, ra2, rb2);          // tile t0+1 -> set 2
        }
        __syncthreads();
        int t = t0;
        for (; t + 3 < t1; t += 2) {       // steady state: no conditionals between the loads and their consumers
            if (is_loader) {
                load_fast((t + 2) * BK, ra, rb);       // two tiles ahead
                store_tile(1, ra2, rb2);               // tile t+1 (in flight since the previous step)
            } else {
                compute(0);
            }
            __syncthreads();
            if (is_loader) {
                load_fast((t + 3) * BK, ra2, rb2);
                store_tile(0, ra, rb);                 // tile t+2
            } else {
                compute(1);
            }
            __syncthreads();
        }
        // drain: at most 3 tiles left (t .. t1-1); tile t is in buffer 0, tile t+1 (if any) is in set 2
        for (int u = t; u < t1; ++u) {
            const int cur = (u - t) & 1;
            if (is_loader) {
                if (u + 1 < t1) {
                    if (cur == 0) {
                        if (u + 2 < t1) load_fast((u + 2) * BK, ra, rb);
                        store_tile(1, ra2, rb2);
                    } else {
                        store_tile(0, ra, rb);
                    }
                }
            } else {
                compute(cur);
            }
            __syncthreads();
        }
    };
    // tiles the fast loader cannot take (ragged K tail, edge tiles of transposed operands, unaligned operands) in a
    // specialised workgroup: producers fill, barrier, consumers compute, barrier -- correct, not overlapped, rare
    auto pipeline_spec_slow = [&](int t0, int t1) {
        for (int t = t0; t < t1; ++t) {
            if (is_loader) {
                load_slow(t * BK, ra, rb);
                store_tile(0, ra, rb);
            }
            __syncthreads();
            if (is_compute) compute(0);
            __syncthreads();
        }
    };
    if (SPEC) {
        if (fast) {
            const int kfull = min(kt1, K / BK);
            pipeline_spec(kt0, kfull);
            if (kfull < kt1) pipeline_spec_slow(kfull > kt0 ? kfull : kt0, kt1);
        } else {
            pipeline_spec_slow(kt0, kt1);
        }
    } else if (fast) {
        const int kfull = min(kt1, K / BK);                   // tiles [kt0, kfull) are complete
        pipeline(kt0, kfull, load_fast);
        if (kfull < kt1) pipeline(kfull > kt0 ? kfull : kt0, kt1, load_slow);   // at most one ragged tile
    } else {
        pipeline(kt0, kt1, load_slow);
    }

    if (TA && TB && g.ones_col) {          // block-uniform branch (the barriers below are reached by every thread)
        if (col0 == 0 && P.bias_grad) {
            float* bs = reinterpret_cast<float*>(smem);     // staging tiles are dead after the last barrier
            for (int i = tid; i < BM; i += NT) bs[i] = 0.f;
            __syncthreads();
            if (is_loader) {
                constexpr int R = RCH;
                if (BS1) {
                    // every chunk of a thread covers the same 4 rows (rq = lane % R): fold the lanes that share rq with
                    // shuffles, one LDS atomic per wave and row
                    float4 t = bsum[0];
                    t.x = coset_sum(t.x, R); t.y = coset_sum(t.y, R); t.z = coset_sum(t.z, R); t.w = coset_sum(t.w, R);
                    if (lane < R) {
                        atomicAdd(bs + lane * 4 + 0, t.x);
                        atomicAdd(bs + lane * 4 + 1, t.y);
                        atomicAdd(bs + lane * 4 + 2, t.z);
                        atomicAdd(bs + lane * 4 + 3, t.w);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NBS; ++i) {
                        const int rq = (ltid + i * NL) % R;
                        atomicAdd(bs + rq * 4 + 0, bsum[i].x);
                        atomicAdd(bs + rq * 4 + 1, bsum[i].y);
                        atomicAdd(bs + rq * 4 + 2, bsum[i].z);
                        atomicAdd(bs + rq * 4 + 3, bsum[i].w);
                    }
                }
            }
            __syncthreads();
            for (int i = tid; i < BM; i += NT) {
                const int row = row0 + i;
                if (row < M) {
                    if (gridDim.y > 1) atomicAdd(P.bias_grad + row, g.alpha * bs[i]);
                    else P.bias_grad[row] = g.alpha * bs[i];
                }
            }
            __syncthreads();
        }
    }

    // epilogue: C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* __restrict__ C = P.C + offC;
    const bool first = blockIdx.y == 0;

    // ---- vector epilogue: stage the accumulator tile through LDS (free after the last barrier) and store whole
    // 16-byte chunks of C rows -- 16 lanes cover one 256-byte row segment instead of 64-byte slivers.
    constexpr int CP = BN + 4;
    constexpr bool kCtFits = (size_t)BM * CP * sizeof(float) <= (size_t)2 * BUF * sizeof(T);
    if (kCtFits && g.vecC && splits == 1) {
        float* Ct = reinterpret_cast<float*>(smem);
        if (is_compute) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Ct[(wm0 + i * 16 + fq * 4 + r) * CP + wn0 + j * 16 + fr] = acc[i][j][r];
        }
        __syncthreads();
        constexpr int NCH = BM * BN / 4;
        for (int q = tid; q < NCH; q += NT) {
            const int rl = q / (BN / 4), c4 = (q % (BN / 4)) * 4;
            const int row = row0 + rl, col = col0 + c4;
            if (row >= M || col >= N) continue;
            const float4 a4 = *reinterpret_cast<const float4*>(Ct + rl * CP + c4);
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
            const bool live = g.row_flag ? (g.row_flag[row / g.row_flag_div] != 0) : true;
            const int nv = min(4, N - col);
            float* dst = P.C ? C + (size_t)row * g.ldc + col : nullptr;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e >= nv) break;
                float x = g.alpha * v[e];
                if (P.bias) x += P.bias[col + e];
                if (!live) x = 0.f;
                if (g.add_vec) x += g.add_vec[col + e];
                if (P.Cpre) P.Cpre[(size_t)row * g.ldc + col + e] = x;
                if (g.act == 1) x = fmaxf(x, 0.f);
                else if (g.act == 2) x = gelu_erf(x);
                if (g.epi_drop.p > 0.f) x *= dropout_scale(g.epi_drop, g.epi_site, (uint64_t)row * N + col + e);
                if (g.relu_ref) {
                    const float rv = g.relu_ref[(size_t)row * g.ld_ref + col + e];
                    if (g.ref_kind == 2) x *= gelu_erf_grad(rv);
                    else if (rv <= 0.f) x = 0.f;
                }
                if (g.accumulate) x += dst[e];
                v[e] = x;
            }
            if (dst) {
                if (nv == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                else for (int e = 0; e < nv; ++e) dst[e] = v[e];
            }
            if (P.Ch) {          // optional bf16 copy of the result for a bf16-in-memory consumer (gemm2.hip)
                bf16_t* dh = reinterpret_cast<bf16_t*>(P.Ch) + offC + (size_t)row * (g.ldch ? g.ldch : g.ldc) + col;
                for (int e = 0; e < nv; ++e) dh[e] = (bf16_t)v[e];
            }
        }
        return;
    }
    if (!is_compute) return;

#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + wm0 + i * 16 + fq * 4 + r;
            if (row >= M) continue;
            const bool live = g.row_flag ? (g.row_flag[row / g.row_flag_div] != 0) : true;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col0 + wn0 + j * 16 + fr;
                if (col >= N) continue;
                float v = g.alpha * acc[i][j][r];
                if (P.bias && first) v += P.bias[col];
                if (!live) v = 0.f;
                if (g.add_vec && first) v += g.add_vec[col];
                float* dst = C + (size_t)row * g.ldc + col;
                if (splits > 1) {
                    atomicAdd(dst, v);
                } else {
                    if (P.Cpre) P.Cpre[(size_t)row * g.ldc + col] = v;
                    if (g.act == 1) v = fmaxf(v, 0.f);
                    else if (g.act == 2) v = gelu_erf(v);
                    if (g.epi_drop.p > 0.f) v *= dropout_scale(g.epi_drop, g.epi_site, (uint64_t)row * N + col);
                    if (g.relu_ref) {                                  // activation-backward factor
                        const float rv = g.relu_ref[(size_t)row * g.ld_ref + col];
                        if (g.ref_kind == 2) v *= gelu_erf_grad(rv);
                        else if (rv <= 0.f) v = 0.f;
                    }
                    if (g.accumulate) v += *dst;
                    if (P.C) *dst = v;
                    if (P.Ch) reinterpret_cast<bf16_t*>(P.Ch)[offC + (size_t)row * (g.ldch ? g.ldch : g.ldc) + col] = (bf16_t)v;
                }
            }
        }
    }
}

thread_local long g_last_grid_threads = 0;   // for the timing tap: lets bench.py match a launch with rocprof's Grid_Size

template <bool BF16, int BM, int BN, int BK, int WM, int WN, bool SPEC = false, bool BBF = false>
int launch_cfg(int layout, const GemmArgs& g, int Mmax, int splits, hipStream_t stream) {
    const int Nlog = g.N;
    dim3 grid(cdiv(Mmax, BM) * cdiv(Nlog, BN), splits, g.nprob * (g.nbatch > 1 ? g.nbatch : 1)), block((SPEC ? 2 : 1) * WM * WN * 64);
    if (grid.x == 0) return IMMTSF_OK;
    g_last_grid_threads = (long)grid.x * grid.y * grid.z * block.x;
    switch (layout) {
        case GEMM_NT: hipLaunchKernelGGL((gemm_kernel<BF16, false, false, BM, BN, BK, WM, WN, SPEC, BBF>), grid, block, 0, stream, g); break;
        case GEMM_NN: hipLaunchKernelGGL((gemm_kernel<BF16, false, true, BM, BN, BK, WM, WN, SPEC, BBF>), grid, block, 0, stream, g); break;
        case GEMM_TN:
            if (BBF) return IMMTSF_EINVAL;      // weights are never the B operand of a weight-gradient GEMM
            hipLaunchKernelGGL((gemm_kernel<BF16, true, true, BM, BN, BK, WM, WN, SPEC, false>), grid, block, 0, stream, g);
            break;
        default: return IMMTSF_EINVAL;
    }
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

static int g_force_old = 0;      // immtsf_debug_gemm_config bit 16: keep bf16 mode on the round-1 kernel (A/B measurements)

// tuning override for tools/gemm_bench.py (0 = heuristic).  Process-wide A/B switches: set them while no call is in flight
int g_variant = 0, g_splitk = 0, g_xcd = 1, g_dbg = 0, g_tn_spec = 1, g_xcd2d = 0;

}  // namespace

void immtsf_gemm_note_grid(long threads) { g_last_grid_threads = threads; }

extern int g_immtsf_ttcn_fused;      // ttcn.hip
extern int g_immtsf_ttcn_bwd_grid;   // ttcn_full.hip

extern "C" int immtsf_debug_gemm_config(int variant, int splitk) {
    g_immtsf_ttcn_fused = (variant & 0x8000) ? 0 : 1;   // bit 15: TTCN on the streaming formulation (A/B measurements)
    g_variant = variant & 0xff;
    g_xcd = (variant & 0x100) ? 0 : 1;      // bit 8 disables the XCD-aware tile order (A/B measurements)
    g_dbg = (variant >> 9) & 15;            // ablation: 1 skip MFMA, 2 skip global loads, 4 skip LDS stores (results wrong)
    g_tn_spec = (variant & 0x2000) ? 0 : 1;  // bit 13 disables the specialised weight-gradient choice (A/B measurements)
    g_xcd2d = (variant & 0x4000) ? 1 : 0;    // bit 14: allow the 2-D XCD order (measured slower at the fusion shapes: off)
    g_force_old = (variant & 0x10000) ? 1 : 0;
    g_immtsf_ttcn_bwd_grid = (variant >> 17) & 1023;      // bits 17..26: persistent workgroups of the on-chip TTCN backward (0 = rule)
    g_splitk = splitk;
    return 0;
}

// ---- optional per-launch timing tap (bench.py's roofline leg): hipEvents bracket every GEMM launch on the
// stream it is launched on.  Off by default; while it is on, launches take g_tap_mu one at a time.
namespace {
struct TapRec { hipEvent_t e0, e1; int meta[10]; int members[24]; };      // members: (M, N, K, dyn) of up to 6 products of a grouped launch
constexpr int kTapCap = 16384;
TapRec* g_tap = nullptr;
int g_tap_n = 0, g_tap_events = 0;
std::atomic<int> g_tap_on{0};
std::mutex g_tap_mu;
}  // namespace

static int launch_gemm_impl(int layout, int precision, GemmArgs& g, hipStream_t stream);

// ---- bf16 twin registry (process-wide, empty by default): a handful of ranges, linear lookup under a reader lock;
// register / unregister take the writer lock, so a lookup sees a range entirely or not at all
namespace {
struct TwinRange { const float* base; const unsigned short* twin; size_t count; };
constexpr int kMaxTwins = 16;
TwinRange g_twins[kMaxTwins];
int g_ntwins = 0;
std::atomic<int> g_twins_on{1};
std::shared_mutex g_twins_mu;
}  // namespace

const void* immtsf_twin_lookup(const float* p, size_t min_elems) {
    if (!g_twins_on.load(std::memory_order_relaxed)) return nullptr;
    std::shared_lock<std::shared_mutex> lk(g_twins_mu);
    for (int i = 0; i < g_ntwins; ++i) {
        const TwinRange& r = g_twins[i];
        if (p >= r.base && p + min_elems <= r.base + r.count) return r.twin + (p - r.base);
    }
    return nullptr;
}

extern "C" int immtsf_bf16_twin_register(const float* base, void* twin, size_t count) {
    if (!base || !twin || count == 0) return IMMTSF_EINVAL;
    std::unique_lock<std::shared_mutex> lk(g_twins_mu);
    for (int i = 0; i < g_ntwins; ++i)
        if (g_twins[i].base == base) { g_twins[i] = TwinRange{base, static_cast<const unsigned short*>(twin), count}; return IMMTSF_OK; }
    if (g_ntwins >= kMaxTwins) return IMMTSF_EUNSUPPORTED;
    g_twins[g_ntwins++] = TwinRange{base, static_cast<const unsigned short*>(twin), count};
    return IMMTSF_OK;
}

extern "C" int immtsf_bf16_twin_unregister(const float* base) {
    std::unique_lock<std::shared_mutex> lk(g_twins_mu);
    for (int i = 0; i < g_ntwins; ++i)
        if (g_twins[i].base == base) { g_twins[i] = g_twins[--g_ntwins]; return IMMTSF_OK; }
    return IMMTSF_OK;
}

extern "C" int immtsf_bf16_twin_enable(int on) { g_twins_on.store(on ? 1 : 0); return IMMTSF_OK; }

// bf16 mode: a launch whose every problem carries a bf16 A (p.Ah) and a bf16 B (p.Bh, or a registered twin of p.B) goes
// to the bf16-in-memory kernel (gemm2.hip) when that kernel implements the argument combination; everything else (fp32
// parity mode, batched form, row-mapped weight gradients, K or N below the 8-element chunk) runs on the kernel below,
// which needs the fp32 operands.
static thread_local int g_last_path = 0;       // for the timing tap: 1 = the kernel in this file, 2 = gemm2.hip (bf16 operands in memory)
static int route_gemm(int layout, int precision, GemmArgs& g, hipStream_t stream) {
    if (precision == 1 && g.nbatch <= 1 && !g_force_old) {
        bool have = true;
        for (int i = 0; i < g.nprob && have; ++i) {
            GemmProblem& p = g.p[i];
            if (!p.Ah) { have = false; break; }
            if (!p.Bh && p.B && layout != GEMM_TN) {
                const size_t span = layout == GEMM_NT ? (size_t)(g.N - 1) * g.ldb + g.K : (size_t)(g.K - 1) * g.ldb + g.N;
                p.Bh = immtsf_twin_lookup(p.B, span);
            }
            if (!p.Bh) have = false;
        }
        if (have && immtsf_gemm2_supported(layout, g)) {
            g_last_path = 2;
            return immtsf_launch_gemm2(layout, g, stream);
        }
    }
    g_last_path = 1;
    for (int i = 0; i < g.nprob; ++i)
        if (!g.p[i].A || !g.p[i].B || (!g.p[i].C && !g.p[i].Ch) || (!g.p[i].C && g.accumulate))
            return IMMTSF_EUNSUPPORTED;      // bf16-only operands that gemm2 cannot take
    return launch_gemm_impl(layout, precision, g, stream);
}

int immtsf_launch_gemm(int layout, int precision, GemmArgs& g, hipStream_t stream) {
    if (!g_tap_on.load(std::memory_order_relaxed)) return route_gemm(layout, precision, g, stream);
    std::lock_guard<std::mutex> lk(g_tap_mu);
    if (!g_tap || g_tap_n >= kTapCap) return route_gemm(layout, precision, g, stream);
    TapRec& r = g_tap[g_tap_n];
    if (g_tap_n >= g_tap_events) {
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return route_gemm(layout, precision, g, stream);
        g_tap_events = g_tap_n + 1;
    }
    const int m[8] = {layout, precision, g.M, g.N, g.K, g.nprob, g.nbatch > 1 ? g.nbatch : 1, g.dyn ? 1 + g.dyn_which : 0};
    for (int i = 0; i < 8; ++i) r.meta[i] = m[i];
    (void)hipEventRecord(r.e0, stream);
    const int rc = route_gemm(layout, precision, g, stream);
    (void)hipEventRecord(r.e1, stream);
    r.meta[8] = (int)g_last_grid_threads;
    r.meta[9] = g_last_path;
    ++g_tap_n;
    return rc;
}

int immtsf_launch_gemm_tn_list(int precision, GemmArgs* list, int n, hipStream_t stream) {
    if (n <= 0) return IMMTSF_OK;
    // one grouped launch when every product has bf16 operands in memory and fits the grouped kernel (and no timing tap is
    // recording per-launch times)
    if (precision == 1 && n >= 2 && !g_force_old) {
        bool have = true;
        for (int i = 0; i < n && have; ++i) have = list[i].nprob == 1 && list[i].nbatch <= 1 && list[i].p[0].Ah && list[i].p[0].Bh;
        if (have && !g_tap_on.load(std::memory_order_relaxed)) {
            const int rc = immtsf_launch_gemm2_group_tn(list, n, stream);
            if (rc != IMMTSF_EUNSUPPORTED) return rc;
        } else if (have) {
            // the timing tap records the grouped launch as ONE record (layout code 3, path 3, nprob = members; their shapes in the
            // record's member table), so that bench.py ranks and re-times the launch that runs in the step
            std::lock_guard<std::mutex> lk(g_tap_mu);
            if (g_tap && g_tap_n < kTapCap && n <= 6) {
                TapRec& r = g_tap[g_tap_n];
                bool ev = true;
                if (g_tap_n >= g_tap_events) {
                    ev = hipEventCreate(&r.e0) == hipSuccess && hipEventCreate(&r.e1) == hipSuccess;
                    if (ev) g_tap_events = g_tap_n + 1;
                }
                if (ev) {
                    (void)hipEventRecord(r.e0, stream);
                    const int rc = immtsf_launch_gemm2_group_tn(list, n, stream);
                    if (rc != IMMTSF_EUNSUPPORTED) {
                        (void)hipEventRecord(r.e1, stream);
                        const int m[10] = {3, precision, list[0].M, list[0].N, list[0].K, n, 1, 0, (int)g_last_grid_threads, 3};
                        for (int i = 0; i < 10; ++i) r.meta[i] = m[i];
                        for (int i = 0; i < 6; ++i) {
                            r.members[4 * i] = i < n ? list[i].M : 0;
                            r.members[4 * i + 1] = i < n ? list[i].N : 0;
                            r.members[4 * i + 2] = i < n ? list[i].K : 0;
                            r.members[4 * i + 3] = i < n && list[i].dyn ? 1 : 0;
                        }
                        ++g_tap_n;
                        return rc;
                    }
                }
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        const int rc = immtsf_launch_gemm(GEMM_TN, precision, list[i], stream);
        if (rc != 0) return rc;
    }
    return IMMTSF_OK;
}

extern "C" int immtsf_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(g_tap_mu);
    if (on && !g_tap) g_tap = new TapRec[kTapCap];
    g_tap_on.store(on ? 1 : 0);
    g_tap_n = 0;
    return 0;
}

// host arrays: meta[10*max] (layout, precision, M, N, K, nprob, nbatch, dyn, grid threads, path), ms[max], members[24*max] (optional)
extern "C" int immtsf_timing_collect(int max, int* meta, float* ms, int* members) {
    std::lock_guard<std::mutex> lk(g_tap_mu);
    const int n = g_tap_n < max ? g_tap_n : max;
    for (int i = 0; i < n; ++i) {
        (void)hipEventSynchronize(g_tap[i].e1);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_tap[i].e0, g_tap[i].e1);
        ms[i] = t;
        for (int k = 0; k < 10; ++k) meta[10 * i + k] = g_tap[i].meta[k];
        if (members)
            for (int k = 0; k < 24; ++k) members[24 * i + k] = g_tap[i].meta[0] == 3 ? g_tap[i].members[k] : 0;
    }
    g_tap_n = 0;
    return n;
}

// ---- skinny products: a reduction or an output dimension of <= 16 (MMF_XAttn_Add's query projection Y (BT, C) -> (BT, d)
// and its data gradient back to C columns, C = 8 at the benchmark configuration).  As 64 x 64 MFMA tiles these are ~11 us
// of padding and pipeline fill each, on the serial section of the step between the two forward and the two backward
// branches; as plain fp32 FMAs they are bound by their 6 MB of output / input: ~3-4 us.  Exact fp32 in both precision modes.
namespace {
constexpr int SKINNY = 16;
// K == KB (8 or 16, rows 16-byte aligned): C[m, n] = rowflag(alpha sum_k A[m, k] opB(n, k) + bias[n]); a thread owns 4
// consecutive n of one row m; A's row and (NT) B's rows come in as float4s, no per-tap branches
template <bool TB, int KB>
__global__ __launch_bounds__(256) void skinny_k_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                        float* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N,
                                                        float alpha, const unsigned char* __restrict__ row_flag, int flag_div) {
    const int n4 = N >> 2;                       // N % 4 == 0 (host-checked)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)M * n4) return;
    const int m = (int)(i / n4), n0 = (int)(i % n4) * 4;
    float a[KB];
#pragma unroll
    for (int k4 = 0; k4 < KB / 4; ++k4) {
        const float4 v = reinterpret_cast<const float4*>(A + (size_t)m * lda)[k4];
        a[4 * k4] = v.x; a[4 * k4 + 1] = v.y; a[4 * k4 + 2] = v.z; a[4 * k4 + 3] = v.w;
    }
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (TB) {                                    // B[k][n]: one float4 of 4 consecutive n per k
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const float4 b = *reinterpret_cast<const float4*>(B + (size_t)k * ldb + n0);
            o[0] = fmaf(a[k], b.x, o[0]); o[1] = fmaf(a[k], b.y, o[1]); o[2] = fmaf(a[k], b.z, o[2]); o[3] = fmaf(a[k], b.w, o[3]);
        }
    } else {                                     // B[n][k]: KB / 4 float4s per n
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k4 = 0; k4 < KB / 4; ++k4) {
                const float4 b = reinterpret_cast<const float4*>(B + (size_t)(n0 + j) * ldb)[k4];
                o[j] = fmaf(a[4 * k4], b.x, fmaf(a[4 * k4 + 1], b.y, fmaf(a[4 * k4 + 2], b.z, fmaf(a[4 * k4 + 3], b.w, o[j]))));
            }
    }
    const bool live = row_flag ? row_flag[m / flag_div] != 0 : true;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = live ? alpha * o[j] + (bias ? bias[n0 + j] : 0.f) : 0.f;
    *reinterpret_cast<float4*>(C + (size_t)m * ldc + n0) = make_float4(o[0], o[1], o[2], o[3]);
}
// NN, N == NB (8 or 16): C[m, n] (+)= alpha sum_k A[m, k] B[k, n]; a wave owns one row m, lanes stride over k with NB
// accumulators each (B's row k is NB / 4 float4s), butterfly over the wave at the end
template <int NB>
__global__ __launch_bounds__(256) void skinny_n_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                        float* __restrict__ C, int ldc, int M, int K, float alpha, int accumulate) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float acc[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[n] = 0.f;
#pragma unroll 4
    for (int k = lane; k < K; k += 64) {
        const float a = A[(size_t)m * lda + k];
#pragma unroll
        for (int n4 = 0; n4 < NB / 4; ++n4) {
            const float4 b = reinterpret_cast<const float4*>(B + (size_t)k * ldb)[n4];
            acc[4 * n4] = fmaf(a, b.x, acc[4 * n4]); acc[4 * n4 + 1] = fmaf(a, b.y, acc[4 * n4 + 1]);
            acc[4 * n4 + 2] = fmaf(a, b.z, acc[4 * n4 + 2]); acc[4 * n4 + 3] = fmaf(a, b.w, acc[4 * n4 + 3]);
        }
    }
#pragma unroll
    for (int n = 0; n < NB; ++n)
        acc[n] = wave_sum(acc[n]);
    if (lane < NB) {
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n) v = lane == n ? acc[n] : v;
        float* dst = C + (size_t)m * ldc + lane;
        *dst = alpha * v + (accumulate ? *dst : 0.f);
    }
}
inline bool skinny_plain(const GemmArgs& g) {
    return g.nprob == 1 && g.nbatch <= 1 && !g.dyn && !g.a_rowmap && !g.b_rowmap && !g.add_vec && g.act == 0 && !g.relu_ref &&
           g.epi_drop.p <= 0.f && !g.p[0].Ch && !g.p[0].Cpre && !g.p[0].bias_grad && g.p[0].A && g.p[0].B && g.p[0].C;
}
}  // namespace

static int launch_gemm_impl(int layout, int precision, GemmArgs& g, hipStream_t stream) {
    if (g.nprob < 1 || g.nprob > IMMTSF_GEMM_MAX_PROBLEMS) return IMMTSF_EINVAL;
    if (g.M < 0 || g.N < 0 || g.K < 0) return IMMTSF_EINVAL;
    if (g.M == 0 || g.N == 0) return IMMTSF_OK;
    if (g_variant == 0 && skinny_plain(g) && g.M >= 256) {
        const uintptr_t al = reinterpret_cast<uintptr_t>(g.p[0].A) | reinterpret_cast<uintptr_t>(g.p[0].B) | reinterpret_cast<uintptr_t>(g.p[0].C);
        const bool al16 = (al & 15) == 0 && (g.lda & 3) == 0 && (g.ldb & 3) == 0 && (g.ldc & 3) == 0;
        if (al16 && (layout == GEMM_NT || layout == GEMM_NN) && (g.K == 8 || g.K == 16) && g.N >= 64 && (g.N & 3) == 0 && !g.accumulate) {
            const long thr = (long)g.M * (g.N / 4);
            const dim3 grid((unsigned)((thr + 255) / 256));
            const int fd = g.row_flag_div > 0 ? g.row_flag_div : 1;
#define IMMTSF_SKINNY_K(TB, KB)                                                                                                      \
    hipLaunchKernelGGL((skinny_k_kernel<TB, KB>), grid, dim3(256), 0, stream, g.p[0].A, g.lda, g.p[0].B, g.ldb, g.p[0].C, g.ldc, g.p[0].bias, \
                       g.M, g.N, g.alpha, g.row_flag, fd)
            if (layout == GEMM_NT) { if (g.K == 8) IMMTSF_SKINNY_K(false, 8); else IMMTSF_SKINNY_K(false, 16); }
            else { if (g.K == 8) IMMTSF_SKINNY_K(true, 8); else IMMTSF_SKINNY_K(true, 16); }
#undef IMMTSF_SKINNY_K
            IMMTSF_LAUNCH_CHECK();
            return IMMTSF_OK;
        }
        if (al16 && layout == GEMM_NN && (g.N == 8 || g.N == 16) && g.K >= 64 && !g.p[0].bias && !g.row_flag) {
            if (g.N == 8)
                hipLaunchKernelGGL(skinny_n_kernel<8>, dim3(cdiv(g.M, 4)), dim3(256), 0, stream, g.p[0].A, g.lda, g.p[0].B, g.ldb, g.p[0].C, g.ldc,
                                   g.M, g.K, g.alpha, g.accumulate);
            else
                hipLaunchKernelGGL(skinny_n_kernel<16>, dim3(cdiv(g.M, 4)), dim3(256), 0, stream, g.p[0].A, g.lda, g.p[0].B, g.ldb, g.p[0].C, g.ldc,
                                   g.M, g.K, g.alpha, g.accumulate);
            IMMTSF_LAUNCH_CHECK();
            return IMMTSF_OK;
        }
    }
    // vector-load eligibility is a property of every problem's base pointers and the leading dims
    bool va = (g.lda % 4) == 0, vb = (g.ldb % 4) == 0;
    for (int i = 0; i < g.nprob; ++i) {
        va = va && ((reinterpret_cast<uintptr_t>(g.p[i].A) & 15) == 0);
        vb = vb && ((reinterpret_cast<uintptr_t>(g.p[i].B) & 15) == 0);
    }
    if (g.nbatch > 1) {
        va = va && (g.sA_o % 4 == 0) && (g.sA_i % 4 == 0);
        vb = vb && (g.sB_o % 4 == 0) && (g.sB_i % 4 == 0);
    }
    g.vecA = va ? 1 : 0;
    g.vecB = vb ? 1 : 0;
    bool vc = (g.ldc % 4) == 0;
    bool all_c = true;       // split-K needs an fp32 C to add into and cannot serve a bf16 copy
    for (int i = 0; i < g.nprob; ++i) {
        vc = vc && ((reinterpret_cast<uintptr_t>(g.p[i].C) & 15) == 0);
        all_c = all_c && g.p[i].C != nullptr && g.p[i].Ch == nullptr;
    }
    if (g.nbatch > 1) vc = vc && (g.sC_o % 4 == 0) && (g.sC_i % 4 == 0);
    g.vecC = vc ? 1 : 0;
    g.dbg = g_dbg;
    {   // XCD-aware order pays when several L2s would otherwise stream the same mid-sized operands; it hurts once
        // the grid is large enough that the default order already keeps every XCD on its own tile rows
        const long gx = (long)cdiv(g.M, 64) * cdiv(g.N, 64);
        g.xcd_remap = 0;
        // ... unless the whole B operand fits one L2 beside a few A panels (N x K <= 3 MB of fp32, e.g. 768 x 768): then
        // the contiguous order also wins on large grids (r01j, 32768x768x768: NT 123 -> 85 us, NN 169 -> 144 us; 4096^3,
        // whose B is 64 MB, loses 30 % with it and keeps the default order)
        const bool b_fits_l2 = (size_t)g.N * g.K * sizeof(float) <= (size_t)3 << 20;
        // (a device-side row count -- ragged notes, ~half of the allocation bound -- leaves the XCDs that own the tail of the
        // tile rows with nothing to do under a contiguous order: those launches keep the round-robin order)
        const bool dyn_rows = g.dyn && g.dyn_which == 0;
        if (g_xcd && !dyn_rows && gx >= 64 && (gx < 2048 || b_fits_l2) && g.N >= 256) {
            // 8 = contiguous tile rows per XCD; 4 / 2 / 1 = xm of the 2-D variant: least (A panels + B panels) per L2
            const int tm = cdiv(g.M, 64), tn = cdiv(g.N, 64);
            int best = 8;
            float cost = (float)tm / 8.f + (float)tn;
            for (int xm = 4; xm >= 1; xm >>= 1) {
                const float c = (float)tm / (float)xm + (float)tn / (float)(8 / xm);
                if (c < cost * 0.85f) { cost = c; best = xm; }
            }
            g.xcd_remap = (g_xcd2d || best == 8) ? best : 8;
        }
    }
    if (g.row_flag && g.row_flag_div <= 0) return IMMTSF_EINVAL;
    if (g.nbatch > 1 && g.batch_inner <= 0) return IMMTSF_EINVAL;
    if (g.ones_col && layout != GEMM_TN) return IMMTSF_EINVAL;
    const int Mmax = g.M;   // g.M is the allocation-time upper bound when `dyn` overrides M
    const int nbz = g.nprob * (g.nbatch > 1 ? g.nbatch : 1);

    // ---- split-K: only for plain overwrite epilogues on a dense C (the launcher zero-fills it)
    const bool can_split = all_c && !g.no_split && g.act == 0 && !g.relu_ref && g.epi_drop.p <= 0.f && g.nbatch <= 1 && (g.accumulate || g.ldc == g.N) && !(g.dyn && g.dyn_which == 0);
    int splits = 1;
    // weight-gradient GEMMs with 96..230 output tiles (768x768 .. 768x1152 at the fusion dims): the wave-specialised
    // kernel, unsplit -- its consumer waves never wait on global loads (the k-major fragment reads carry a conservative
    // vmcnt(0)), no atomics, nothing to pre-zero.  r01g: 15.4 vs 19.4 us isolated at 768x1152x1117, step 1.46 -> 1.43 ms
    // (it only pays since the bias gradient stopped being a virtual tile column, whose edge tiles ran the checked loader).
    bool tn_spec = false;
    if (layout == GEMM_TN && precision == 1 && g_variant == 0 && g_tn_spec && va && vb && !g.a_rowmap && !g.b_rowmap && g.nbatch <= 1) {
        const long tiles = (long)cdiv(Mmax, 64) * cdiv(g.N, 64) * nbz;
        tn_spec = tiles >= 96 && tiles <= 230 && g.K >= 256;
    }
    if (can_split && !tn_spec) {
        if (g_splitk > 0) splits = g_splitk;
        else {
            const long tiles = (long)cdiv(Mmax, 64) * cdiv(g.N, 64) * nbz;
            const int ksteps = cdiv(g.K, 64);
            if ((tiles < 192 && ksteps >= 8) || (tiles < 256 && ksteps >= 32)) {   // r01g: at 200+ tiles the 8-wave tiles beat split-K
                splits = (int)(512 / tiles);        // all workgroups co-resident (2 per CU): no tail round
                if (splits > ksteps / 2) splits = ksteps / 2;
                if (splits > 512) splits = 512;
                if (splits < 1) splits = 1;
            }
        }
    }
    if (splits > 1 && !g.c_prezeroed && !g.accumulate) {
        for (int i = 0; i < g.nprob; ++i) {
            // a fill KERNEL, not hipMemsetAsync: captured into a hipGraph, the memset node left C dirty on the second and later replays
            // (ROCm 7.2; tests/test_gpu_train.py::test_split_k_gemm_zero_fill_survives_graph_replay: split-K 128 x 16 x 2048, replay 0 exact, replays 1.. off by 1.0) -- which no trainer-owned
            // step ever hit (their split-K outputs are pre-zeroed gradient sinks), but the graph-cached drop-in seam did
            if (int rc = launch_fill(g.p[i].C, 0.f, (size_t)Mmax * g.N, stream)) return rc;
            if (g.p[i].bias_grad)
                if (int rc = launch_fill(g.p[i].bias_grad, 0.f, (size_t)Mmax, stream)) return rc;
        }
    }

    // bf16 twin of the B operand (weights; forward NT and data-gradient NN): every problem of the launch must have one
    bool bbf = false;
    if (precision == 1 && layout != GEMM_TN && g.nbatch <= 1 && !g.b_rowmap) {
        bbf = true;
        for (int i = 0; i < g.nprob && bbf; ++i) {
            const size_t span = layout == GEMM_NT ? (size_t)(g.N - 1) * g.ldb + g.K : (size_t)(g.K - 1) * g.ldb + g.N;
            if (!g.p[i].Bh) g.p[i].Bh = immtsf_twin_lookup(g.p[i].B, span);
            bbf = g.p[i].Bh != nullptr;
        }
        if (bbf) {
            bool vb8 = (g.ldb % 8) == 0;
            for (int i = 0; i < g.nprob; ++i) vb8 = vb8 && ((reinterpret_cast<uintptr_t>(g.p[i].Bh) & 15) == 0);
            if (!vb8) bbf = false;      // misaligned twin: stay on the fp32 operand (its own vecB flag is already set)
        }
    }
    if (precision == 1) {
        int v = g_variant;
        if (v == 0) {
            const long tiles128 = (long)cdiv(Mmax, 128) * cdiv(g.N, 128) * nbz;
            const bool dyn_k = g.dyn && g.dyn_which == 1;
            // below ~512 big tiles the grid cannot fill 256 CUs with 128x128 tiles: 64x64 tiles with EIGHT waves
            // (more wavefronts per CU to cover the load -> LDS -> MFMA latency chain), 128-deep K tiles when every
            // K tile is then full (r01g sweep: 10-25 % over the 4-wave 64x64x64 tile at the fusion shapes)
            if (tn_spec) v = 15;
            else if (tiles128 >= 512) v = 4;
            else if (layout == GEMM_TN || dyn_k || (g.K % 128) != 0) v = 11;
            else v = 14;
            // (variant 17, a 64x96 tile = exactly one workgroup per CU at 2048x768, is 8 % faster in isolation -- 11.4 vs
            // 12.5 us -- but its 87 KB of LDS leaves no room for the other stream's kernels on the CU: the step gets 1 %
            // slower, r01j A/B 1.190 vs 1.176 ms.  Kept for tools/gemm_twin_bench.py only.)
            // reductions that are a multiple of 32 but not of 64 (padded small-model dims): 32-deep K tiles keep every
            // tile on the straight-line loader
            if (!dyn_k && (g.K % 64) != 0 && (g.K % 32) == 0 && g.K <= 512) v = 7;
        }
        switch (v) {
            case 1: return launch_cfg<true, 64, 64, 64, 2, 2>(layout, g, Mmax, splits, stream);
            case 2: return launch_cfg<true, 64, 64, 128, 2, 2>(layout, g, Mmax, splits, stream);
            case 3: return launch_cfg<true, 128, 64, 64, 2, 2>(layout, g, Mmax, splits, stream);
            case 4:
                if (bbf) return launch_cfg<true, 128, 128, 64, 2, 2, false, true>(layout, g, Mmax, splits, stream);
                return launch_cfg<true, 128, 128, 64, 2, 2>(layout, g, Mmax, splits, stream);
            case 5: return launch_cfg<true, 32, 64, 64, 2, 2>(layout, g, Mmax, splits, stream);
            case 6: return launch_cfg<true, 64, 128, 64, 2, 2>(layout, g, Mmax, splits, stream);
            case 7:
                if (bbf) return launch_cfg<true, 64, 64, 32, 2, 2, false, true>(layout, g, Mmax, splits, stream);
                return launch_cfg<true, 64, 64, 32, 2, 2>(layout, g, Mmax, splits, stream);
            case 11:
                if (bbf) return launch_cfg<true, 64, 64, 64, 2, 4, false, true>(layout, g, Mmax, splits, stream);
                return launch_cfg<true, 64, 64, 64, 2, 4>(layout, g, Mmax, splits, stream);
            case 14:
                if (bbf) return launch_cfg<true, 64, 64, 128, 2, 4, false, true>(layout, g, Mmax, splits, stream);
                return launch_cfg<true, 64, 64, 128, 2, 4>(layout, g, Mmax, splits, stream);
            case 17:
                if (bbf) return launch_cfg<true, 64, 96, 128, 4, 2, false, true>(layout, g, Mmax, splits, stream);
                return launch_cfg<true, 64, 96, 128, 4, 2>(layout, g, Mmax, splits, stream);
            case 15: return launch_cfg<true, 64, 64, 64, 2, 2, true>(layout, g, Mmax, splits, stream);
            case 16: return launch_cfg<true, 64, 64, 32, 2, 2, true>(layout, g, Mmax, splits, stream);
            default: return IMMTSF_EINVAL;
        }
    }
    if (precision == 0) {
        const long tiles128 = (long)cdiv(Mmax, 128) * cdiv(g.N, 128) * nbz;
        if (tiles128 >= 512) return launch_cfg<false, 128, 128, 16, 2, 2>(layout, g, Mmax, splits, stream);
        return launch_cfg<false, 64, 64, 16, 2, 2>(layout, g, Mmax, splits, stream);
    }
    return IMMTSF_EINVAL;
}
