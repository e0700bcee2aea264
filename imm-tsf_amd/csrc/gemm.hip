// LDS-tiled MFMA GEMM for gfx950.  One kernel template serves the three layouts of a linear layer's
// forward (NT), data gradient (NN) and weight gradient (TN), in two arithmetic modes:
//   precision 0: exact fp32 on v_mfma_f32_16x16x4_f32   (parity mode: k-ordered fmaf chain, no rounding of inputs)
//   precision 1: bf16 operands, fp32 accumulate on v_mfma_f32_16x16x32_bf16 (operands are rounded RNE while
//                they are staged into LDS; HBM tensors stay fp32)
// Tiles are staged global -> registers -> LDS with the next tile's global loads issued before the current
// tile's MFMAs (issue-early / write-late), 64-lane wavefronts, each wave owning a (BM/WM)x(BN/WN) block of
// 16x16 accumulators.  Both LDS images are [row][k] with k contiguous, so the MFMA fragment of a lane is one
// ds_read_b128 (bf16) / ds_read_b32 (fp32); rows are padded by one access width against bank conflicts.
// Arbitrary M/N/K are supported by zero-filling the tile edges; M or K may live in device memory (ragged
// note count) so that no host synchronisation is needed to size the launch.
#include "gemm.hpp"

namespace {

template <bool BF16> struct LdsElem { typedef float T; };
template <> struct LdsElem<true> { typedef bf16_t T; };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <bool BF16, bool TA, bool TB, int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmArgs g) {
    typedef typename LdsElem<BF16>::T T;
    constexpr int NT = WM * WN * 64;
    constexpr int PAD = BF16 ? 8 : 1;
    constexpr int LDK = BK + PAD;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int CA = (BM * BK / 4) / NT, CB = (BN * BK / 4) / NT;
    static_assert(CA >= 1 && CB >= 1, "tile too small for the thread count");
    static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "chunking must be exact");

    __shared__ __attribute__((aligned(16))) T smem[(BM + BN) * LDK];
    T* As = smem;
    T* Bs = smem + BM * LDK;

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
    const int N = g.N + ((TA && TB && g.ones_col) ? 1 : 0);   // logical N incl. the virtual ones column
    const int Nreal = g.N;
    if (g.dyn) {
        const int dv = *g.dyn;
        if (g.dyn_which == 0) M = dv; else K = dv;
    }
    const int tiles_n = (N + BN - 1) / BN;
    const int row0 = (blockIdx.x / tiles_n) * BM, col0 = (blockIdx.x % tiles_n) * BN;
    if (row0 >= M) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const float* __restrict__ A = P.A + offA;
    const float* __restrict__ Bp = P.B + offB;

    float4 ra[CA], rb[CB];

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const int c = tid + i * NT;
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
            const int c = tid + i * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!TB) {
                const int r = c / (BK / 4), kq = c % (BK / 4);
                const int gcol = col0 + r, gk = k0 + kq * 4;
                if (gcol < N && gk < K) {
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
                if (gk < K && gcol < N) {
                    const int sk = g.b_rowmap ? g.b_rowmap[gk] : gk;
                    const float* src = Bp + (size_t)sk * g.ldb + gcol;
                    if (g.vecB && gcol + 3 < Nreal) v = *reinterpret_cast<const float4*>(src);
                    else {   // columns >= Nreal are the virtual ones column (only reachable when N == Nreal + 1)
                        v.x = (gcol < Nreal) ? src[0] : 1.f;
                        if (gcol + 1 < N) v.y = (gcol + 1 < Nreal) ? src[1] : 1.f;
                        if (gcol + 2 < N) v.z = (gcol + 2 < Nreal) ? src[2] : 1.f;
                        if (gcol + 3 < N) v.w = (gcol + 3 < Nreal) ? src[3] : 1.f;
                    }
                }
            }
            rb[i] = v;
        }
    };

    auto put4 = [&](T* base, bool transposed, int BR, int c, const float4& v) {
        if (!transposed) {
            const int r = c / (BK / 4), kq = c % (BK / 4);
            T* dst = base + r * LDK + kq * 4;
            if (BF16) {
                bf16x4 h;
                h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
                *reinterpret_cast<bf16x4*>(dst) = h;
            } else {
                dst[0] = (T)v.x; dst[1] = (T)v.y; dst[2] = (T)v.z; dst[3] = (T)v.w;
            }
        } else {
            const int kk = c / (BR / 4), rq = c % (BR / 4);
            T* dst = base + (rq * 4) * LDK + kk;
            dst[0] = (T)v.x; dst[LDK] = (T)v.y; dst[2 * LDK] = (T)v.z; dst[3 * LDK] = (T)v.w;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < CA; ++i) put4(As, TA, BM, tid + i * NT, ra[i]);
#pragma unroll
        for (int i = 0; i < CB; ++i) put4(Bs, TB, BN, tid + i * NT, rb[i]);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int nk = (K + BK - 1) / BK;
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) load_tile((t + 1) * BK);
        if (BF16) {
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(As) + (wm0 + i * 16 + fr) * LDK + kk * 32 + fq * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(Bs) + (wn0 + j * 16 + fr) * LDK + kk * 32 + fq * 8);
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
                for (int i = 0; i < TM; ++i) a[i] = reinterpret_cast<const float*>(As)[(wm0 + i * 16 + fr) * LDK + kk * 4 + fq];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = reinterpret_cast<const float*>(Bs)[(wn0 + j * 16 + fr) * LDK + kk * 4 + fq];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (t + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

    // epilogue: C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* __restrict__ C = P.C + offC;
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
                if (col >= Nreal) {   // virtual ones column: bias gradient
                    if (P.bias_grad) P.bias_grad[row] = g.alpha * acc[i][j][r];
                    continue;
                }
                float v = g.alpha * acc[i][j][r];
                if (P.bias) v += P.bias[col];
                if (!live) v = 0.f;
                if (g.add_vec) v += g.add_vec[col];
                if (g.act == 1) v = fmaxf(v, 0.f);
                else if (g.act == 2) v = gelu_erf(v);
                float* dst = C + (size_t)row * g.ldc + col;
                if (g.accumulate) v += *dst;
                *dst = v;
            }
        }
    }
}

template <bool BF16, int BM, int BN, int BK, int WM, int WN>
int launch_cfg(int layout, const GemmArgs& g, int Mmax, hipStream_t stream) {
    const int Nlog = g.N + ((layout == GEMM_TN && g.ones_col) ? 1 : 0);
    dim3 grid(cdiv(Mmax, BM) * cdiv(Nlog, BN), 1, g.nprob * (g.nbatch > 1 ? g.nbatch : 1)), block(WM * WN * 64);
    if (grid.x == 0) return IMMTSF_OK;
    switch (layout) {
        case GEMM_NT: hipLaunchKernelGGL((gemm_kernel<BF16, false, false, BM, BN, BK, WM, WN>), grid, block, 0, stream, g); break;
        case GEMM_NN: hipLaunchKernelGGL((gemm_kernel<BF16, false, true, BM, BN, BK, WM, WN>), grid, block, 0, stream, g); break;
        case GEMM_TN: hipLaunchKernelGGL((gemm_kernel<BF16, true, true, BM, BN, BK, WM, WN>), grid, block, 0, stream, g); break;
        default: return IMMTSF_EINVAL;
    }
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

}  // namespace

// ---- optional per-launch timing tap (bench.py's roofline leg): hipEvents bracket every GEMM launch on the
// stream it is launched on.  Off by default; the only process-global state in the library.
namespace {
struct TapRec { hipEvent_t e0, e1; int meta[8]; };
constexpr int kTapCap = 16384;
TapRec* g_tap = nullptr;
int g_tap_n = 0, g_tap_on = 0, g_tap_events = 0;
}  // namespace

static int launch_gemm_impl(int layout, int precision, GemmArgs& g, hipStream_t stream);

int immtsf_launch_gemm(int layout, int precision, GemmArgs& g, hipStream_t stream) {
    if (!g_tap_on || g_tap_n >= kTapCap) return launch_gemm_impl(layout, precision, g, stream);
    TapRec& r = g_tap[g_tap_n];
    if (g_tap_n >= g_tap_events) {
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return launch_gemm_impl(layout, precision, g, stream);
        g_tap_events = g_tap_n + 1;
    }
    const int m[8] = {layout, precision, g.M, g.N, g.K, g.nprob, g.nbatch > 1 ? g.nbatch : 1, g.dyn ? 1 + g.dyn_which : 0};
    for (int i = 0; i < 8; ++i) r.meta[i] = m[i];
    (void)hipEventRecord(r.e0, stream);
    const int rc = launch_gemm_impl(layout, precision, g, stream);
    (void)hipEventRecord(r.e1, stream);
    ++g_tap_n;
    return rc;
}

extern "C" int immtsf_timing_enable(int on) {
    if (on && !g_tap) g_tap = new TapRec[kTapCap];
    g_tap_on = on ? 1 : 0;
    g_tap_n = 0;
    return 0;
}

// host arrays: meta[8*max] (layout, precision, M, N, K, nprob, nbatch, dyn), ms[max]; returns the record count
extern "C" int immtsf_timing_collect(int max, int* meta, float* ms) {
    const int n = g_tap_n < max ? g_tap_n : max;
    for (int i = 0; i < n; ++i) {
        (void)hipEventSynchronize(g_tap[i].e1);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_tap[i].e0, g_tap[i].e1);
        ms[i] = t;
        for (int k = 0; k < 8; ++k) meta[8 * i + k] = g_tap[i].meta[k];
    }
    g_tap_n = 0;
    return n;
}

static int launch_gemm_impl(int layout, int precision, GemmArgs& g, hipStream_t stream) {
    if (g.nprob < 1 || g.nprob > IMMTSF_GEMM_MAX_PROBLEMS) return IMMTSF_EINVAL;
    if (g.M < 0 || g.N < 0 || g.K < 0) return IMMTSF_EINVAL;
    if (g.M == 0 || g.N == 0) return IMMTSF_OK;
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
    if (g.row_flag && g.row_flag_div <= 0) return IMMTSF_EINVAL;
    if (g.nbatch > 1 && g.batch_inner <= 0) return IMMTSF_EINVAL;
    if (g.ones_col && layout != GEMM_TN) return IMMTSF_EINVAL;
    const int Mmax = g.M;   // g.M is the allocation-time upper bound when `dyn` overrides M
    const long tiles128 = (long)cdiv(Mmax, 128) * cdiv(g.N, 128) * g.nprob;
    if (precision == 1) {
        if (tiles128 >= 512) return launch_cfg<true, 128, 128, 64, 2, 2>(layout, g, Mmax, stream);
        return launch_cfg<true, 64, 64, 64, 2, 2>(layout, g, Mmax, stream);
    }
    if (precision == 0) {
        if (tiles128 >= 512) return launch_cfg<false, 128, 128, 16, 2, 2>(layout, g, Mmax, stream);
        return launch_cfg<false, 64, 64, 16, 2, 2>(layout, g, Mmax, stream);
    }
    return IMMTSF_EINVAL;
}
