// MMF_XAttn_Add (fusions/MMF_XAttn_Add.py:56-103) in its LOW-RANK form (round 3).
//
// The block's queries come from Y_ts, which has only C columns (the series variables, C = 8 at the benchmark configuration), and
// its attention output is consumed only by residual_head, which has only C rows.  Between proj_{q,k,v} and the MHA in-projections
// and between out_proj and residual_head there is no nonlinearity, so per head h (head dimension E, m = the head's rows)
//
//     S_h[t,s]  = scale q_h[t] . k_h[s] = [Y[t] | 1] G_h [E_txt[s] | 1]^T         G_h = scale [W_Qf,h | b_q,h]^T [W_Kf,h | b_k,h]   (C+1) x (d+1)
//     delta[t]  = sum_h sum_s A_h[t,s] ( U_h [E_txt[s] | 1]^T ) + b_HO               U_h = W_HO,h [W_Vf,h | b_v,h]                    C x (d+1)
//
// with W_Qf = W_in,q W_q, W_Kf = W_in,k W_k, W_Vf = W_in,v W_v, W_HO = W_res W_out, b_HO = W_res b_out + b_res.  G_h and U_h depend on
// parameters only.  The text side therefore needs ONE skinny projection per step, P = [E_txt | 1] W_fold^T with W_fold = the stacked
// rows of (G_h, U_h): (2C+1) H columns instead of the 2d columns of (k | v), and the whole T x T attention of a window (scores,
// softmax, dropout, mix, bias, LayerNorm(C), dropout, kappa blend) is a few hundred FMAs per row on those columns: one kernel per
// direction, fp32 in both precision modes.  What disappears: the (B T) x 2d x d key/value projection with its data and weight
// gradients (the largest GEMMs of the block: 0.93 TFLOP per step at 4096 windows), the MFMA tile attention over d-wide rows and
// its 400 MB context tensors, the C-wide head over them.  The gradients of the ORIGINAL parameters follow from dW_fold = dP^T [E|1]
// by the chain rule through the factors above -- products with at most (C+1) H rows or rank, three small multi-job launches.
// Same function, same state_dict; results equal to the reference's up to fp32 reassociation (bf16 mode: E_txt and W_fold are rounded
// once for the P product; everything else is fp32).
//
// Split for two-stream scheduling like the full-rank form: the P half (fold + projection; backward: dE_txt and every parameter
// gradient except LayerNorm's) depends only on the text side; the Q half (attention + head) is the serial section between the
// backbone's forward and backward.  b_HO travels from the P half to the Q half as a C-vector and its gradient back.
#include "../../include/immtsf.h"
#include "gemm.hpp"
#include "rowops.hpp"
#include "tail.hpp"
#include "block_util.hpp"
#include "t2v_fold.hpp"
#include <math.h>
#include <algorithm>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------------------------------------
// small multi-job launches: every product of the fold / chain rule has <= 16 rows or rank <= 16 per group
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int XR = 8;             // rows of an NN / NT job (the host splits taller products); OUTER jobs walk their rank in chunks of XR
constexpr int XJ_MAX = 20;
enum { XJ_NN = 0, XJ_NT = 1, XJ_OUTER = 2, XJ_COPY = 3 };
struct XJob {
    int form, nblk;
    const float* L; int lsr, lsn;     // NN / NT: L[r * lsr + k * lsn]   OUTER: P[j * lsr + m * lsn]   COPY: L[i * lsn] (null: zeros)
    int R;                            // NN / NT: rows   OUTER: rank per group
    const float* W; int ldw, qsn;     // NN: W[k * ldw + n]   NT: W[m * ldw + k]   OUTER: Q[j * ldw + n * qsn]
    int K, N, M;                      // NN: K, N   NT: K, N (= output columns m)   OUTER: M x N output   COPY: N elements
    float* out; int osr, osn;         // out[r * osr + n * osn]  (OUTER: out[m * osr + n * osn])
    unsigned short* out16;            // NN / COPY: optional bf16 copy, same indexing
    float alpha;
    const float *lb, *wb;             // NT: alpha * (dot + lb[r] * wb[m])
    const float* radd;                // NT: + radd[r]
    const float* xrow;                // NT: out[R * osr + m * osn] = xrow[m]  (one more row, copied)
    int ng; long pgs, qgs;            // OUTER: sum over ng groups, P += pgs, Q += qgs per group
};
struct XJobs { int first[XJ_MAX]; int n; int lds_floats; XJob j[XJ_MAX]; };       // first[i]: first workgroup of job i

__device__ __forceinline__ unsigned short f2bf(float v) {
    return __builtin_bit_cast(unsigned short, static_cast<__bf16>(v));
}

constexpr int XJ_LK = 16 * 1024;      // most floats of a staged L operand (dynamic LDS, sized per launch; 0: every job reads L from global memory)

// the <= 16 rows of an NN / NT job's L operand as an LDS image Ls[r * K + k] (one round of coalesced loads; the products then read it
// as broadcasts / 16-byte rows instead of issuing R dependent global loads per k step)
__device__ __forceinline__ bool xjob_stage_l(const XJob& J, float* Ls, int cap) {
    if (J.R * J.K > cap) return false;
    if (J.lsn == 1) {
        for (int i = threadIdx.x; i < J.R * J.K; i += 256) {
            const int r = i / J.K, k = i - r * J.K;
            Ls[i] = J.L[(size_t)r * J.lsr + k];
        }
    } else {       // k-major source (W_q read as its transpose): consecutive threads take consecutive source elements
        for (int i = threadIdx.x; i < J.R * J.K; i += 256) {
            const int k = i / J.R, r = i - k * J.R;
            Ls[r * J.K + k] = J.L[(size_t)r * J.lsr + (size_t)k * J.lsn];
        }
    }
    __syncthreads();
    return true;
}

template <bool STAGED>
__device__ __forceinline__ void xjob_nn_acc(const XJob& J, int kg, int n, const float* Ls, float4 (&acc)[XR]) {
    constexpr int U = STAGED ? 8 : 6;
    const int R = J.R;
    for (int k0 = kg; k0 < J.K; k0 += 64 * U) {
        float4 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            w[u] = k0 + 64 * u < J.K ? *reinterpret_cast<const float4*>(J.W + (size_t)(k0 + 64 * u) * J.ldw + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (STAGED) {
#pragma unroll
            for (int r = 0; r < XR; ++r)
                if (r < R) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int k = k0 + 64 * u;
                        const float l = k < J.K ? Ls[r * J.K + k] : 0.f;
                        acc[r].x = fmaf(l, w[u].x, acc[r].x); acc[r].y = fmaf(l, w[u].y, acc[r].y);
                        acc[r].z = fmaf(l, w[u].z, acc[r].z); acc[r].w = fmaf(l, w[u].w, acc[r].w);
                    }
                }
        } else {        // every L value of the round requested before the first use
#pragma unroll
            for (int rb = 0; rb < XR; rb += 8) {        // (eight rows at a time: the operand registers of sixteen would halve the occupancy)
                if (rb >= R) break;
                float l[8][U];
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int k = k0 + 64 * u;
                        l[r][u] = (rb + r < R && k < J.K) ? J.L[(size_t)(rb + r) * J.lsr + (size_t)k * J.lsn] : 0.f;
                    }
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        acc[rb + r].x = fmaf(l[r][u], w[u].x, acc[rb + r].x); acc[rb + r].y = fmaf(l[r][u], w[u].y, acc[rb + r].y);
                        acc[rb + r].z = fmaf(l[r][u], w[u].z, acc[rb + r].z); acc[rb + r].w = fmaf(l[r][u], w[u].w, acc[rb + r].w);
                    }
            }
        }
    }
}

// NN: out[r][n] = alpha * sum_k L[r][k] W[k][n].  A workgroup owns 16 output columns: thread = (4-column group eq, k-group kg of 64);
// up to 12 k steps of W are in flight per thread
__device__ __forceinline__ void xjob_nn(const XJob& J, int blk, float* Ls, int cap, float* red /* [4][XR][16] */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, eq = tid & 3, kg = tid >> 2, n = blk * 16 + eq * 4, R = J.R;
    const bool staged = xjob_stage_l(J, Ls, cap);
    float4 acc[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < J.N) {
        if (staged) xjob_nn_acc<true>(J, kg, n, Ls, acc);
        else xjob_nn_acc<false>(J, kg, n, Ls, acc);
    }
    // the sixteen k-groups of a wave sit 4 lanes apart: lane ^ 4, ^ 8 (row rotates), ^ 16, ^ 32
#define XR_RED(v) v += IMMTSF_DPP(v, 0x124); v += IMMTSF_DPP(v, 0x128); v = xor32_sum(xor16_sum(v))
#pragma unroll
    for (int r = 0; r < XR; ++r)
        if (r < R) {
            XR_RED(acc[r].x); XR_RED(acc[r].y); XR_RED(acc[r].z); XR_RED(acc[r].w);
            if (lane < 4) *reinterpret_cast<float4*>(&red[(wave * XR + r) * 16 + eq * 4]) = acc[r];
        }
#undef XR_RED
    __syncthreads();
    for (int x = tid; x < R * 16; x += 256) {
        const int r = x >> 4, col = x & 15, nn = blk * 16 + col;
        if (nn < J.N) {
            const float v = J.alpha * ((red[(0 * XR + r) * 16 + col] + red[(1 * XR + r) * 16 + col]) +
                                       (red[(2 * XR + r) * 16 + col] + red[(3 * XR + r) * 16 + col]));
            const size_t o = (size_t)r * J.osr + (size_t)nn * J.osn;
            J.out[o] = v;
            if (J.out16) J.out16[o] = f2bf(v);
        }
    }
}

// NT: out[r][m] = alpha * (sum_k L[r][k] W[m][k] + lb[r] wb[m]) + radd[r].  A wave owns two output columns m, the lanes stride k.
// Row-major L is read straight from global memory (16-byte loads, a round's worth in flight); a k-major L is staged in LDS.
__device__ __forceinline__ void xjob_nt(const XJob& J, int blk, float* Ls, int cap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, R = J.R;
    const bool staged = J.lsn != 1 && xjob_stage_l(J, Ls, cap);
    const int m0 = blk * 8 + wave * 2;
    if (m0 >= J.N) return;
    const bool two = m0 + 1 < J.N;
    const float* w0 = J.W + (size_t)m0 * J.ldw;
    const float* w1 = J.W + (size_t)(two ? m0 + 1 : m0) * J.ldw;
    float a0[XR], a1[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) a0[r] = a1[r] = 0.f;
    const bool vec = (J.K & 3) == 0 && (J.ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(J.W) & 15) == 0;
    if (vec && staged) {
        constexpr int U = 4;                 // 16-byte steps per lane in flight (K <= 1024 per round)
        for (int k0 = lane * 4; k0 < J.K; k0 += 256 * U) {
            float4 x0[U], x1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 256 * u;
                x0[u] = k < J.K ? *reinterpret_cast<const float4*>(w0 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
                x1[u] = k < J.K ? *reinterpret_cast<const float4*>(w1 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int r = 0; r < XR; ++r)
                if (r < R) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int k = k0 + 256 * u;
                        if (k < J.K) {
                            const float4 l = *reinterpret_cast<const float4*>(Ls + r * J.K + k);
                            a0[r] = fmaf(l.x, x0[u].x, fmaf(l.y, x0[u].y, fmaf(l.z, x0[u].z, fmaf(l.w, x0[u].w, a0[r]))));
                            a1[r] = fmaf(l.x, x1[u].x, fmaf(l.y, x1[u].y, fmaf(l.z, x1[u].z, fmaf(l.w, x1[u].w, a1[r]))));
                        }
                    }
                }
        }
    } else if (vec && J.lsn == 1 && (J.lsr & 3) == 0 && (reinterpret_cast<uintptr_t>(J.L) & 15) == 0) {
        constexpr int U = 3;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = lane * 4; k0 < J.K; k0 += 256 * U) {
            float4 x0[U], x1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 256 * u;
                x0[u] = k < J.K ? *reinterpret_cast<const float4*>(w0 + k) : z4;
                x1[u] = k < J.K ? *reinterpret_cast<const float4*>(w1 + k) : z4;
            }
#pragma unroll
            for (int rb = 0; rb < XR; rb += 8) {
                if (rb >= R) break;
                float4 l[8][U];
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int k = k0 + 256 * u;
                        l[r][u] = (rb + r < R && k < J.K) ? *reinterpret_cast<const float4*>(J.L + (size_t)(rb + r) * J.lsr + k) : z4;
                    }
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        a0[rb + r] = fmaf(l[r][u].x, x0[u].x, fmaf(l[r][u].y, x0[u].y, fmaf(l[r][u].z, x0[u].z, fmaf(l[r][u].w, x0[u].w, a0[rb + r]))));
                        a1[rb + r] = fmaf(l[r][u].x, x1[u].x, fmaf(l[r][u].y, x1[u].y, fmaf(l[r][u].z, x1[u].z, fmaf(l[r][u].w, x1[u].w, a1[rb + r]))));
                    }
            }
        }
    } else if (staged) {
#pragma unroll 4
        for (int k = lane; k < J.K; k += 64) {
            const float x0 = w0[k], x1 = w1[k];
#pragma unroll
            for (int r = 0; r < XR; ++r)
                if (r < R) {
                    const float l = Ls[r * J.K + k];
                    a0[r] = fmaf(l, x0, a0[r]);
                    a1[r] = fmaf(l, x1, a1[r]);
                }
        }
    } else {
#pragma unroll 4
        for (int k = lane; k < J.K; k += 64) {
            const float x0 = w0[k], x1 = w1[k];
#pragma unroll
            for (int r = 0; r < XR; ++r)
                if (r < R) {
                    const float l = J.L[(size_t)r * J.lsr + (size_t)k * J.lsn];
                    a0[r] = fmaf(l, x0, a0[r]);
                    a1[r] = fmaf(l, x1, a1[r]);
                }
        }
    }
    float mine0 = 0.f, mine1 = 0.f;
#pragma unroll
    for (int r = 0; r < XR; ++r)
        if (r < R) {
            const float v0 = wave_sum(a0[r]), v1 = wave_sum(a1[r]);
            if (lane == r) { mine0 = v0; mine1 = v1; }
        }
    if (lane < R) {
        const float lbv = J.lb ? J.lb[lane] : 0.f, ra = J.radd ? J.radd[lane] : 0.f;
        float v = mine0;
        if (J.lb) v = fmaf(lbv, J.wb[m0], v);
        J.out[(size_t)lane * J.osr + (size_t)m0 * J.osn] = fmaf(v, J.alpha, ra);
        if (two) {
            v = mine1;
            if (J.lb) v = fmaf(lbv, J.wb[m0 + 1], v);
            J.out[(size_t)lane * J.osr + (size_t)(m0 + 1) * J.osn] = fmaf(v, J.alpha, ra);
        }
    }
    if (J.xrow && lane == 63) {
        J.out[(size_t)R * J.osr + (size_t)m0 * J.osn] = J.xrow[m0];
        if (two) J.out[(size_t)R * J.osr + (size_t)(m0 + 1) * J.osn] = J.xrow[m0 + 1];
    }
}

// OUTER: out[m][n] = alpha * sum_g sum_{j < R} P_g[j][m] Q_g[j][n].  A workgroup owns 16 rows x 256 columns, a thread 4 x 4 of them;
// a group's R rows of both operands are requested together
__device__ __forceinline__ void xjob_outer(const XJob& J, int blk) {
    const int nbn = (J.N + 255) / 256, bm = blk / nbn, bn = blk - bm * nbn;
    const int m0 = bm * 16 + (threadIdx.x >> 6) * 4, n0 = bn * 256 + (threadIdx.x & 63) * 4;
    if (m0 >= J.M || n0 >= J.N) return;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    const bool vq = J.qsn == 1 && n0 + 3 < J.N && (J.ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(J.W) & 15) == 0 && (J.qgs & 3) == 0;
    const bool vp = J.lsn == 1 && m0 + 3 < J.M && (J.lsr & 3) == 0 && (reinterpret_cast<uintptr_t>(J.L) & 15) == 0 && (J.pgs & 3) == 0;
    for (int g = 0; g < J.ng; ++g) {
        const float* P = J.L + g * J.pgs;
        const float* Q = J.W + g * J.qgs;
        for (int j0 = 0; j0 < J.R; j0 += XR) {
            float4 pv[XR], qv[XR];
#pragma unroll
            for (int jj = 0; jj < XR; ++jj) {
                const int j = j0 + jj;
                pv[jj] = qv[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < J.R) {
                    if (vp) pv[jj] = *reinterpret_cast<const float4*>(P + (size_t)j * J.lsr + m0);
                    else {
                        const float* pp = P + (size_t)j * J.lsr;
                        pv[jj].x = pp[(size_t)m0 * J.lsn];
                        if (m0 + 1 < J.M) pv[jj].y = pp[(size_t)(m0 + 1) * J.lsn];
                        if (m0 + 2 < J.M) pv[jj].z = pp[(size_t)(m0 + 2) * J.lsn];
                        if (m0 + 3 < J.M) pv[jj].w = pp[(size_t)(m0 + 3) * J.lsn];
                    }
                    if (vq) qv[jj] = *reinterpret_cast<const float4*>(Q + (size_t)j * J.ldw + n0);
                    else {
                        const float* qq = Q + (size_t)j * J.ldw;
                        qv[jj].x = qq[(size_t)n0 * J.qsn];
                        if (n0 + 1 < J.N) qv[jj].y = qq[(size_t)(n0 + 1) * J.qsn];
                        if (n0 + 2 < J.N) qv[jj].z = qq[(size_t)(n0 + 2) * J.qsn];
                        if (n0 + 3 < J.N) qv[jj].w = qq[(size_t)(n0 + 3) * J.qsn];
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < XR; ++jj) {
                const float p[4] = {pv[jj].x, pv[jj].y, pv[jj].z, pv[jj].w}, q[4] = {qv[jj].x, qv[jj].y, qv[jj].z, qv[jj].w};
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(p[a], q[b], acc[a][b]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if (m0 + a >= J.M) break;
        float* o = J.out + (size_t)(m0 + a) * J.osr + (size_t)n0 * J.osn;
        if (J.osn == 1 && n0 + 3 < J.N && (J.osr & 3) == 0 && (reinterpret_cast<uintptr_t>(J.out) & 15) == 0) {
            *reinterpret_cast<float4*>(o) = make_float4(J.alpha * acc[a][0], J.alpha * acc[a][1], J.alpha * acc[a][2], J.alpha * acc[a][3]);
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (n0 + b < J.N) o[(size_t)b * J.osn] = J.alpha * acc[a][b];
        }
    }
}

typedef const XJobs* XJobsK;
__global__ __launch_bounds__(256, 2) void xjobs_kernel(XJobs js_arg) {
    extern __shared__ __attribute__((aligned(16))) float Ls[];      // lds_floats floats
    __shared__ __attribute__((aligned(16))) float red[4 * XR * 16];
    // the job table is read where it lies, in the kernel-argument segment (scalar loads at a run-time offset): indexing the by-value
    // parameter with a run-time index makes hipcc copy all 3 KB of it into registers / scratch
    XJobsK js = (XJobsK)(const void*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)js_arg;
    const int n = js->n;
    int ji = 0;
#pragma unroll
    for (int i = 1; i < XJ_MAX; ++i) ji += (i < n && (int)blockIdx.x >= js->first[i]) ? 1 : 0;
    const int blk = blockIdx.x - js->first[ji];
    const XJob J = js->j[ji];
    const int cap = js->lds_floats;
    if (J.form == XJ_NN) xjob_nn(J, blk, Ls, cap, red);
    else if (J.form == XJ_NT) xjob_nt(J, blk, Ls, cap);
    else if (J.form == XJ_OUTER) xjob_outer(J, blk);
    else {
        for (int i = blk * 1024 + threadIdx.x; i < min(J.N, (blk + 1) * 1024); i += 256) {
            const float v = J.L ? J.alpha * J.L[(size_t)i * J.lsn] : 0.f;
            if (J.out) J.out[(size_t)i * J.osn] = v;
            if (J.out16) J.out16[(size_t)i * J.osn] = f2bf(v);
        }
    }
}

struct XJobList {
    std::vector<XJob> jobs;
    XJob* add(int form, int nblk) {
        jobs.emplace_back();
        XJob* j = &jobs.back();
        memset(j, 0, sizeof(XJob));
        j->form = form; j->nblk = nblk; j->alpha = 1.f; j->lsn = 1; j->osn = 1; j->qsn = 1; j->ng = 1;
        return j;
    }
    // (products taller than XR rows are split into row chunks: a job's accumulators are sized for XR)
    void nn(const float* L, int lsr, int R, const float* W, int ldw, int K, int N, float* out, int osr, float alpha, unsigned short* out16 = nullptr,
            int osn = 1) {
        for (int r0 = 0; r0 < R; r0 += XR) {
            XJob* j = add(XJ_NN, cdiv(N, 16));
            j->L = L + (size_t)r0 * lsr; j->lsr = lsr; j->R = std::min(XR, R - r0); j->W = W; j->ldw = ldw; j->K = K; j->N = N;
            j->out = out + (size_t)r0 * osr; j->osr = osr; j->osn = osn; j->alpha = alpha;
            j->out16 = out16 ? out16 + (size_t)r0 * osr : nullptr;
        }
    }
    // lb / radd: per-row vectors (advance with the row chunk), wb: per-column vector, xrow: copied into row R (by the last chunk)
    void nt(const float* L, int lsr, int lsn, int R, const float* W, int ldw, int K, int Mout, float* out, int osr, float alpha,
            const float* lb = nullptr, const float* wb = nullptr, const float* radd = nullptr, const float* xrow = nullptr) {
        for (int r0 = 0; r0 < R; r0 += XR) {
            XJob* j = add(XJ_NT, cdiv(Mout, 8));
            const int rc = std::min(XR, R - r0);
            j->L = L + (size_t)r0 * lsr; j->lsr = lsr; j->lsn = lsn; j->R = rc; j->W = W; j->ldw = ldw; j->K = K; j->N = Mout;
            j->out = out + (size_t)r0 * osr; j->osr = osr; j->alpha = alpha;
            j->lb = lb ? lb + r0 : nullptr; j->wb = wb; j->radd = radd ? radd + r0 : nullptr;
            j->xrow = (xrow && r0 + rc == R) ? xrow : nullptr;
        }
    }
    XJob* outer(const float* P, int lsr, int rank, const float* Q, int ldq, int M, int N, float* out, int osr, float alpha) {
        XJob* j = add(XJ_OUTER, cdiv(M, 16) * cdiv(N, 256));
        j->L = P; j->lsr = lsr; j->R = rank; j->W = Q; j->ldw = ldq; j->M = M; j->N = N; j->out = out; j->osr = osr; j->alpha = alpha;
        return j;
    }
    void copy(const float* src, int n, float* out, unsigned short* out16 = nullptr) {
        XJob* j = add(XJ_COPY, cdiv(n, 1024));
        j->L = src; j->N = n; j->out = out; j->out16 = out16;
    }
    // stage: the L operands of the NN jobs (and of NT jobs with a k-major L) go through LDS -- for launches that run beside nothing
    // LDS-hungry; without it every job reads L from global memory (no LDS beyond 4 KB: the backward's launches share the chip with the
    // backbone's backward, whose workgroups would otherwise wait for LDS behind several hundred 30 KB workgroups)
    int launch(hipStream_t s, bool stage) {
        for (size_t b0 = 0; b0 < jobs.size(); b0 += XJ_MAX) {
            XJobs js;
            memset(&js, 0, sizeof(js));
            int blocks = 0, fl = 0;
            const int n = (int)std::min<size_t>(XJ_MAX, jobs.size() - b0);
            for (int i = 0; i < n; ++i) {
                const XJob& j = jobs[b0 + i];
                js.j[i] = j;
                js.first[i] = blocks;
                blocks += j.nblk;
                const bool wants = j.form == XJ_NN || (j.form == XJ_NT && j.lsn != 1);
                if (stage && wants && j.R * j.K <= XJ_LK && j.R * j.K > fl) fl = j.R * j.K;
            }
            js.n = n;
            js.lds_floats = fl;
            if (!blocks) continue;
            hipLaunchKernelGGL(xjobs_kernel, dim3(blocks), dim3(256), (size_t)fl * sizeof(float), s, js);
            IMMTSF_LAUNCH_CHECK();
        }
        return IMMTSF_OK;
    }
};

// ------------------------------------------------------------------------------------------------------------------------------
// shapes
// ------------------------------------------------------------------------------------------------------------------------------
struct XRDims { int B, T, C, Cq, H, d, E, Wd, PW; };       // Wd = 2C+1 columns per head, PW = row pitch of P (multiple of 8)
inline XRDims xr_dims(const immtsf_fusion_cfg* c) {
    XRDims x;
    x.B = c->B; x.T = c->T; x.C = c->C; x.Cq = c->C + 1; x.H = c->H; x.d = c->d; x.E = c->d / c->H; x.Wd = 2 * c->C + 1;
    x.PW = (x.H * x.Wd + 7) & ~7;
    return x;
}
inline int xq_mask_bytes(int T) { return (((T + 3) / 4) + 3) & ~3; }       // dropout keep bits of a (row, head): 4 keys per byte
// LDS floats of the Q kernels per window: P rows, Y rows, ddelta rows, lse, D, dropout keep bits
#define XQ_PL(CM) (2 * (CM) + 4)       // pitch of a key row padded to the template width CM (xrank_q_train_kernel): k weights | v weights | k bias
inline size_t xq_lds_floats(const XRDims& x) {
    const size_t masks = (size_t)x.H * x.T * (xq_mask_bytes(x.T) / 4);
    const size_t packed = (size_t)x.T * (x.PW + 2 * x.C + 2 * x.H) + masks;          // forward / backward kernels: rows as P has them
    const int CM = x.C <= 8 ? 8 : 16, TP = (x.T + 3) & ~3;
    const size_t padded = (size_t)x.H * TP * XQ_PL(CM) + (size_t)x.T * (2 * CM + 2 * x.H) + masks;      // training kernel
    return packed > padded ? packed : padded;
}
constexpr size_t XQ_LDS_MAX = 60 * 1024;
bool xr_supported(const immtsf_fusion_cfg* c) {
    if (bad_cfg(c) || c->C < 1 || c->C > 15 || c->H > 4 || ((c->d / c->H) & 3) || c->d > 4096) return false;
    const XRDims x = xr_dims(c);
    if (x.PW > 128) return false;
    return xq_lds_floats(x) * sizeof(float) <= XQ_LDS_MAX;
}
inline bool xr_hf(const immtsf_fusion_cfg* c) { return c->precision == 1 && c->d >= 16 && (c->d % 16) == 0; }

// forward workspace of the P half: the factors of the fold (kept for the backward)
struct XPWs {
    float *AqT, *WHO, *GA, *UA, *Wf, *Wfb;
    unsigned short *Wf16, *E16;
    float *Wc, *bc;              // the "_z" form: W_fold composed with the producer's last linear map, Wc = W_fold W_po, bc = W_fold b_po + b_fold
    unsigned short *Wc16, *Wpo16;      // Wpo16: bf16 image of the producer's weight when no twin is registered
    size_t bytes;
};
XPWs carve_xp(const immtsf_fusion_cfg* c, void* base) {
    const XRDims x = xr_dims(c);
    const size_t d = x.d;
    const bool hf = xr_hf(c);
    Carver k(base);
    XPWs w;
    w.AqT = k.take<float>(x.Cq * d);
    w.WHO = k.take<float>(x.C * d);
    w.GA = k.take<float>((size_t)x.H * x.Cq * d);
    w.UA = k.take<float>((size_t)x.H * x.C * d);
    w.Wf = k.take<float>((size_t)x.PW * d);
    w.Wfb = k.take<float>(x.PW);
    w.Wf16 = hf ? k.take<unsigned short>((size_t)x.PW * d) : nullptr;
    w.E16 = hf ? k.take<unsigned short>((size_t)x.B * x.T * d) : nullptr;
    w.Wc = k.take<float>((size_t)x.PW * d);
    w.bc = k.take<float>(x.PW);
    w.Wc16 = hf ? k.take<unsigned short>((size_t)x.PW * d) : nullptr;
    w.Wpo16 = hf ? k.take<unsigned short>(d * d) : nullptr;
    w.bytes = k.bytes();
    return w;
}
struct XPScratch {
    float *dWf, *dWfb, *T1, *RW, *dAqT, *dWHO;
    float *dWc, *dbc;            // the "_z" form
    unsigned short *dP16, *dWc16;
    void* sk;
    size_t skb, bytes;
};
XPScratch carve_xp_scratch(const immtsf_fusion_cfg* c, void* base) {
    const XRDims x = xr_dims(c);
    const size_t d = x.d, BT = (size_t)x.B * x.T;
    const bool hf = xr_hf(c);
    Carver k(base);
    XPScratch s;
    s.dWf = k.take<float>((size_t)x.PW * d);
    s.dWfb = k.take<float>(x.PW);
    s.T1 = k.take<float>((size_t)x.H * x.Cq * d);
    s.RW = k.take<float>((size_t)x.H * x.C * d);
    s.dAqT = k.take<float>(x.Cq * d);
    s.dWHO = k.take<float>(x.C * d);
    s.dP16 = hf ? k.take<unsigned short>(BT * x.PW) : nullptr;
    s.dWc = k.take<float>((size_t)x.PW * d);
    s.dbc = k.take<float>(x.PW);
    s.dWc16 = hf ? k.take<unsigned short>((size_t)x.PW * d) : nullptr;
    s.skb = hf ? immtsf_gemm3_tn_ws_bytes(x.PW, (int)d, (int)BT) : 0;
    s.sk = s.skb ? k.take<unsigned char>(s.skb) : nullptr;
    s.bytes = k.bytes();
    return s;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Q half: a group of `parts` adjacent lanes per (window, row) -- query row in the first phase, key row in the second (backward);
// the lanes of a group split the other index and meet by DPP sums
// ------------------------------------------------------------------------------------------------------------------------------
struct XQDims { int B, T, C, H, PW, Wd, wpb, TB, parts, psh; float kappa; };
struct XQWs {
    float *lse, *xhat, *rstd, *slabs;
    unsigned int* ticket;
    size_t bytes;
};
inline int xq_wpb(const XRDims& x) {
    int wpb = 256 / x.T;
    if (wpb < 1) wpb = 1;
    const size_t per = xq_lds_floats(x) * sizeof(float);
    while (wpb > 1 && per * wpb > XQ_LDS_MAX) --wpb;
    // enough workgroups for the chip before windows share one
    while (wpb > 1 && cdiv(x.B, wpb) < 256) --wpb;
    return wpb;
}
XQWs carve_xq(const immtsf_fusion_cfg* c, void* base) {
    const XRDims x = xr_dims(c);
    const size_t BT = (size_t)x.B * x.T;
    Carver k(base);
    XQWs w;
    w.lse = k.take<float>(BT * x.H);
    w.xhat = k.take<float>(BT * x.C);
    w.rstd = k.take<float>(BT);
    w.slabs = k.take<float>((size_t)cdiv(x.B, xq_wpb(x)) * 3 * x.C);
    w.ticket = k.take<unsigned int>(4);
    w.bytes = k.bytes();
    return w;
}

struct DropL { uint64_t seed; float p, inv_keep; };
__device__ __forceinline__ DropL drop_local(const DropCfg& d) {        // the device-side seed counter read ONCE (a dependent load per call otherwise)
    DropL l;
    l.seed = d.seed + (d.seed_dev ? *d.seed_dev : 0ull);
    l.p = d.p; l.inv_keep = d.inv_keep;
    return l;
}
// elements idx .. idx + 3 of a site (only the first n matter): one Philox call when idx is a multiple of 4
__device__ __forceinline__ void drop4(const DropL& d, uint64_t site, uint64_t idx, int n, float (&s)[4]) {
    if (d.p <= 0.f) { s[0] = s[1] = s[2] = s[3] = 1.f; return; }
    if ((idx & 3) == 0) {
        const Philox4 r = philox4x32_10(d.seed, site, idx >> 2);
        const uint32_t bits[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = (float)(bits[e] >> 8) * (1.0f / 16777216.0f) >= d.p ? d.inv_keep : 0.f;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = e < n ? dropout_scale(d.seed, site, idx + e, d.p, d.inv_keep) : 0.f;
    }
}
__device__ __forceinline__ float group_max(float v, int CT) {       // CT <= 16
    if (CT >= 2) v = fmaxf(v, IMMTSF_DPP(v, 0xB1));
    if (CT >= 4) v = fmaxf(v, IMMTSF_DPP(v, 0x4E));
    if (CT >= 8) v = fmaxf(v, IMMTSF_DPP(v, 0x141));
    if (CT >= 16) v = fmaxf(v, IMMTSF_DPP(v, 0x140));
    return v;
}
template <int CM>
__device__ __forceinline__ void load_row(const float* __restrict__ p, int C, float (&v)[CM]) {
    if ((C & 3) == 0) {
#pragma unroll
        for (int c = 0; c < CM; c += 4)
            if (c < C) {
                const float4 t = *reinterpret_cast<const float4*>(p + c);
                v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
            } else {
                v[c] = v[c + 1] = v[c + 2] = v[c + 3] = 0.f;
            }
    } else {
#pragma unroll
        for (int c = 0; c < CM; ++c) v[c] = c < C ? p[c] : 0.f;
    }
}
// dropout scales of the C output columns of a row (site SITE_XADD_OUT, index row * C + c)
template <int CM>
__device__ __forceinline__ void drop_row(const DropL& d, uint64_t o0, int C, float (&v)[CM]) {
#pragma unroll
    for (int c = 0; c < CM; c += 4)
        if (c < C) {
            float s[4];
            drop4(d, SITE_XADD_OUT, o0 + c, C - c, s);
            v[c] = s[0]; v[c + 1] = s[1]; v[c + 2] = s[2]; v[c + 3] = s[3];
        } else {
            v[c] = v[c + 1] = v[c + 2] = v[c + 3] = 0.f;
        }
}

template <int CM>
__global__ __launch_bounds__(256) void xrank_q_fwd_kernel(XQDims q, const float* __restrict__ Y, const float* __restrict__ P,
                                                           const float* __restrict__ bHO, const unsigned char* __restrict__ mtxt,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ Yout, float* __restrict__ lse,
                                                           float* __restrict__ xhat, float* __restrict__ rstd, unsigned int* ticket, DropCfg drop) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int T = q.T, C = q.C, H = q.H, PW = q.PW, parts = q.parts, psh = q.psh, nch = (T + 3) >> 2;
    const int b0 = blockIdx.x * q.wpb, nw = min(q.wpb, q.B - b0), rows = nw * T;
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticket = 0u;       // the backward's last-workgroup ticket starts at zero
    {   // the windows' P rows are contiguous
        const float4* src = reinterpret_cast<const float4*>(P + (size_t)b0 * T * PW);
        for (int i = threadIdx.x; i < rows * PW / 4; i += 256) reinterpret_cast<float4*>(lds)[i] = src[i];
    }
    const DropL dl = drop_local(drop);
    float gm[CM], bt[CM], bh[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) { gm[c] = c < C ? gamma[c] : 0.f; bt[c] = c < C ? beta[c] : 0.f; bh[c] = c < C ? bHO[c] : 0.f; }
    __syncthreads();
    const float inv = 1.f / (1.f + q.kappa);
    for (int idx = threadIdx.x; idx < (rows << psh); idx += 256) {
        const int row = idx >> psh, part = idx & (parts - 1);
        const int lw = row / T, t = row - lw * T, b = b0 + lw;
        const size_t grow = (size_t)b * T + t;
        float y[CM], delta[CM];
        load_row<CM>(Y + grow * C, C, y);
        const bool live = mtxt[b] != 0;
#pragma unroll
        for (int c = 0; c < CM; ++c) delta[c] = bh[c];
        if (live) {
            for (int h = 0; h < H; ++h) {
                const float* Ph = lds + (size_t)lw * T * PW + h * q.Wd;
                float mx = -INFINITY;
                for (int j = part; j < nch; j += parts) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s = 4 * j + u;
                        if (s < T) {
                            const float* kw = Ph + s * PW;
                            float sc = kw[C];
#pragma unroll
                            for (int c = 0; c < CM; ++c)
                                if (c < C) sc = fmaf(y[c], kw[c], sc);
                            mx = fmaxf(mx, sc);
                        }
                    }
                }
                mx = group_max(mx, parts);
                float sum = 0.f, acc[CM];
#pragma unroll
                for (int c = 0; c < CM; ++c) acc[c] = 0.f;
                const uint64_t base = (((uint64_t)b * H + h) * T + t) * T;
                for (int j = part; j < nch; j += parts) {
                    float dr[4];
                    drop4(dl, SITE_XADD_ATTN, base + 4 * j, T - 4 * j, dr);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s = 4 * j + u;
                        if (s < T) {
                            const float* kw = Ph + s * PW;
                            float sc = kw[C];
#pragma unroll
                            for (int c = 0; c < CM; ++c)
                                if (c < C) sc = fmaf(y[c], kw[c], sc);
                            const float e = __expf(sc - mx);
                            sum += e;
                            const float ed = e * dr[u];
#pragma unroll
                            for (int c = 0; c < CM; ++c)
                                if (c < C) acc[c] = fmaf(ed, kw[C + 1 + c], acc[c]);
                        }
                    }
                }
                sum = group_sum(sum, parts);
                const float is = 1.f / sum;
                if (part == 0) lse[grow * H + h] = mx + __logf(sum);
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) delta[c] = fmaf(group_sum(acc[c], parts), is, delta[c]);
            }
        }
        if (part != 0) continue;
        float mu = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < C) mu += delta[c];
        mu /= (float)C;
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < C) { const float tt = delta[c] - mu; var = fmaf(tt, tt, var); }
        const float rs = 1.0f / sqrtf(var / (float)C + 1e-5f);
        rstd[grow] = rs;
        float ds[CM];
        if (live) drop_row<CM>(dl, (uint64_t)grow * C, C, ds);
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < C) {
                const float hh = (delta[c] - mu) * rs;
                const size_t o = grow * C + c;
                xhat[o] = hh;
                const float v = live ? fmaf(hh, gm[c], bt[c]) * ds[c] : 0.f;
                Yout[o] = (y[c] + q.kappa * v) * inv;
            }
    }
}

template <int CM>
__global__ __launch_bounds__(256) void xrank_q_bwd_kernel(XQDims q, const float* __restrict__ Y, const float* __restrict__ P,
                                                           const unsigned char* __restrict__ mtxt, const float* __restrict__ gamma,
                                                           const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                           const float* __restrict__ lse, const float* __restrict__ dYout,
                                                           float* __restrict__ dY, float* __restrict__ dP,
                                                           unsigned short* __restrict__ dP16, float* __restrict__ slabs, unsigned int* ticket,
                                                           float* __restrict__ g_lnw, float* __restrict__ g_lnb, float* __restrict__ g_bho,
                                                           DropCfg drop) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[4][64];
    __shared__ int s_last;
    const int T = q.T, C = q.C, H = q.H, PW = q.PW, TB = q.TB, parts = q.parts, psh = q.psh, nch = (T + 3) >> 2;
    const int b0 = blockIdx.x * q.wpb, nw = min(q.wpb, q.B - b0), rows = nw * T;
    float* Ps = lds;                                   // [rows][PW]
    float* Ys = Ps + (size_t)q.wpb * T * PW;           // [rows][C]
    float* Ds = Ys + (size_t)q.wpb * T * C;            // ddelta [rows][C]
    float* Ls = Ds + (size_t)q.wpb * T * C;            // lse [rows][H]
    float* Dd = Ls + (size_t)q.wpb * T * H;            // D [rows][H]
    unsigned char* Mk = reinterpret_cast<unsigned char*>(Dd + (size_t)q.wpb * T * H);      // keep bits [rows][H][TB]: 4 keys per byte
    {
        const float4* src = reinterpret_cast<const float4*>(P + (size_t)b0 * T * PW);
        for (int i = threadIdx.x; i < rows * PW / 4; i += 256) reinterpret_cast<float4*>(Ps)[i] = src[i];
    }
    const DropL dl = drop_local(drop);
    float gm[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) gm[c] = c < C ? gamma[c] : 0.f;
    __syncthreads();
    const float inv = 1.f / (1.f + q.kappa);
    float s_w[CM], s_b[CM], s_d[CM];       // partial sums: d ln_w, d ln_b, d b_HO
#pragma unroll
    for (int c = 0; c < CM; ++c) s_w[c] = s_b[c] = s_d[c] = 0.f;
    // ---- phase 1: group = query row, its lanes split the keys in chunks of four
    for (int idx = threadIdx.x; idx < (rows << psh); idx += 256) {
        const int row = idx >> psh, part = idx & (parts - 1);
        const int lw = row / T, t = row - lw * T, b = b0 + lw;
        const size_t grow = (size_t)b * T + t;
        const bool live = mtxt[b] != 0;
        float y[CM], gy[CM], dd[CM], dy[CM];
        load_row<CM>(Y + grow * C, C, y);
        load_row<CM>(dYout + grow * C, C, gy);
        {
            float xh[CM], dsc[CM], g[CM], m1 = 0.f, m2 = 0.f;
            float rs = 0.f;
            if (live) {
                load_row<CM>(xhat + grow * C, C, xh);
                rs = rstd[grow];
                drop_row<CM>(dl, (uint64_t)grow * C, C, dsc);
            }
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                dy[c] = part == 0 ? gy[c] * inv : 0.f;
                g[c] = 0.f;
                if (c < C && live) {
                    const float dn = q.kappa * inv * gy[c] * dsc[c];
                    if (part == 0) { s_w[c] = fmaf(dn, xh[c], s_w[c]); s_b[c] += dn; }
                    g[c] = dn * gm[c];
                    m1 += g[c];
                    m2 = fmaf(g[c], xh[c], m2);
                } else if (!live) {
                    xh[c] = 0.f;
                }
            }
            m1 /= (float)C; m2 /= (float)C;
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                dd[c] = (c < C && live) ? rs * (g[c] - m1 - xh[c] * m2) : 0.f;
                if (part == 0) {
                    s_d[c] += dd[c];
                    if (c < C) { Ys[row * C + c] = y[c]; Ds[row * C + c] = dd[c]; }
                }
            }
        }
        if (live) {
            for (int h = 0; h < H; ++h) {
                const float* Ph = Ps + (size_t)lw * T * PW + h * q.Wd;
                const float l = lse[grow * H + h];
                const uint64_t base = (((uint64_t)b * H + h) * T + t) * T;
                unsigned char* mk = Mk + ((size_t)row * H + h) * TB;
                // D = sum_s A_drop[t,s] dA[t,s] from the SAME a and dA values the gradient below uses: where the text rows of a window
                // are nearly equal, dA - D is a difference of nearly equal numbers, and a D formed any other way (e.g. from the
                // forward's output) leaves its own rounding in every dS
                float D = 0.f;
                for (int j = part; j < nch; j += parts) {
                    float dr[4];
                    drop4(dl, SITE_XADD_ATTN, base + 4 * j, T - 4 * j, dr);
                    unsigned int bits = 0u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s = 4 * j + u;
                        if (s < T) {
                            const float* kw = Ph + s * PW;
                            float sc = kw[C], dA = 0.f;
#pragma unroll
                            for (int c = 0; c < CM; ++c)
                                if (c < C) { sc = fmaf(y[c], kw[c], sc); dA = fmaf(dd[c], kw[C + 1 + c], dA); }
                            D = fmaf(__expf(sc - l) * dr[u], dA, D);
                            if (dr[u] != 0.f) bits |= 1u << u;
                        }
                    }
                    mk[j] = (unsigned char)bits;
                }
                D = group_sum(D, parts);
                if (part == 0) { Ls[row * H + h] = l; Dd[row * H + h] = D; }
                for (int j = part; j < nch; j += parts) {
                    const unsigned int bits = mk[j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s = 4 * j + u;
                        if (s < T) {
                            const float* kw = Ph + s * PW;
                            float sc = kw[C], dA = 0.f;
#pragma unroll
                            for (int c = 0; c < CM; ++c)
                                if (c < C) { sc = fmaf(y[c], kw[c], sc); dA = fmaf(dd[c], kw[C + 1 + c], dA); }
                            const float dr = (bits >> u) & 1u ? dl.inv_keep : 0.f;
                            const float ds = __expf(sc - l) * (dA * dr - D);
#pragma unroll
                            for (int c = 0; c < CM; ++c)
                                if (c < C) dy[c] = fmaf(ds, kw[c], dy[c]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < C) {
                const float v = group_sum(dy[c], parts);
                if (part == 0) dY[grow * C + c] = v;
            }
    }
    __syncthreads();
    // ---- phase 2: group = key row, its lanes split the query rows
    for (int idx = threadIdx.x; idx < (rows << psh); idx += 256) {
        const int row = idx >> psh, part = idx & (parts - 1);
        const int lw = row / T, s = row - lw * T, b = b0 + lw;
        const size_t grow = (size_t)b * T + s;
        const bool live = mtxt[b] != 0;
        float* out = dP + grow * PW;
        unsigned short* out16 = dP16 ? dP16 + grow * PW : nullptr;
        for (int h = 0; h < H; ++h) {
            float kw[CM], vw[CM], dkw[CM], dvw[CM], dkb = 0.f;
#pragma unroll
            for (int c = 0; c < CM; ++c) { kw[c] = 0.f; vw[c] = 0.f; dkw[c] = 0.f; dvw[c] = 0.f; }
            if (live) {
                const float* Ph = Ps + ((size_t)lw * T + s) * PW + h * q.Wd;
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) { kw[c] = Ph[c]; vw[c] = Ph[C + 1 + c]; }
                const float kb = Ph[C];
                for (int t = part; t < T; t += parts) {
                    const int r2 = lw * T + t;
                    const float* yt = Ys + r2 * C;
                    const float* dt = Ds + r2 * C;
                    float sc = kb, dA = 0.f;
#pragma unroll
                    for (int c = 0; c < CM; ++c)
                        if (c < C) { sc = fmaf(yt[c], kw[c], sc); dA = fmaf(dt[c], vw[c], dA); }
                    const float a = __expf(sc - Ls[r2 * H + h]);
                    const float dr = (Mk[((size_t)r2 * H + h) * TB + (s >> 2)] >> (s & 3)) & 1u ? dl.inv_keep : 0.f;
                    const float ds = a * (dA * dr - Dd[r2 * H + h]);
                    const float ad = a * dr;
                    dkb += ds;
#pragma unroll
                    for (int c = 0; c < CM; ++c)
                        if (c < C) { dkw[c] = fmaf(ds, yt[c], dkw[c]); dvw[c] = fmaf(ad, dt[c], dvw[c]); }
                }
            }
            dkb = group_sum(dkb, parts);
#pragma unroll
            for (int c = 0; c < CM; ++c)
                if (c < C) { dkw[c] = group_sum(dkw[c], parts); dvw[c] = group_sum(dvw[c], parts); }
            if (part == 0) {
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) {
                        out[h * q.Wd + c] = dkw[c];
                        out[h * q.Wd + C + 1 + c] = dvw[c];
                        if (out16) { out16[h * q.Wd + c] = f2bf(dkw[c]); out16[h * q.Wd + C + 1 + c] = f2bf(dvw[c]); }
                    }
                out[h * q.Wd + C] = dkb;
                if (out16) out16[h * q.Wd + C] = f2bf(dkb);
            }
        }
        if (part == 0)
            for (int c = H * q.Wd; c < PW; ++c) {
                out[c] = 0.f;
                if (out16) out16[c] = 0;
            }
    }
    // ---- the three C-vectors: workgroup sums -> slab; the last workgroup to finish adds the slabs in index order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        float mine = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            const float a = wave_sum(s_w[c]), bb = wave_sum(s_b[c]), d2 = wave_sum(s_d[c]);
            if (c < C) {
                if (lane == c) mine = a;
                if (lane == C + c) mine = bb;
                if (lane == 2 * C + c) mine = d2;
            }
        }
        red[wave][lane] = mine;
    }
    __syncthreads();
    if ((int)threadIdx.x < 3 * C)
        slabs[(size_t)blockIdx.x * 3 * C + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    {   // value = lane (< 3C), the four waves take every fourth slab
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const int G = gridDim.x;
        if (lane < 3 * C) {
            int g = wave;
            for (; g + 12 < G; g += 16) {
                a0 += __builtin_nontemporal_load(slabs + (size_t)g * 3 * C + lane);
                a1 += __builtin_nontemporal_load(slabs + (size_t)(g + 4) * 3 * C + lane);
                a2 += __builtin_nontemporal_load(slabs + (size_t)(g + 8) * 3 * C + lane);
                a3 += __builtin_nontemporal_load(slabs + (size_t)(g + 12) * 3 * C + lane);
            }
            for (; g < G; g += 4) a0 += __builtin_nontemporal_load(slabs + (size_t)g * 3 * C + lane);
        }
        red[wave][lane] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if ((int)threadIdx.x < 3 * C) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        const int k = threadIdx.x / C, c = threadIdx.x - k * C;
        (k == 0 ? g_lnw : k == 1 ? g_lnb : g_bho)[c] = v;
    }
    if (threadIdx.x == 0) *ticket = 0u;
}

// ---- training step: forward, masked-MSE loss (lib/evaluation.py:17-62 with the per-variable observation counts known beforehand, as
// immtsf_masked_mse_counted) and backward of the Q half in ONE launch.  Every quantity the backward needs from the forward and from
// the loss is local to a row once the counts are given, so a workgroup runs its windows' forward, forms d loss / d Y_out in registers
// and goes straight on: the serial section between the backbone's forward and backward is one kernel instead of three (and the
// LayerNorm statistics, the attention's log-sum-exp and d loss / d Y_out never travel through memory).
// CM floats from a 16-byte aligned LDS address
template <int CM>
__device__ __forceinline__ void lds_vec(const float* p, float (&v)[CM]) {
#pragma unroll
    for (int c = 0; c < CM; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + c);
        v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
    }
}
template <int CM>
__device__ __forceinline__ void lds_put(float* p, const float (&v)[CM]) {
#pragma unroll
    for (int c = 0; c < CM; c += 4) *reinterpret_cast<float4*>(p + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
}

// The key rows sit in LDS PADDED to the template width (XQ_PL(CM) floats per (head, key): k weights | v weights | k bias, zeros beyond
// C and beyond T): every inner loop is unconditional float4 LDS reads and FMAs over CM columns.  (With `if (c < C)` around each
// element the compiler emitted a scalar branch and a dependent LDS round trip per element: 2600 branches, 21 000 lines of ISA, 32 us
// for 64 windows; this form: see DESIGN.md section 4c.)
template <int CM>
__global__ __launch_bounds__(256) void xrank_q_train_kernel(XQDims q, const float* __restrict__ Y, const float* __restrict__ P,
                                                             const float* __restrict__ bHO, const unsigned char* __restrict__ mtxt,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ truth, const float* __restrict__ tmask,
                                                             const float* __restrict__ cnt, float grad_scale, float* __restrict__ Yout,
                                                             float* __restrict__ dY, float* __restrict__ dP,
                                                             unsigned short* __restrict__ dP16, float* __restrict__ slabs, unsigned int* ticket,
                                                             float* __restrict__ g_lnw, float* __restrict__ g_lnb, float* __restrict__ g_bho,
                                                             float* __restrict__ loss, int* done_flag, DropCfg drop) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[4][64];
    __shared__ int s_last;
    __shared__ unsigned char s_live[256];      // the windows' M_txt flags (wpb <= 256)
    constexpr int PL = XQ_PL(CM);
    const int T = q.T, C = q.C, H = q.H, PW = q.PW, TB = q.TB, parts = q.parts, psh = q.psh, nch = (T + 3) >> 2, TP = nch * 4;
    const int b0 = blockIdx.x * q.wpb, nw = min(q.wpb, q.B - b0), rows = nw * T, total = rows << psh;
    float* Ks = lds;                                       // [wpb][H][TP][PL]
    float* Ys = Ks + (size_t)q.wpb * H * TP * PL;          // [rows][CM]
    float* Ds = Ys + (size_t)q.wpb * T * CM;               // ddelta [rows][CM]
    float* Ls = Ds + (size_t)q.wpb * T * CM;               // lse [rows][H]
    float* Dd = Ls + (size_t)q.wpb * T * H;                // D [rows][H]
    unsigned char* Mk = reinterpret_cast<unsigned char*>(Dd + (size_t)q.wpb * T * H);      // keep bits [rows][H][TB]: 4 keys per byte
    // the first row's global operands are on their way while the key rows are staged
    int idx = threadIdx.x;
    float y[CM], tr[CM], tm[CM];
    if (idx < total) {
        const size_t grow = (size_t)b0 * T + (idx >> psh);
        load_row<CM>(Y + grow * C, C, y);
        load_row<CM>(truth + grow * C, C, tr);
        load_row<CM>(tmask + grow * C, C, tm);
    }
    if ((int)threadIdx.x < nw) s_live[threadIdx.x] = mtxt[b0 + threadIdx.x];
    for (int i = threadIdx.x; i < nw * H * TP; i += 256) {       // a thread per (window, head, key): its 2C+1 loads in flight together
        const int s = i % TP, r = i / TP;
        const int h = r % H, lw = r / H;
        const float* src = P + ((size_t)(b0 + lw) * T + (s < T ? s : 0)) * PW + h * q.Wd;
        float kw[CM], vw[CM], kb[4];
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            const int cc = c < C ? c : 0;
            kw[c] = src[cc];
            vw[c] = src[C + 1 + cc];
        }
        kb[0] = src[C]; kb[1] = kb[2] = kb[3] = 0.f;
        const bool in = s < T;
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            kw[c] = (in && c < C) ? kw[c] : 0.f;
            vw[c] = (in && c < C) ? vw[c] : 0.f;
        }
        kb[0] = in ? kb[0] : 0.f;
        float* dst = Ks + (size_t)i * PL;
        lds_put<CM>(dst, kw);
        lds_put<CM>(dst + CM, vw);
        lds_put<4>(dst + 2 * CM, kb);
    }
    const DropL dl = drop_local(drop);
    float gm[CM], bt[CM], bh[CM], lsc[CM], navail = 0.f;      // lsc: 1 / (count + 1e-8) of a variable (0 beyond C)
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        gm[c] = c < C ? gamma[c] : 0.f; bt[c] = c < C ? beta[c] : 0.f; bh[c] = c < C ? bHO[c] : 0.f;
        const float n_c = c < C ? cnt[c] : 0.f;
        lsc[c] = c < C ? 1.f / (n_c + 1e-8f) : 0.f;
        navail += (c < C && n_c != 0.f) ? 1.f : 0.f;
    }
    float s_e = 0.f;      // partial loss
    __syncthreads();
    const float inv = 1.f / (1.f + q.kappa);
    float s_w[CM], s_b[CM], s_d[CM];       // partial sums: d ln_w, d ln_b, d b_HO
#pragma unroll
    for (int c = 0; c < CM; ++c) s_w[c] = s_b[c] = s_d[c] = 0.f;
    // ---- phase 1: group = query row, its lanes split the keys in chunks of four
    for (; idx < total; idx += 256) {
        const int row = idx >> psh, part = idx & (parts - 1);
        const int lw = row / T, t = row - lw * T, b = b0 + lw;
        const size_t grow = (size_t)b * T + t;
        const bool live = s_live[lw] != 0;
        float gy[CM], dd[CM], dy[CM], delta[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) delta[c] = bh[c];
        if (live) {       // ---- forward of the row
#pragma unroll 1
            for (int h = 0; h < H; ++h) {
                const float* Kh = Ks + (size_t)(lw * H + h) * TP * PL;
                float mx = -INFINITY;
#pragma unroll 1
                for (int j = part; j < nch; j += parts) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s = 4 * j + u;
                        const float* kr = Kh + s * PL;
                        float kw[CM];
                        lds_vec<CM>(kr, kw);
                        float sc = kr[2 * CM];
#pragma unroll
                        for (int c = 0; c < CM; ++c) sc = fmaf(y[c], kw[c], sc);
                        mx = fmaxf(mx, s < T ? sc : -INFINITY);
                    }
                }
                mx = group_max(mx, parts);
                float sum = 0.f, acc[CM];
#pragma unroll
                for (int c = 0; c < CM; ++c) acc[c] = 0.f;
                const uint64_t base = (((uint64_t)b * H + h) * T + t) * T;
                unsigned char* mk = Mk + ((size_t)row * H + h) * TB;
#pragma unroll 1
                for (int j = part; j < nch; j += parts) {
                    float dr[4];
                    drop4(dl, SITE_XADD_ATTN, base + 4 * j, T - 4 * j, dr);
                    unsigned int bits = 0u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s = 4 * j + u;
                        const float* kr = Kh + s * PL;
                        float kw[CM], vw[CM];
                        lds_vec<CM>(kr, kw);
                        lds_vec<CM>(kr + CM, vw);
                        float sc = kr[2 * CM];
#pragma unroll
                        for (int c = 0; c < CM; ++c) sc = fmaf(y[c], kw[c], sc);
                        const float e = s < T ? __expf(sc - mx) : 0.f;
                        sum += e;
                        const float ed = e * dr[u];
#pragma unroll
                        for (int c = 0; c < CM; ++c) acc[c] = fmaf(ed, vw[c], acc[c]);
                        bits |= (dr[u] != 0.f ? 1u : 0u) << u;
                    }
                    mk[j] = (unsigned char)bits;        // the keep bits of the row's keys, for both backward phases
                }
                sum = group_sum(sum, parts);
                const float is = 1.f / sum;
                Ls[row * H + h] = mx + __logf(sum);     // (every lane of the group: the same value)
#pragma unroll
                for (int c = 0; c < CM; ++c) delta[c] = fmaf(group_sum(acc[c], parts), is, delta[c]);
            }
        }
        {   // ---- LayerNorm(C), dropout, blend; the loss and its gradient; LayerNorm backward -- all in the row's registers
            float mu = 0.f;
#pragma unroll
            for (int c = 0; c < CM; ++c) mu += delta[c];          // (0 beyond C)
            mu /= (float)C;
            float var = 0.f;
#pragma unroll
            for (int c = 0; c < CM; ++c) { const float tt = c < C ? delta[c] - mu : 0.f; var = fmaf(tt, tt, var); }
            const float rs = 1.0f / sqrtf(var / (float)C + 1e-5f);
            float xh[CM], dsc[CM], g[CM], m1 = 0.f, m2 = 0.f;
            if (live) drop_row<CM>(dl, (uint64_t)grow * C, C, dsc);
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                const bool on = c < C && live;
                xh[c] = (delta[c] - mu) * rs;
                const float v = on ? fmaf(xh[c], gm[c], bt[c]) * dsc[c] : 0.f;
                const float yo = (y[c] + q.kappa * v) * inv;
                if (part == 0 && c < C && Yout) Yout[grow * C + c] = yo;
                const float dlt = tr[c] - yo;
                if (part == 0) s_e = fmaf(dlt * dlt, tm[c] * lsc[c], s_e);
                gy[c] = c < C ? -dlt * tm[c] * lsc[c] * (grad_scale * 2.f / navail) : 0.f;
                dy[c] = part == 0 ? gy[c] * inv : 0.f;
                const float dn = on ? q.kappa * inv * gy[c] * dsc[c] : 0.f;
                if (part == 0) { s_w[c] = fmaf(dn, on ? xh[c] : 0.f, s_w[c]); s_b[c] += dn; }
                g[c] = dn * gm[c];
                m1 += g[c];
                m2 = fmaf(g[c], on ? xh[c] : 0.f, m2);
            }
            m1 /= (float)C; m2 /= (float)C;
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                dd[c] = (c < C && live) ? rs * (g[c] - m1 - xh[c] * m2) : 0.f;
                if (part == 0) s_d[c] += dd[c];
            }
            if (part == 0) { lds_put<CM>(Ys + row * CM, y); lds_put<CM>(Ds + row * CM, dd); }
        }
        if (live) {
#pragma unroll 1
            for (int h = 0; h < H; ++h) {
                const float* Kh = Ks + (size_t)(lw * H + h) * TP * PL;
                const float l = Ls[row * H + h];
                const unsigned char* mk = Mk + ((size_t)row * H + h) * TB;
                // D = sum_s A_drop[t,s] dA[t,s] from the SAME a and dA values the gradient below uses: where the text rows of a window
                // are nearly equal, dA - D is a difference of nearly equal numbers, and a D formed any other way (e.g. from the
                // forward's output) leaves its own rounding in every dS.  (Keys beyond T are zero rows: dA = 0 and k weights = 0, so
                // they add nothing to D or to dY.)
                float D = 0.f;
#pragma unroll 1
                for (int j = part; j < nch; j += parts) {
                    const unsigned int bits = mk[j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float* kr = Kh + (4 * j + u) * PL;
                        float kw[CM], vw[CM];
                        lds_vec<CM>(kr, kw);
                        lds_vec<CM>(kr + CM, vw);
                        float sc = kr[2 * CM], dA = 0.f;
#pragma unroll
                        for (int c = 0; c < CM; ++c) { sc = fmaf(y[c], kw[c], sc); dA = fmaf(dd[c], vw[c], dA); }
                        D = fmaf(__expf(sc - l) * ((bits >> u) & 1u ? dl.inv_keep : 0.f), dA, D);
                    }
                }
                D = group_sum(D, parts);
                if (part == 0) Dd[row * H + h] = D;
#pragma unroll 1
                for (int j = part; j < nch; j += parts) {
                    const unsigned int bits = mk[j];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float* kr = Kh + (4 * j + u) * PL;
                        float kw[CM], vw[CM];
                        lds_vec<CM>(kr, kw);
                        lds_vec<CM>(kr + CM, vw);
                        float sc = kr[2 * CM], dA = 0.f;
#pragma unroll
                        for (int c = 0; c < CM; ++c) { sc = fmaf(y[c], kw[c], sc); dA = fmaf(dd[c], vw[c], dA); }
                        const float dr = (bits >> u) & 1u ? dl.inv_keep : 0.f;
                        const float ds = __expf(sc - l) * (dA * dr - D);
#pragma unroll
                        for (int c = 0; c < CM; ++c) dy[c] = fmaf(ds, kw[c], dy[c]);
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            const float v = group_sum(dy[c], parts);
            if (part == 0 && c < C) dY[grow * C + c] = v;
        }
        const int nxt = idx + 256;        // the next row's operands
        if (nxt < total) {
            const size_t g2 = (size_t)b0 * T + (nxt >> psh);
            load_row<CM>(Y + g2 * C, C, y);
            load_row<CM>(truth + g2 * C, C, tr);
            load_row<CM>(tmask + g2 * C, C, tm);
        }
    }
    __syncthreads();
    // dY_ts is complete for this workgroup: the LAST workgroup to get here publishes `done_flag` (a device flag another stream's
    // spin kernel waits on, immtsf_flag_wait) -- the backbone's backward starts while phase 2 and the reductions still run
    if (done_flag && threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(ticket + 1, 1u) == gridDim.x - 1) {
            ticket[1] = 0u;
            __threadfence();
            __hip_atomic_store(done_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- phase 2: group = key row, its lanes split the query rows
    for (int i2 = threadIdx.x; i2 < total; i2 += 256) {
        const int row = i2 >> psh, part = i2 & (parts - 1);
        const int lw = row / T, s = row - lw * T, b = b0 + lw;
        const size_t grow = (size_t)b * T + s;
        const bool alive = s_live[lw] != 0;
        float* out = dP + grow * PW;
        unsigned short* out16 = dP16 ? dP16 + grow * PW : nullptr;
#pragma unroll 1
        for (int h = 0; h < H; ++h) {
            float dkw[CM], dvw[CM], dkb = 0.f;
#pragma unroll
            for (int c = 0; c < CM; ++c) { dkw[c] = 0.f; dvw[c] = 0.f; }
            if (alive) {
                const float* kr = Ks + ((size_t)(lw * H + h) * TP + s) * PL;
                float kw[CM], vw[CM];
                lds_vec<CM>(kr, kw);
                lds_vec<CM>(kr + CM, vw);
                const float kb = kr[2 * CM];
#pragma unroll 2
                for (int t = part; t < T; t += parts) {
                    const int r2 = lw * T + t;
                    float yt[CM], dt[CM];
                    lds_vec<CM>(Ys + r2 * CM, yt);
                    lds_vec<CM>(Ds + r2 * CM, dt);
                    float sc = kb, dA = 0.f;
#pragma unroll
                    for (int c = 0; c < CM; ++c) { sc = fmaf(yt[c], kw[c], sc); dA = fmaf(dt[c], vw[c], dA); }
                    const float a = __expf(sc - Ls[r2 * H + h]);
                    const float dr = (Mk[((size_t)r2 * H + h) * TB + (s >> 2)] >> (s & 3)) & 1u ? dl.inv_keep : 0.f;
                    const float ds = a * (dA * dr - Dd[r2 * H + h]);
                    const float ad = a * dr;
                    dkb += ds;
#pragma unroll
                    for (int c = 0; c < CM; ++c) { dkw[c] = fmaf(ds, yt[c], dkw[c]); dvw[c] = fmaf(ad, dt[c], dvw[c]); }
                }
            }
            dkb = group_sum(dkb, parts);
#pragma unroll
            for (int c = 0; c < CM; ++c) { dkw[c] = group_sum(dkw[c], parts); dvw[c] = group_sum(dvw[c], parts); }
            if (part == 0) {
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) {
                        out[h * q.Wd + c] = dkw[c];
                        out[h * q.Wd + C + 1 + c] = dvw[c];
                        if (out16) { out16[h * q.Wd + c] = f2bf(dkw[c]); out16[h * q.Wd + C + 1 + c] = f2bf(dvw[c]); }
                    }
                out[h * q.Wd + C] = dkb;
                if (out16) out16[h * q.Wd + C] = f2bf(dkb);
            }
        }
        if (part == 0)
            for (int c = H * q.Wd; c < PW; ++c) {
                out[c] = 0.f;
                if (out16) out16[c] = 0;
            }
    }
    // ---- the three C-vectors: workgroup sums -> slab; the last workgroup to finish adds the slabs in index order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        float mine = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            const float a = wave_sum(s_w[c]), bb = wave_sum(s_b[c]), d2 = wave_sum(s_d[c]);
            if (c < C) {
                if (lane == c) mine = a;
                if (lane == C + c) mine = bb;
                if (lane == 2 * C + c) mine = d2;
            }
        }
        const float e2 = wave_sum(s_e);
        if (lane == 3 * C) mine = e2;
        red[wave][lane] = mine;
    }
    __syncthreads();
    const int NV = 3 * C + 1;
    if ((int)threadIdx.x < NV)
        slabs[(size_t)blockIdx.x * NV + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    {   // value = lane (< 3C + 1), the four waves take every fourth slab, eight loads in flight
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = 0.f;
        const int G = gridDim.x;
        if (lane < NV) {
            int g = wave;
            for (; g + 28 < G; g += 32) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += __builtin_nontemporal_load(slabs + (size_t)(g + 4 * u) * NV + lane);
            }
            for (; g < G; g += 4) a[0] += __builtin_nontemporal_load(slabs + (size_t)g * NV + lane);
        }
        red[wave][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    __syncthreads();
    if ((int)threadIdx.x < NV) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        const int k = threadIdx.x / C, c = threadIdx.x - k * C;
        if (k == 3) loss[0] = v / navail;
        else (k == 0 ? g_lnw : k == 1 ? g_lnb : g_bho)[c] = v;
    }
    if (threadIdx.x == 0) *ticket = 0u;
}

inline XQDims xq_dims(const immtsf_fusion_cfg* c) {
    const XRDims x = xr_dims(c);
    XQDims q;
    q.B = x.B; q.T = x.T; q.C = x.C; q.H = x.H; q.PW = x.PW; q.Wd = x.Wd; q.wpb = xq_wpb(x); q.TB = xq_mask_bytes(x.T); q.kappa = c->kappa;
    // lanes per row: as many as 256 threads give the workgroup's rows, at most 8 (and one chunk of four keys each)
    int parts = 1, psh = 0;
    const int nch = (x.T + 3) / 4;
    while (parts < 8 && q.wpb * x.T * parts * 2 <= 256 && parts * 2 <= nch) { parts *= 2; ++psh; }
    q.parts = parts; q.psh = psh;
    return q;
}

// ------------------------------------------------------------------------------------------------ rank-PW expansion
// C (M, N) = A (M, K) B (K, N) with K = PW <= 32 (the low-rank projection's data gradient: dZ = dP W_c, 131 072 x 768 x 24 at 4096
// windows).  As a GEMM it is one 64-deep K step of padding per tile and then all epilogue: 532 us for 403 MB of output on gemm2's
// 256 x 128 tiles.  It is a row kernel: a thread keeps its four columns of B in registers (K float4), the workgroup's rows of A sit
// in LDS, every output row is K broadcast-FMAs per lane and one 16-byte store (+ the bf16 image when asked).  Exact fp32.
constexpr int RX_ROWS = 32, RX_KMAX = 32;
template <int K>
__global__ __launch_bounds__(256) void rank_expand_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb,
                                                           float* __restrict__ Cm, bf16_t* __restrict__ Ch, int M, int N) {
    __shared__ float As[RX_ROWS][K];
    const int r0 = blockIdx.x * RX_ROWS, rows = min(RX_ROWS, M - r0), c4 = threadIdx.x;
    for (int i = threadIdx.x; i < RX_ROWS * K; i += blockDim.x) {
        const int r = i / K, k = i - r * K;
        As[r][k] = r < rows ? A[(size_t)(r0 + r) * lda + k] : 0.f;
    }
    float4 b[K];
    const bool live = c4 * 4 < N;
#pragma unroll
    for (int k = 0; k < K; ++k) b[k] = live ? *reinterpret_cast<const float4*>(Bm + (size_t)k * ldb + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (!live) return;
    for (int r = 0; r < rows; ++r) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float a = As[r][k];
            acc.x = fmaf(a, b[k].x, acc.x); acc.y = fmaf(a, b[k].y, acc.y); acc.z = fmaf(a, b[k].z, acc.z); acc.w = fmaf(a, b[k].w, acc.w);
        }
        *reinterpret_cast<float4*>(Cm + (size_t)(r0 + r) * N + c4 * 4) = acc;
        if (Ch) {
            const bf16x4 h = {(bf16_t)acc.x, (bf16_t)acc.y, (bf16_t)acc.z, (bf16_t)acc.w};
            *reinterpret_cast<bf16x4*>(Ch + (size_t)(r0 + r) * N + c4 * 4) = h;
        }
    }
}
}  // namespace
// (external linkage: gr_train.hip uses the pair too)
// applies from 4096 rows on (below, the GEMM's few tiles are as quick), K in {8, 16, 24, 32}, N % 4 == 0, N <= 1024, dense 16-byte aligned C
bool rank_expand_ok(int M, int N, int K, const void* Bm, int ldb, const void* Cm, const void* Ch) {
    return M >= 4096 && (K == 8 || K == 16 || K == 24 || K == 32) && (N & 3) == 0 && N <= 1024 && (ldb & 3) == 0 &&
           ((reinterpret_cast<uintptr_t>(Bm) | reinterpret_cast<uintptr_t>(Cm)) & 15) == 0 && (reinterpret_cast<uintptr_t>(Ch) & 7) == 0;
}
int launch_rank_expand(const float* A, int lda, const float* Bm, int ldb, float* Cm, void* Ch, int M, int N, int K, hipStream_t s) {
    const dim3 grid(cdiv(M, RX_ROWS)), block(((N / 4 + 63) / 64) * 64);
#define RX(KK) hipLaunchKernelGGL((rank_expand_kernel<KK>), grid, block, 0, s, A, lda, Bm, ldb, Cm, static_cast<bf16_t*>(Ch), M, N)
    if (K == 8) RX(8); else if (K == 16) RX(16); else if (K == 24) RX(24); else RX(32);
#undef RX
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
namespace {

}  // namespace

extern "C" {

/* row pitch (floats) of P / dP, 0 when the shape is outside the low-rank path's limits (C <= 15, H <= 4, (2C+1) H <= 128, d % 4 == 0,
 * one window's rows within 60 KB of LDS) */
int32_t immtsf_mmf_xrank_pw(const immtsf_fusion_cfg* cfg) { return xr_supported(cfg) ? xr_dims(cfg).PW : 0; }
size_t immtsf_mmf_xrank_p_workspace_bytes(const immtsf_fusion_cfg* cfg) { return xr_supported(cfg) ? carve_xp(cfg, nullptr).bytes : 0; }
size_t immtsf_mmf_xrank_p_scratch_bytes(const immtsf_fusion_cfg* cfg) { return xr_supported(cfg) ? carve_xp_scratch(cfg, nullptr).bytes : 0; }
size_t immtsf_mmf_xrank_q_workspace_bytes(const immtsf_fusion_cfg* cfg) { return xr_supported(cfg) ? carve_xq(cfg, nullptr).bytes : 0; }

/* the fold alone (parameters only: W_fold, its bias, its bf16 image and the factors the backward needs go to `workspace`, b_HO to
 * bHO): may run ahead of time on ANY stream -- e.g. a third one beside the text side and the backbone -- before
 * immtsf_mmf_xrank_p_forward(..., folded = 1, ...) on a stream ordered behind it */
int immtsf_mmf_xrank_fold(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, float* bHO, void* workspace, size_t workspace_bytes,
                          immtsf_stream_t stream) {
    if (!xr_supported(cfg) || !p || !bHO || !workspace) return IMMTSF_EINVAL;
    XPWs w = carve_xp(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const XRDims x = xr_dims(cfg);
    const int d = x.d, C = x.C, Cq = x.Cq, H = x.H, E = x.E;
    const bool hf = xr_hf(cfg);
    const float scale = sqrtf(1.0f / (float)E);
    const float *Wiq = p->attn_in_w, *Wik = p->attn_in_w + (size_t)d * d, *Wiv = p->attn_in_w + (size_t)2 * d * d;
    const float *biq = p->attn_in_b, *bik = p->attn_in_b + d, *biv = p->attn_in_b + 2 * d;
    {   // AqT = [W_in,q W_q | b_q]^T ((C+1) x d);  W_HO = W_res W_out (C x d);  b_HO = W_res b_out + b_res
        XJobList L;
        L.nt(p->proj_q_w, 1, C, C, Wiq, d, d, d, w.AqT, d, 1.f, nullptr, nullptr, nullptr, biq);
        L.nn(p->res_w, d, C, p->attn_out_w, d, d, d, w.WHO, d, 1.f);
        L.nt(p->res_w, d, 1, C, p->attn_out_b, d, d, 1, bHO, 1, 1.f, nullptr, nullptr, p->res_b);
        CHECK(L.launch(s, true));
    }
    {   // per head: GA_h = AqT[:, head] W_in,k[head, :],  UA_h = W_HO[:, head] W_in,v[head, :]  and the bias columns of W_fold
        XJobList L;
        for (int h = 0; h < H; ++h) {
            const size_t o = (size_t)h * E;
            L.nn(w.AqT + o, d, Cq, Wik + o * d, d, E, d, w.GA + (size_t)h * Cq * d, d, 1.f);
            L.nt(w.AqT + o, d, 1, Cq, bik + o, E, E, 1, w.Wfb + h * x.Wd, 1, scale);
            L.nn(w.WHO + o, d, C, Wiv + o * d, d, E, d, w.UA + (size_t)h * C * d, d, 1.f);
            L.nt(w.WHO + o, d, 1, C, biv + o, E, E, 1, w.Wfb + h * x.Wd + Cq, 1, 1.f);
        }
        if (x.PW > H * x.Wd) L.copy(nullptr, x.PW - H * x.Wd, w.Wfb + H * x.Wd);
        CHECK(L.launch(s, true));
    }
    {   // W_fold rows: G_h = scale GA_h W_k,  U_h = UA_h W_v  (+ the bf16 image, zero rows up to PW)
        XJobList L;
        for (int h = 0; h < H; ++h) {
            const size_t r0 = (size_t)h * x.Wd;
            L.nn(w.GA + (size_t)h * Cq * d, d, Cq, p->proj_k_w, d, d, d, w.Wf + r0 * d, d, scale, hf ? w.Wf16 + r0 * d : nullptr);
            L.nn(w.UA + (size_t)h * C * d, d, C, p->proj_v_w, d, d, d, w.Wf + (r0 + Cq) * d, d, 1.f, hf ? w.Wf16 + (r0 + Cq) * d : nullptr);
        }
        if (x.PW > H * x.Wd) L.copy(nullptr, (x.PW - H * x.Wd) * d, w.Wf + (size_t)H * x.Wd * d, hf ? w.Wf16 + (size_t)H * x.Wd * d : nullptr);
        CHECK(L.launch(s, true));
    }
    return IMMTSF_OK;
}

int immtsf_mmf_xrank_p_forward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt, float* P, float* bHO,
                               void* workspace, size_t workspace_bytes, int32_t folded, immtsf_stream_t stream) {
    if (!xr_supported(cfg) || !p || !E_txt || !P || !bHO || !workspace) return IMMTSF_EINVAL;
    XPWs w = carve_xp(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    if (!folded) CHECK(immtsf_mmf_xrank_fold(cfg, p, bHO, workspace, workspace_bytes, stream));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const XRDims x = xr_dims(cfg);
    const int d = x.d, BT = x.B * x.T, prec = cfg->precision;
    const bool hf = xr_hf(cfg);
    Mat Em = cmat(E_txt);
    if (hf && cfg->in_h) {
        Em.h = const_cast<void*>(cfg->in_h);
    } else if (hf) {
        CHECK(launch_f32_to_bf16(E_txt, w.E16, (size_t)BT * d, s));
        Em.h = w.E16;
    }
    {   // P = E W_fold^T + b_fold
        GemmArgs g = gemm_args(BT, x.PW, d, d, d, x.PW);
        set_problem2(g, 0, Em, mat(w.Wf, w.Wf16), mat(P), w.Wfb);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    return IMMTSF_OK;
}

/* ---- the "_z" form: the text side's producer ends in a linear map E_txt = Z W_po^T + b_po (TTF_T2V_XAttn's proj_out,
 * fusions/TTF_T2V_XAttn.py:182) whose only consumer is this projection, so P = [Z | 1] [W_fold W_po | W_fold b_po + b_fold]^T: the
 * (B T) x d x d product, its data gradient and its weight gradient become PW-row products (PW = 24 at cfg2), E_txt and dE_txt are
 * never formed.  proj_w (d, d), proj_b (d): the producer's parameters; their gradients come out of the backward here. */
static int xrank_compose_z(const immtsf_fusion_cfg* cfg, const float* proj_w, const float* proj_b, const XPWs& w, hipStream_t s) {
    const XRDims x = xr_dims(cfg);
    const int d = x.d, prec = cfg->precision;
    const bool hf = xr_hf(cfg);
    Mat Wpo;
    CHECK(weight_mat(hf, proj_w, (size_t)d * d, w.Wpo16, s, &Wpo));
    {   // Wc = W_fold W_po  (PW x d)
        GemmArgs g = gemm_args(x.PW, d, d, d, d, d);
        set_problem2(g, 0, mat(w.Wf, w.Wf16), Wpo, mat(w.Wc, w.Wc16), nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    {   // bc = W_fold b_po + b_fold
        VecJobList l;
        l.add(VJ_MV, w.Wf, d, proj_b, w.Wfb, w.bc, x.PW, d);
        CHECK(launch_vecjobs(l, s));
    }
    return IMMTSF_OK;
}

/* immtsf_mmf_xrank_fold + the composition with the producer's last linear map (Wc, bc): everything of the "_z" forward that depends on
 * parameters only -- may run ahead of time on any stream, before immtsf_mmf_xrank_p_forward_z(..., folded = 1, ...) */
int immtsf_mmf_xrank_fold_z(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* proj_w, const float* proj_b, float* bHO,
                            void* workspace, size_t workspace_bytes, immtsf_stream_t stream) {
    if (!proj_w || !proj_b) return IMMTSF_EINVAL;
    CHECK(immtsf_mmf_xrank_fold(cfg, p, bHO, workspace, workspace_bytes, stream));
    return xrank_compose_z(cfg, proj_w, proj_b, carve_xp(cfg, workspace), static_cast<hipStream_t>(stream));
}

int immtsf_mmf_xrank_p_forward_z(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* proj_w, const float* proj_b,
                                 const float* Z, float* P, float* bHO, void* workspace, size_t workspace_bytes, int32_t folded,
                                 immtsf_stream_t stream) {
    if (!xr_supported(cfg) || !p || !proj_w || !proj_b || !Z || !P || !bHO || !workspace) return IMMTSF_EINVAL;
    XPWs w = carve_xp(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    if (!folded) CHECK(immtsf_mmf_xrank_fold_z(cfg, p, proj_w, proj_b, bHO, workspace, workspace_bytes, stream));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const XRDims x = xr_dims(cfg);
    const int d = x.d, BT = x.B * x.T, prec = cfg->precision;
    const bool hf = xr_hf(cfg);
    Mat Zm = cmat(Z);
    if (hf && cfg->in_h) {
        Zm.h = const_cast<void*>(cfg->in_h);
    } else if (hf) {
        CHECK(launch_f32_to_bf16(Z, w.E16, (size_t)BT * d, s));
        Zm.h = w.E16;
    }
    {   // P = Z Wc^T + bc
        GemmArgs g = gemm_args(BT, x.PW, d, d, d, x.PW);
        set_problem2(g, 0, Zm, mat(w.Wc, w.Wc16), mat(P), w.bc);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    return IMMTSF_OK;
}

/* the data half of the "_z" backward: dZ = dP Wc, dWc = dP^T Z (+ column sums) */
int immtsf_mmf_xrank_p_backward_data_z(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* proj_w, const float* proj_b,
                                       const float* Z, const float* dP, float* dZ, void* workspace, size_t workspace_bytes, void* scratch,
                                       size_t scratch_bytes, immtsf_stream_t stream) {
    if (!xr_supported(cfg) || !p || !proj_w || !proj_b || !Z || !dP || !dZ || !workspace || !scratch) return IMMTSF_EINVAL;
    XPWs w = carve_xp(cfg, workspace);
    XPScratch sc = carve_xp_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const XRDims x = xr_dims(cfg);
    const int d = x.d, BT = x.B * x.T, prec = cfg->precision;
    const bool hf = xr_hf(cfg);
    Mat Wpo;
    CHECK(weight_mat(hf, proj_w, (size_t)d * d, w.Wpo16, s, &Wpo));
    Mat dPm = cmat(dP), Zm = cmat(Z, hf ? (cfg->aux_h ? cfg->aux_h : static_cast<const void*>(w.E16)) : nullptr);
    const bool seed_only = cfg->bwd_phase != 0 && !(cfg->bwd_phase & IMMTSF_BWD_PHASE_A);      // (the images were made by the data call)
    if (hf && !cfg->aux_h && !seed_only) CHECK(launch_f32_to_bf16(Z, w.E16, (size_t)BT * d, s));
    if (hf && cfg->in_h) {
        dPm.h = const_cast<void*>(cfg->in_h);
    } else if (hf) {
        if (!seed_only) CHECK(launch_f32_to_bf16(dP, sc.dP16, (size_t)BT * x.PW, s));
        dPm.h = sc.dP16;
    }
    // immtsf_fusion_cfg.bwd_phase: 0 = both halves; IMMTSF_BWD_PHASE_A = only dZ (what the producer's backward waits for);
    // IMMTSF_BWD_WGRAD_A = only the chain's seeds dWc / dbc (parameter-gradient work: any stream ordered behind dP)
    const bool do_data = cfg->bwd_phase == 0 || (cfg->bwd_phase & IMMTSF_BWD_PHASE_A), do_seed = cfg->bwd_phase == 0 || (cfg->bwd_phase & IMMTSF_BWD_WGRAD_A);
    if (do_data && !(cfg->form & IMMTSF_FORM_LOWRANK_OUT)) {      // (LOWRANK_OUT: the producer's backward forms dZ from (dP, Wc) itself)
        if (rank_expand_ok(BT, d, x.PW, w.Wc, d, dZ, hf ? cfg->out_h : nullptr)) {       // dZ = dP Wc: many rows, rank PW (see rank_expand_kernel)
            CHECK(launch_rank_expand(dP, x.PW, w.Wc, d, dZ, hf ? cfg->out_h : nullptr, BT, d, x.PW, s));
        } else {
            GemmArgs g = gemm_args(BT, d, x.PW, x.PW, d, d);
            set_problem2(g, 0, dPm, mat(w.Wc, w.Wc16), mat(dZ, hf ? cfg->out_h : nullptr), nullptr);
            CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        }
    }
    if (do_seed) {   // dWc = dP^T Z,  dbc = column sums of dP
        GemmArgs g = gemm_args(x.PW, d, BT, x.PW, d, d);
        set_problem2(g, 0, dPm, Zm, mat(sc.dWc), nullptr, sc.dbc);
        g.c_prezeroed = 0;
        g.ws = sc.sk; g.ws_bytes = sc.skb;
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, g, s));
        if (hf) CHECK(launch_f32_to_bf16(sc.dWc, sc.dWc16, (size_t)x.PW * d, s));
    }
    return IMMTSF_OK;
}

int immtsf_mmf_xrank_lowrank_basis(const immtsf_fusion_cfg* cfg, void* workspace, size_t workspace_bytes, const float** basis, int32_t* rank) {
    if (!xr_supported(cfg) || !workspace || !basis || !rank) return IMMTSF_EINVAL;
    XPWs w = carve_xp(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    *basis = w.Wc;
    *rank = xr_dims(cfg).PW;
    return IMMTSF_OK;
}

/* the parameter-only step between ..._backward_data_z and immtsf_mmf_xrank_p_backward_params: dW_fold = dWc W_po^T + dbc b_po^T, db_fold
 * = dbc (left in `scratch`) and the producer's gradients dW_po = W_fold^T dWc, db_po = W_fold^T dbc -- any stream ordered behind the data
 * half (immtsf.train.FlagStep leaves it, with the parameter chain, to the branch that has time to spare) */
int immtsf_mmf_xrank_p_backward_pre_z(const immtsf_fusion_cfg* cfg, const float* proj_w, const float* proj_b, void* workspace,
                                      size_t workspace_bytes, void* scratch, size_t scratch_bytes, float* g_proj_w, float* g_proj_b,
                                      immtsf_stream_t stream) {
    if (!xr_supported(cfg) || !proj_w || !proj_b || !workspace || !scratch || !g_proj_w || !g_proj_b) return IMMTSF_EINVAL;
    XPWs w = carve_xp(cfg, workspace);
    XPScratch sc = carve_xp_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const XRDims x = xr_dims(cfg);
    const int d = x.d, prec = cfg->precision;
    const bool hf = xr_hf(cfg);
    Mat Wpo;
    CHECK(weight_mat(hf, proj_w, (size_t)d * d, w.Wpo16, s, &Wpo));
    {   // dW_fold = dWc W_po^T
        GemmArgs g = gemm_args(x.PW, d, d, d, d, d);
        set_problem2(g, 0, mat(sc.dWc, sc.dWc16), Wpo, mat(sc.dWf), nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    {   // dW_po = W_fold^T dWc  (a rank-PW product)
        GemmArgs g = gemm_args(d, d, x.PW, d, d, d);
        set_problem2(g, 0, mat(w.Wf, w.Wf16), mat(sc.dWc, sc.dWc16), mat(g_proj_w), nullptr);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, g, s));
    }
    {   // db_po = W_fold^T dbc;  dW_fold += dbc b_po^T;  db_fold = dbc
        VecJobList l;
        l.add(VJ_MVT, w.Wf, d, sc.dbc, nullptr, g_proj_b, x.PW, d);
        l.rank1(sc.dWf, d, x.PW, d, 1, sc.dbc, proj_b);
        VecJob& c = l.add(VJ_COPY, sc.dbc, x.PW, nullptr, nullptr, sc.dWfb, 1, x.PW);
        c.ldy = x.PW;
        CHECK(launch_vecjobs(l, s));
    }
    return IMMTSF_OK;
}

/* the data half of immtsf_mmf_xrank_p_backward: dE = dP W_fold and dW_fold = dP^T E (+ its column sums) into `scratch` */
int immtsf_mmf_xrank_p_backward_data(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt, const float* dP,
                                     float* dE_txt, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                     immtsf_stream_t stream) {
    if (!xr_supported(cfg) || !p || !E_txt || !dP || !dE_txt || !workspace || !scratch) return IMMTSF_EINVAL;
    XPWs w = carve_xp(cfg, workspace);
    XPScratch sc = carve_xp_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const XRDims x = xr_dims(cfg);
    const int d = x.d, BT = x.B * x.T, prec = cfg->precision;
    const bool hf = xr_hf(cfg);
    Mat dPm = cmat(dP), Em = cmat(E_txt, hf ? (cfg->aux_h ? cfg->aux_h : static_cast<const void*>(w.E16)) : nullptr);
    if (hf && !cfg->aux_h) CHECK(launch_f32_to_bf16(E_txt, w.E16, (size_t)BT * d, s));      // (the forward may have been given its own image)
    if (hf && cfg->in_h) {
        dPm.h = const_cast<void*>(cfg->in_h);
    } else if (hf) {
        CHECK(launch_f32_to_bf16(dP, sc.dP16, (size_t)BT * x.PW, s));
        dPm.h = sc.dP16;
    }
    if (rank_expand_ok(BT, d, x.PW, w.Wf, d, dE_txt, hf ? cfg->out_h : nullptr)) {  // dE = dP W_fold: many rows, rank PW (see rank_expand_kernel)
        CHECK(launch_rank_expand(dP, x.PW, w.Wf, d, dE_txt, hf ? cfg->out_h : nullptr, BT, d, x.PW, s));
    } else {
        GemmArgs g = gemm_args(BT, d, x.PW, x.PW, d, d);
        set_problem2(g, 0, dPm, mat(w.Wf, w.Wf16), mat(dE_txt, hf ? cfg->out_h : nullptr), nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    {   // dW_fold = dP^T E,  d b_fold = column sums of dP
        GemmArgs g = gemm_args(x.PW, d, BT, x.PW, d, d);
        set_problem2(g, 0, dPm, Em, mat(sc.dWf), nullptr, sc.dWfb);
        g.c_prezeroed = 0;
        g.ws = sc.sk; g.ws_bytes = sc.skb;
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, g, s));
    }
    return IMMTSF_OK;
}

/* the parameter half: the chain rule from dW_fold (in `scratch`, left there by ..._backward_data on a stream this call is ordered
 * behind) to the block's parameter gradients -- three dependent multi-job launches, of which this call runs [first, last) (0 <= first
 * <= last <= 3): parameter-only work that nothing but the optimizer waits for, so a caller may run its tail on another stream
 * (immtsf.train.FlagStep hands it to the backbone's branch, which finishes its backward earlier) */
int immtsf_mmf_xrank_p_backward_params(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* dbHO, void* workspace,
                                       size_t workspace_bytes, void* scratch, size_t scratch_bytes, const immtsf_xadd_params* gr,
                                       int32_t first, int32_t last, immtsf_stream_t stream) {
    if (!xr_supported(cfg) || !p || !gr || !dbHO || !workspace || !scratch || first < 0 || last > 3 || first > last) return IMMTSF_EINVAL;
    XPWs w = carve_xp(cfg, workspace);
    XPScratch sc = carve_xp_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const XRDims x = xr_dims(cfg);
    const int d = x.d, C = x.C, Cq = x.Cq, H = x.H, E = x.E, Wd = x.Wd;
    const float scale = sqrtf(1.0f / (float)E);
    const float *Wiq = p->attn_in_w, *Wik = p->attn_in_w + (size_t)d * d, *Wiv = p->attn_in_w + (size_t)2 * d * d;
    const float *bik = p->attn_in_b + d, *biv = p->attn_in_b + 2 * d;
    float *g_iq = gr->attn_in_w, *g_ik = gr->attn_in_w + (size_t)d * d, *g_iv = gr->attn_in_w + (size_t)2 * d * d;
    float *gb_iq = gr->attn_in_b, *gb_ik = gr->attn_in_b + d, *gb_iv = gr->attn_in_b + 2 * d;
    if (first <= 0 && last > 0) {   // T1_h = dG_h W_k^T, RW_h = dU_h W_v^T;  dW_k = scale sum_h GA_h^T dG_h, dW_v = sum_h UA_h^T dU_h;  d b_k, d b_v
        XJobList L;
        for (int h = 0; h < H; ++h) {
            const size_t r0 = (size_t)h * Wd;
            L.nt(sc.dWf + r0 * d, d, 1, Cq, p->proj_k_w, d, d, d, sc.T1 + (size_t)h * Cq * d, d, 1.f);
            L.nt(sc.dWf + (r0 + Cq) * d, d, 1, C, p->proj_v_w, d, d, d, sc.RW + (size_t)h * C * d, d, 1.f);
            const size_t o = (size_t)h * E;
            // d b_k[m] = scale sum_j AqT[j][m] d b_fold[(h, j)]: an M x 1 outer product
            L.outer(w.AqT + o, d, Cq, sc.dWfb + r0, 1, E, 1, gb_ik + o, 1, scale);
            L.outer(w.WHO + o, d, C, sc.dWfb + r0 + Cq, 1, E, 1, gb_iv + o, 1, 1.f);
        }
        XJob* a = L.outer(w.GA, d, Cq, sc.dWf, d, d, d, gr->proj_k_w, d, scale);
        a->ng = H; a->pgs = (long)Cq * d; a->qgs = (long)Wd * d;
        XJob* b = L.outer(w.UA, d, C, sc.dWf + (size_t)Cq * d, d, d, d, gr->proj_v_w, d, 1.f);
        b->ng = H; b->pgs = (long)C * d; b->qgs = (long)Wd * d;
        CHECK(L.launch(s, false));
    }
    if (first <= 1 && last > 1) {   // dAqT_h = scale (T1_h W_in,k[head]^T + d b_fold b_k^T), dW_HO,h = RW_h W_in,v[head]^T + d b_fold b_v^T;  dW_in,k, dW_in,v
        XJobList L;
        for (int h = 0; h < H; ++h) {
            const size_t r0 = (size_t)h * Wd, o = (size_t)h * E;
            L.nt(sc.T1 + (size_t)h * Cq * d, d, 1, Cq, Wik + o * d, d, d, E, sc.dAqT + o, d, scale, sc.dWfb + r0, bik + o);
            L.nt(sc.RW + (size_t)h * C * d, d, 1, C, Wiv + o * d, d, d, E, sc.dWHO + o, d, 1.f, sc.dWfb + r0 + Cq, biv + o);
            L.outer(w.AqT + o, d, Cq, sc.T1 + (size_t)h * Cq * d, d, E, d, g_ik + o * d, d, scale);
            L.outer(w.WHO + o, d, C, sc.RW + (size_t)h * C * d, d, E, d, g_iv + o * d, d, 1.f);
        }
        CHECK(L.launch(s, false));
    }
    if (first <= 2 && last > 2) {   // the query side and the output side: W_Qf = W_in,q W_q, W_HO = W_res W_out, b_HO = W_res b_out + b_res
        XJobList L;
        L.outer(sc.dAqT, d, C, p->proj_q_w, 1, d, d, g_iq, d, 1.f)->qsn = C;            // dW_in,q[m][n] = sum_c dAqT[c][m] W_q[n][c]
        L.copy(sc.dAqT + (size_t)C * d, d, gb_iq);                                       // d b_q = row C of dAqT
        L.nn(sc.dAqT, d, C, Wiq, d, d, d, gr->proj_q_w, 1, 1.f, nullptr, C);             // dW_q[k][c] = sum_m dAqT[c][m] W_in,q[m][k]
        L.outer(p->res_w, d, C, sc.dWHO, d, d, d, gr->attn_out_w, d, 1.f);               // dW_out = W_res^T dW_HO
        L.outer(p->res_w, d, C, dbHO, 1, d, 1, gr->attn_out_b, 1, 1.f);                  // d b_out = W_res^T d b_HO
        L.nt(sc.dWHO, d, 1, C, p->attn_out_w, d, d, d, gr->res_w, d, 1.f, dbHO, p->attn_out_b);     // dW_res = dW_HO W_out^T + d b_HO b_out^T
        L.copy(dbHO, C, gr->res_b);
        CHECK(L.launch(s, false));
    }
    return IMMTSF_OK;
}

/* dP (B*T, PW), dbHO (C) -> dE_txt (B*T, d) and the gradients of every parameter except LayerNorm's (all overwritten) */
int immtsf_mmf_xrank_p_backward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt, const float* dP,
                                const float* dbHO, float* dE_txt, void* workspace, size_t workspace_bytes, void* scratch,
                                size_t scratch_bytes, const immtsf_xadd_params* gr, immtsf_stream_t stream) {
    if (!gr || !dbHO) return IMMTSF_EINVAL;
    if (int rc = immtsf_mmf_xrank_p_backward_data(cfg, p, E_txt, dP, dE_txt, workspace, workspace_bytes, scratch, scratch_bytes, stream)) return rc;
    return immtsf_mmf_xrank_p_backward_params(cfg, p, dbHO, workspace, workspace_bytes, scratch, scratch_bytes, gr, 0, 3, stream);
}

int immtsf_mmf_xrank_q_forward(const immtsf_fusion_cfg* cfg, const float* ln_w, const float* ln_b, const float* Y_ts, const float* P,
                               const float* bHO, const uint8_t* M_txt, float* Y_out, void* workspace, size_t workspace_bytes,
                               immtsf_stream_t stream) {
    if (!xr_supported(cfg) || !ln_w || !ln_b || !Y_ts || !P || !bHO || !M_txt || !Y_out || !workspace) return IMMTSF_EINVAL;
    XQWs w = carve_xq(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    const XQDims q = xq_dims(cfg);
    const DropCfg drop = drop_of(cfg);
    const size_t lds = (size_t)q.wpb * q.T * q.PW * sizeof(float);
    const int grid = cdiv(q.B, q.wpb);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (q.C <= 8)
        hipLaunchKernelGGL(xrank_q_fwd_kernel<8>, dim3(grid), dim3(256), lds, s, q, Y_ts, P, bHO, M_txt, ln_w, ln_b, Y_out, w.lse, w.xhat, w.rstd,
                           w.ticket, drop);
    else
        hipLaunchKernelGGL(xrank_q_fwd_kernel<16>, dim3(grid), dim3(256), lds, s, q, Y_ts, P, bHO, M_txt, ln_w, ln_b, Y_out, w.lse, w.xhat, w.rstd,
                           w.ticket, drop);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

/* dY_out -> dY_ts (B*T, C), dP (B*T, PW) (+ its bf16 image in cfg->out_h when the bf16 dataflow is on), d b_HO (C), d ln_w, d ln_b */
int immtsf_mmf_xrank_q_backward(const immtsf_fusion_cfg* cfg, const float* ln_w, const float* Y_ts, const float* P, const uint8_t* M_txt,
                                const float* dY_out, float* dY_ts, float* dP, float* dbHO, float* d_ln_w, float* d_ln_b, void* workspace,
                                size_t workspace_bytes, immtsf_stream_t stream) {
    if (!xr_supported(cfg) || !ln_w || !Y_ts || !P || !M_txt || !dY_out || !dY_ts || !dP || !dbHO || !d_ln_w || !d_ln_b || !workspace)
        return IMMTSF_EINVAL;
    XQWs w = carve_xq(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    const XQDims q = xq_dims(cfg);
    const XRDims x = xr_dims(cfg);
    const DropCfg drop = drop_of(cfg);
    const size_t lds = xq_lds_floats(x) * q.wpb * sizeof(float);
    const int grid = cdiv(q.B, q.wpb);
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned short* dP16 = (xr_hf(cfg) && cfg->out_h) ? static_cast<unsigned short*>(cfg->out_h) : nullptr;
    if (q.C <= 8)
        hipLaunchKernelGGL(xrank_q_bwd_kernel<8>, dim3(grid), dim3(256), lds, s, q, Y_ts, P, M_txt, ln_w, w.xhat, w.rstd, w.lse, dY_out, dY_ts, dP,
                           dP16, w.slabs, w.ticket, d_ln_w, d_ln_b, dbHO, drop);
    else
        hipLaunchKernelGGL(xrank_q_bwd_kernel<16>, dim3(grid), dim3(256), lds, s, q, Y_ts, P, M_txt, ln_w, w.xhat, w.rstd, w.lse, dY_out, dY_ts, dP,
                           dP16, w.slabs, w.ticket, d_ln_w, d_ln_b, dbHO, drop);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

size_t immtsf_mmf_xrank_q_train_scratch_bytes(const immtsf_fusion_cfg* cfg) {
    if (!xr_supported(cfg)) return 0;
    const XRDims x = xr_dims(cfg);
    return (size_t)cdiv(x.B, xq_wpb(x)) * (3 * x.C + 1) * sizeof(float) + 256;
}

/* Training step of the Q half in one launch: forward, masked-MSE loss against `truth` under `mask` with the per-variable observation
 * counts `cnt` (C floats; the loss of immtsf_masked_mse_counted: mean over the variables with a non-zero count of sum err^2 / count),
 * and backward seeded with d loss = grad_scale.  Y_out may be NULL.  ticket: two zero-initialised device words that the call leaves
 * zero (calls sharing them must be ordered); scratch: q_train_scratch_bytes.  done_flag (may be NULL): set to 1 (release, device
 * scope) as soon as dY_ts is complete -- before the rest of the kernel has run -- for a consumer on another stream that waits with
 * immtsf_flag_wait. */
int immtsf_mmf_xrank_q_train(const immtsf_fusion_cfg* cfg, const float* ln_w, const float* ln_b, const float* Y_ts, const float* P,
                             const float* bHO, const uint8_t* M_txt, const float* truth, const float* mask, const float* cnt,
                             float grad_scale, float* Y_out, float* loss, float* dY_ts, float* dP, float* dbHO, float* d_ln_w,
                             float* d_ln_b, void* scratch, size_t scratch_bytes, uint32_t* ticket, int32_t* done_flag,
                             immtsf_stream_t stream) {
    if (!xr_supported(cfg) || !ln_w || !ln_b || !Y_ts || !P || !bHO || !M_txt || !truth || !mask || !cnt || !loss || !dY_ts || !dP || !dbHO ||
        !d_ln_w || !d_ln_b || !scratch || !ticket)
        return IMMTSF_EINVAL;
    if (scratch_bytes < immtsf_mmf_xrank_q_train_scratch_bytes(cfg)) return IMMTSF_EWORKSPACE;
    const XQDims q = xq_dims(cfg);
    const XRDims x = xr_dims(cfg);
    const DropCfg drop = drop_of(cfg);
    const size_t lds = xq_lds_floats(x) * q.wpb * sizeof(float);
    const int grid = cdiv(q.B, q.wpb);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* slabs = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(scratch) + 255) & ~uintptr_t(255));
    unsigned short* dP16 = (xr_hf(cfg) && cfg->out_h) ? static_cast<unsigned short*>(cfg->out_h) : nullptr;
    if (q.C <= 8)
        hipLaunchKernelGGL(xrank_q_train_kernel<8>, dim3(grid), dim3(256), lds, s, q, Y_ts, P, bHO, M_txt, ln_w, ln_b, truth, mask, cnt, grad_scale, Y_out,
                           dY_ts, dP, dP16, slabs, ticket, d_ln_w, d_ln_b, dbHO, loss, done_flag, drop);
    else
        hipLaunchKernelGGL(xrank_q_train_kernel<16>, dim3(grid), dim3(256), lds, s, q, Y_ts, P, bHO, M_txt, ln_w, ln_b, truth, mask, cnt, grad_scale, Y_out,
                           dY_ts, dP, dP16, slabs, ticket, d_ln_w, d_ln_b, dbHO, loss, done_flag, drop);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

}  // extern "C"
