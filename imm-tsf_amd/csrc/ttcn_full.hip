// The whole time-aware patch encoder of one patch on one CU (bf16 MFMA mode, ttcn_dim <= 32, 1 + te_dim <= 16, L <= 64):
// LearnableTE, the three filter-generator layers, the masked softmax over the patch's slots and the meta-filter pooling,
// forward in one kernel and backward in one kernel.  Reference: models/tPatchGNN.py:176-195.
//
// Why: the streaming formulation (ttcn.hip) runs the layers as GEMMs over all P*L slots and therefore moves the
// (P*L, F*K) filter tensor through HBM six times per step (46 MB a pass at the benchmark shape) plus 19 launches.  But a
// patch's working set is tiny -- X: L x F, h1/h2: L x 32, filter tile: L x F*K -- every contraction has K <= 32 (one
// MFMA k-step) or M, N <= 32, and the softmax runs over the tile's ROWS, which the 16x16 accumulator layout keeps in
// 4 lane groups x 4 registers (two xor-shuffles finish a column).  So a workgroup builds X from (x, t), runs the MLP
// through small bf16 LDS tiles, produces each 16-column slice of the filter tile with RT MFMAs, normalises and pools
// it in registers; nothing but the (P, K) result and the pooled contributions goes back to HBM.  The backward RECOMPUTES
// all of that, forms d(logits) in registers, parks it in LDS as bf16 and runs the six backward products as MFMAs on
// LDS tiles; weight / bias / time-embedding gradients are accumulated in registers across the patches of a persistent
// workgroup and added to HBM once per workgroup.
//
// Column order: filter column c = k*F + f in the reference; here c' = f*32 + k (W3 is packed that way), so a 16-column
// tile has ONE f: the pooling-path gradient of X (a sum over k) is a 16-lane shuffle reduction instead of 11k LDS
// atomics per patch (measured: 40 us of an 88 us kernel).
#include "ttcn.hpp"
#include "../../include/immtsf.h"
#include "common.hpp"
#include <stdio.h>
#include <stdlib.h>

int g_immtsf_ttcn_bwd_grid = 0;

namespace {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));       // element-wise arithmetic on these compiles to v_pk_{fma,mul,add}_f32

constexpr int KP = 32;        // padded ttcn_dim and padded feature width (one MFMA k-step)
constexpr int PT = 40;        // pitch (bf16 elements) of the 32-wide LDS tiles: 80-byte rows, conflict-free b128 reads


struct FD { int P, L, F, K, NCq, dbg; };

// slab of padded gradients (fp32), one per workgroup of the backward (summed by ttcn_unpack_kernel): offsets in floats
struct Slab { int W1, b1, W2, b2, W3, b3, te, Tb, total; };
__host__ __device__ inline Slab slab_of(int F) {
    Slab s;
    int o = 0;
    s.W1 = o; o += KP * KP;          // [k1][f]
    s.b1 = o; o += KP;
    s.W2 = o; o += KP * KP;          // [k2][k1]
    s.b2 = o; o += KP;
    s.W3 = o; o += F * 32 * KP;      // [c'][k2]
    s.b3 = o; o += F * 32;
    s.te = o; o += 2 * KP;           // [f] d/dw part, [32+f] d/db part
    s.Tb = o; o += KP;
    s.total = o;
    return s;
}

__device__ __forceinline__ bf16x8 load8_bf16(const float* __restrict__ src) {     // 8 consecutive fp32 -> bf16x8 (RNE)
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    bf16x8 r;
    r[0] = (bf16_t)a.x; r[1] = (bf16_t)a.y; r[2] = (bf16_t)a.z; r[3] = (bf16_t)a.w;
    r[4] = (bf16_t)b.x; r[5] = (bf16_t)b.y; r[6] = (bf16_t)b.z; r[7] = (bf16_t)b.w;
    return r;
}
__device__ __forceinline__ float col_sum(float v) { return xor32_sum(xor16_sum(v)); }      // over the 4 lane groups that hold one column's rows
__device__ __forceinline__ float col_max(float v) { return xor32_max(xor16_max(v)); }
__device__ __forceinline__ float row_sum16(float v) { return row16_sum(v); }                // over the 16 lanes (columns) of a lane group
// A / B fragment from a row-major bf16 LDS tile: row = row0 + (lane & 15), k = k0 + (lane >> 4) * 8 ..
__device__ __forceinline__ bf16x8 frag_row(const bf16_t* tile, int pitch, int row0, int k0, int fr, int fq) {
    return *reinterpret_cast<const bf16x8*>(tile + (row0 + fr) * pitch + k0 + fq * 8);
}
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
__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

struct TEp { const float *ws, *bs, *wp, *bp; };

// ---- per-patch inputs.  A persistent workgroup loads the NEXT patch's rows (t, x, mask; the backward: the pooled contributions
// and the masked output gradient as well) into registers at the top of a patch and parks them in the other half of a double-buffered LDS
// block just before the patch's last barrier: the global-load latency (1-2 us under load, paid twice per patch when the loads sat in
// front of their uses: a third of a patch's timeline) is hidden behind the patch's own work.
// block layout (floats): t [ROWS] | x [ROWS] | mask [ROWS] | (backward) dp [KP] | ctr [NCq]
template <int RT, bool BWD>
struct PatchIn {
    static constexpr int ROWS = RT * 16, HEAD = 3 * ROWS + (BWD ? KP : 0);
    float row, c[2];
    __device__ __forceinline__ void fetch(const FD& d, int p, const float* __restrict__ x, const float* __restrict__ tt,
                                          const float* __restrict__ mask, const float* __restrict__ ctr, const float* __restrict__ out,
                                          const float* __restrict__ dout, int out_ld) {
        const int tid = threadIdx.x;
        row = 0.f; c[0] = c[1] = 0.f;
        if (tid < 3 * ROWS) {
            const int which = tid / ROWS, l = tid % ROWS;
            if (l < d.L) row = (which == 0 ? tt : which == 1 ? x : mask)[(size_t)p * d.L + l];
        } else if (BWD && tid < HEAD) {
            const int k = tid - 3 * ROWS;
            if (k < d.K) row = out[(size_t)p * out_ld + k] > 0.f ? dout[(size_t)p * out_ld + k] : 0.f;
        }
        if (BWD) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = tid + 256 * j;
                if (i < d.NCq) c[j] = ctr[(size_t)p * d.NCq + i];
            }
        }
    }
    __device__ __forceinline__ void park(const FD& d, float* blk) const {
        const int tid = threadIdx.x;
        if (tid < HEAD) blk[tid] = row;
        if (BWD) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = tid + 256 * j;
                if (i < d.NCq) blk[HEAD + i] = c[j];
            }
        }
    }
};

// a thread's column of X never changes (f = tid & 31): its time-embedding parameters live in registers for the whole kernel
struct XCol { float w, b; int kind; };      // kind 0: the value column, 1: linear, 2: periodic, 3: padding
__device__ __forceinline__ XCol x_col(const FD& d, TEp te) {
    const int f = threadIdx.x & 31;
    XCol c{0.f, 0.f, 3};
    if (f == 0) c.kind = 0;
    else if (f == 1) { c.w = te.ws[0]; c.b = te.bs[0]; c.kind = 1; }
    else if (f < d.F) { c.w = te.wp[f - 2]; c.b = te.bp[f - 2]; c.kind = 2; }
    return c;
}

// X[l, 0] = x ; X[l, 1] = ws*t+bs ; X[l, 1+j] = sin(wp_j t + bp_j) ; zero padding.  Xb: bf16 [ROWS][PT]; Xf: fp32 [ROWS][16] --
// or, TR (backward): [16][ROWS + 4], rows contiguous: the accumulator layout gives a lane four consecutive rows of one column, so its
// share of a column of X (and of Xc, Xct) is ONE 16-byte read.  blk: the patch's staged rows (t | x | ...)
template <int RT> constexpr int xt_pitch() { return RT * 16 + 4; }
template <int RT, bool TR = false>
__device__ __forceinline__ void build_x(const FD& d, const XCol xc, const float* blk, bf16_t* Xb, float* Xf, float* Xc = nullptr) {
    // Xc (backward): d X[l, f] / d(its pre-activation): 0 for the value column, 1 for the linear one, cos(.) for the periodic ones
    const int f = threadIdx.x & 31;
#pragma unroll
    for (int j = 0; j < RT * 2; ++j) {
        const int l = (threadIdx.x >> 5) + 8 * j;
        float v = 0.f, c = 0.f, t = 0.f;
        if (l < d.L && xc.kind != 3) {
            t = blk[l];
            if (xc.kind == 0) v = blk[RT * 16 + l];
            else {
                const float ph = fmaf(xc.w, t, xc.b);
                v = xc.kind == 1 ? ph : __sinf(ph);            // (bf16 mode: v_sin_f32 / v_cos_f32)
                c = xc.kind == 1 ? 1.f : __cosf(ph);
            }
        }
        Xb[l * PT + f] = (bf16_t)v;
        if (f < 16) {
            const int o = TR ? f * xt_pitch<RT>() + l : l * 16 + f;
            Xf[o] = v;
            if (Xc) Xc[o] = c;
        }
    }
}

// one MLP layer on LDS tiles: out[l, n] = relu(sum_k in[l, k] W[n, k] + b[n]); output tiles (rt, nt) dealt to the waves
template <int RT>
__device__ __forceinline__ void mlp_layer(const bf16_t* in, bf16_t* outp, const float* __restrict__ W, const float* __restrict__ b,
                                          int wave, int fr, int fq) {
    for (int t = wave; t < RT * 2; t += 4) {
        const int rt = t >> 1, nt = t & 1;
        const f32x4 acc = mfma(frag_row(in, PT, rt * 16, 0, fr, fq), load8_bf16(W + (nt * 16 + fr) * KP + fq * 8), zero4());
        const float bias = b[nt * 16 + fr];
#pragma unroll
        for (int r = 0; r < 4; ++r) outp[(rt * 16 + fq * 4 + r) * PT + nt * 16 + fr] = (bf16_t)fmaxf(acc[r] + bias, 0.f);
    }
}

// the same with the wave's weight fragment / bias already in registers (tiles t = wave, wave + 4: always column tile wave & 1)
template <int RT>
__device__ __forceinline__ void mlp_layer_frag(const bf16_t* in, bf16_t* outp, const bf16x8 wf, const float bias, int wave, int fr, int fq) {
    for (int t = wave; t < RT * 2; t += 4) {
        const int rt = t >> 1, nt = t & 1;
        const f32x4 acc = mfma(frag_row(in, PT, rt * 16, 0, fr, fq), wf, zero4());
#pragma unroll
        for (int r = 0; r < 4; ++r) outp[(rt * 16 + fq * 4 + r) * PT + nt * 16 + fr] = (bf16_t)fmaxf(acc[r] + bias, 0.f);
    }
}

struct Wts { const float *W1p, *b1p, *W2p, *b2p, *W3q, *b3q; };     // packed fp32 weights (ttcn_pack_kernel)

// ---------------------------------------------------------------------------------------------------- forward
// persistent workgroups (grid <= 2048), 256 threads.  LDS: Xb | h1s | h2s (bf16 [ROWS][PT]) | Xf fp32 [ROWS][16] | ctr fp32 [NCq] | staged rows fp32 [2][3 ROWS]
// MF = ceil(F / 4): the f's a wave owns.  Their layer-3 fragments (bf16 image W3h of the pack kernel), biases and the rows' mask
// values are loaded FIRST, beside the staging loads: in front of their first use they were three dependent round trips (one per
// f of the wave) on a kernel whose whole timeline is ~10 us.
template <int RT, int MF>
__global__ __launch_bounds__(256) void ttcn_full_fwd_kernel(FD d, const float* __restrict__ x, const float* __restrict__ tt,
                                                             const float* __restrict__ mask, TEp te, Wts w, const bf16_t* __restrict__ W3h,
                                                             const float* __restrict__ Tb, float* __restrict__ ctr,
                                                             float* __restrict__ out, int out_ld, int flag_col) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWS = RT * 16, XP = xt_pitch<RT>();
    constexpr float LOG2E = 1.4426950408889634f;
    bf16_t* Xb = reinterpret_cast<bf16_t*>(smem);
    bf16_t* h1s = Xb + ROWS * PT;
    bf16_t* h2s = h1s + ROWS * PT;
    float* Xf = reinterpret_cast<float*>(h2s + ROWS * PT);      // [16][XP]: a lane's four rows of a column are one 16-byte read
    float* cl = Xf + 16 * XP;
    float* pin = cl + d.NCq;                      // [2][3 * ROWS]: the patch's staged rows, double-buffered (PatchIn)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    // persistent workgroups (grid <= P): the layer-3 fragments of this wave's f slots, the MLP's weight fragments of its column tile and
    // the biases are loaded ONCE; a workgroup per patch re-fetched them (L2 round trips in front of every phase) 65 536 times at
    // 4096 windows
    bf16x8 bw[MF][2];
    float b3v[MF][2];               // (log2 units)
    const int tnt = wave & 1;
    typedef PatchIn<RT, false> PI;
    PI nx;
    nx.fetch(d, blockIdx.x, x, tt, mask, nullptr, nullptr, nullptr, 0);
    const XCol xc = x_col(d, te);
    const bf16x8 mw1 = load8_bf16(w.W1p + (tnt * 16 + fr) * KP + fq * 8), mw2 = load8_bf16(w.W2p + (tnt * 16 + fr) * KP + fq * 8);
    const float mb1 = w.b1p[tnt * 16 + fr], mb2 = w.b2p[tnt * 16 + fr];
    const float tb = tid < d.K ? Tb[tid] : 0.f;
#pragma unroll
    for (int j = 0; j < MF; ++j)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int f = wave + 4 * j, c = f * 32 + half * 16 + fr;
            bw[j][half] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            b3v[j][half] = 0.f;
            if (f < d.F) { bw[j][half] = *reinterpret_cast<const bf16x8*>(W3h + (size_t)c * KP + fq * 8); b3v[j][half] = w.b3q[c] * LOG2E; }
        }
    nx.park(d, pin);
    __syncthreads();
    int it = 0;
    for (int p = blockIdx.x; p < d.P; p += gridDim.x, ++it) {
    const float* cur = pin + (it & 1) * PI::HEAD;
    const int pn = p + gridDim.x;
    if (pn < d.P) nx.fetch(d, pn, x, tt, mask, nullptr, nullptr, nullptr, 0);
    build_x<RT, true>(d, xc, cur, Xb, Xf);
    __syncthreads();
    mlp_layer_frag<RT>(Xb, h1s, mw1, mb1, wave, fr, fq);
    __syncthreads();
    mlp_layer_frag<RT>(h1s, h2s, mw2, mb2, wave, fr, fq);
    __syncthreads();
    bf16x8 a[RT];
    // the lane's rows: live (mask != 0), or the value the reference puts in their place: Filter * mask + (1 - mask) * (-1e8) for a
    // masked slot (mask is 0 / 1, so a select is the same value), -inf past L -- in log2 units
    bool on[RT][4];
    float cval[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        a[rt] = frag_row(h2s, PT, rt * 16, 0, fr, fq);
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(cur + 2 * ROWS + rt * 16 + fq * 4);     // (0 past L)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            on[rt][r] = m4[r] != 0.f;
            cval[rt][r] = rt * 16 + fq * 4 + r < d.L ? -1e8f * LOG2E : -INFINITY;
        }
    }
#pragma unroll
    for (int j = 0; j < MF; ++j) {
        const int f = wave + 4 * j;
        if (f < d.F) {
            f32x2 xf[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const f32x4 x4 = *reinterpret_cast<const f32x4*>(Xf + f * XP + rt * 16 + fq * 4);
                xf[rt][0] = f32x2{x4[0], x4[1]};
                xf[rt][1] = f32x2{x4[2], x4[3]};
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c = f * 32 + half * 16 + fr;
                // one 16-column tile of the filter logits -> masked softmax over the patch's rows and the pooling, on PAIRS of rows
                // (v_pk_*_f32); the softmax's 1 / sum is applied to the pooled value, not to the weights
                const float bias2 = b3v[j][half];
                f32x2 v[RT][2];
                float m = -INFINITY;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const f32x4 acc = mfma(a[rt], bw[j][half], zero4());
                    v[rt][0] = f32x2{acc[0], acc[1]} * LOG2E + bias2;
                    v[rt][1] = f32x2{acc[2], acc[3]} * LOG2E + bias2;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            v[rt][h][e] = on[rt][2 * h + e] ? v[rt][h][e] : cval[rt][2 * h + e];
                            m = fmaxf(m, v[rt][h][e]);
                        }
                }
                m = col_max(m);
                f32x2 s2{0.f, 0.f}, p2{0.f, 0.f};
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        v[rt][h] -= m;
                        v[rt][h][0] = __builtin_amdgcn_exp2f(v[rt][h][0]);   // exp2(-inf) = 0 for the padded rows; bf16 mode: v_exp_f32 is ample
                        v[rt][h][1] = __builtin_amdgcn_exp2f(v[rt][h][1]);
                        s2 += v[rt][h];
                        p2 += v[rt][h] * xf[rt][h];
                    }
                const float acc = col_sum(p2[0] + p2[1]) * __builtin_amdgcn_rcpf(col_sum(s2[0] + s2[1]));    // (v_rcp_f32: 1 ulp)
                if (fq == 0) {
                    cl[c] = acc;
                    ctr[(size_t)p * d.NCq + c] = acc;
                }
            }
        }
    }
    if (pn < d.P) nx.park(d, pin + ((it + 1) & 1) * PI::HEAD);     // (that half was last read in the previous patch)
    __syncthreads();
    if (tid < d.K) {
        float s = tb;
        for (int f = 0; f < d.F; ++f) s += cl[f * 32 + tid];
        out[(size_t)p * out_ld + tid] = fmaxf(s, 0.f);
    }
    if (flag_col >= 0 && wave == 1) {
        float any = lane < d.L ? cur[2 * ROWS + lane] : 0.f;
        any = wave_sum(any);
        if (lane == 0) out[(size_t)p * out_ld + flag_col] = any > 0.f ? 1.f : 0.f;
    }
    __syncthreads();        // the next patch rewrites every LDS tile (and cl)
    }   // patches
}

// ---------------------------------------------------------------------------------------------------- backward
// transposed bf16 copies of the weights for the data-gradient products (ttcn_pack_kernel)
struct WtsT { const bf16_t *W3T /*[k2][c']*/, *W2T /*[k1][k2]*/, *W1T /*[f][k1]*/, *W3h /*[c'][k2]*/; };

// persistent workgroups, 256 threads.  LDS: Xb | h1s | h2s (later dz1) | dz2s (bf16 [ROWS][PT]) | dST bf16 [NCq][ROWS + 8] (d(logits),
// TRANSPOSED: a lane's four rows of a column are one 8-byte store) | XfT, XcT fp32 [16][ROWS + 4] |
// staged patch inputs fp32 [2][3 ROWS + 32 + NCq] (PatchIn) | W3s bf16 [NCq][PT] | W1s, W2s | b3s
// MF = ceil(F / 4): the f slots a wave owns (its layer-3 fragments and dW3 tiles live in registers for the whole kernel)
//
// Issue-bound kernel (3 000 instructions per patch and wave when first measured, 40 % of them the element-wise d(logits) phase): that
// phase works on PAIRS of rows (v_pk_fma / mul / add_f32: the accumulator layout hands a lane rows 4 fq .. 4 fq + 3 of a column, two
// pairs), the bias gradients are summed from the accumulators as they are produced (they were two 32-step serial LDS loops on one wave:
// a quarter of the kernel), and no operand of the patch loop comes from global memory.
template <int RT, int MF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT == 2 ? 2 : 1))) void ttcn_full_bwd_kernel(FD d, const float* __restrict__ x, const float* __restrict__ tt,
                                                             const float* __restrict__ mask, TEp te, Wts w, WtsT wt,
                                                             const float* __restrict__ ctr, const float* __restrict__ out,
                                                             const float* __restrict__ dout, int out_ld, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWS = RT * 16, PTT = ROWS + 8, XP = xt_pitch<RT>();
    constexpr float LOG2E = 1.4426950408889634f;
    bf16_t* Xb = reinterpret_cast<bf16_t*>(smem);
    bf16_t* h1s = Xb + ROWS * PT;
    bf16_t* h2s = h1s + ROWS * PT;
    bf16_t* dz2s = h2s + ROWS * PT;
    bf16_t* dz1s = h2s;                                     // (h2 is dead once dz2 and dW3 are done: a barrier earlier)
    bf16_t* dST = dz2s + ROWS * PT;                         // [NCq][PTT]
    float* Xf = reinterpret_cast<float*>(dST + d.NCq * PTT);
    float* Xc = Xf + 16 * XP;
    typedef PatchIn<RT, true> PI;
    const int PIN = PI::HEAD + d.NCq;                       // floats of one staged patch: t | x | mask | dp | ctr
    float* pin = Xc + 16 * XP;                              // [2][PIN]
    bf16_t* W3s = reinterpret_cast<bf16_t*>(pin + 2 * PIN); // [NCq][PT]: layer-3 weights, rows c' (k2 contiguous) -- see below
    bf16_t* W1s = W3s + d.NCq * PT;                         // [k1][PT] (f contiguous), [k2][PT] (k1 contiguous): the MLP's weights; their
    bf16_t* W2s = W1s + KP * PT;                            // transposes (data-gradient products) are read through the hardware transpose
    float* b3s = reinterpret_cast<float*>(W2s + KP * PT);   // [NCq] layer-3 bias, in log2 units (x log2 e)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const Slab sl = slab_of(d.F);
    PI nx;
    nx.fetch(d, blockIdx.x, x, tt, mask, ctr, out, dout, out_ld);
    const XCol xc = x_col(d, te);
    // The layer-3 weights are read twice per patch (logits: rows c' as B fragments; dz2 = dS W3: the same image through the
    // hardware transpose) -- 17 dependent L2 round trips per patch per wave when fetched from global memory, a fifth of the
    // patch's timeline.  A persistent workgroup stages them once.
    for (int i = tid; i < d.NCq * 4; i += 256) {
        const int c = i >> 2, q = i & 3;
        *reinterpret_cast<bf16x8*>(W3s + c * PT + q * 8) = *reinterpret_cast<const bf16x8*>(wt.W3h + (size_t)c * KP + q * 8);
    }
    for (int i = tid; i < d.NCq; i += 256) b3s[i] = w.b3q[i] * LOG2E;
    if (tid < 2 * KP * 4) {
        const int which = tid >> 7, r = (tid >> 2) & 31, q = tid & 3;
        *reinterpret_cast<bf16x8*>((which ? W2s : W1s) + r * PT + q * 8) = load8_bf16((which ? w.W2p : w.W1p) + r * KP + q * 8);
    }

    // ---- operands that never change: registers for the whole kernel
    f32x4 accW3[MF][2][2];        // dW3 tiles: [f slot][half][k2 tile]
    f32x2 accB3[MF][2];           // db3: this lane's rows only, as two running pairs (summed once, after the last patch)
#pragma unroll
    for (int j = 0; j < MF; ++j)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            accW3[j][half][0] = zero4(); accW3[j][half][1] = zero4();
            accB3[j][half] = f32x2{0.f, 0.f};
        }
    // per-wave tile roles of the 32x32 products: (tmt, tnt)
    const int tnt = wave & 1, tmt = wave >> 1;
    // the MLP's biases of this wave's output column tile (tnt)
    const float mb1 = w.b1p[tnt * 16 + fr], mb2 = w.b2p[tnt * 16 + fr];
    f32x4 accW2 = zero4(), accW1 = zero4();
    float sb2 = 0.f, sb1 = 0.f;     // db2 / db1 of column tnt * 16 + fr: this lane's rows, straight from the accumulators of dz2 / dz1
    float accT = 0.f;               // 128 <= tid < 160: dT_bias[tid - 128]
    float te_w = 0.f, te_b = 0.f;   // time-embedding gradients of column f = fr through the MLP (waves that own a dXm tile)
    f32x2 tpw[MF], tpb[MF];         // ... and through the pooling, of this wave's f slots: per-lane partial sums over the lane's (row, k)
#pragma unroll
    for (int j = 0; j < MF; ++j) tpw[j] = tpb[j] = f32x2{0.f, 0.f};

    nx.park(d, pin);
    __syncthreads();
    int it = 0;
    for (int p = blockIdx.x; p < d.P; p += gridDim.x, ++it) {
        // ---- stage: this patch's rows are in LDS already; the next patch's loads start here
        const float* cur = pin + (it & 1) * PIN;
        const float *dp = cur + 3 * ROWS, *cts = cur + PI::HEAD;
        const int pn = p + gridDim.x;
        if (pn < d.P) nx.fetch(d, pn, x, tt, mask, ctr, out, dout, out_ld);
        build_x<RT, true>(d, xc, cur, Xb, Xf, Xc);
        __syncthreads();
        if (tid >= 128 && tid < 128 + KP) accT += dp[tid - 128];
        mlp_layer_frag<RT>(Xb, h1s, frag_row(W1s, PT, tnt * 16, 0, fr, fq), mb1, wave, fr, fq);
        __syncthreads();
        mlp_layer_frag<RT>(h1s, h2s, frag_row(W2s, PT, tnt * 16, 0, fr, fq), mb2, wave, fr, fq);
        __syncthreads();
        bf16x8 a[RT];
        // the lane's rows: live (mask != 0), or the value the reference puts in their place: Filter * mask + (1 - mask) * (-1e8) for a
        // masked slot (mask is 0 / 1, so a select is the same value), -inf past L.  `any`: the patch has a live row at all -- the
        // mask factor of d(logits) only matters when it has none (a masked row's weight is exactly 0 next to a live one)
        bool on[RT][4];
        float cval[RT][4], anyf = 0.f;
        f32x4 t4[RT];            // the rows' times (0 past L)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            a[rt] = frag_row(h2s, PT, rt * 16, 0, fr, fq);
            t4[rt] = *reinterpret_cast<const f32x4*>(cur + rt * 16 + fq * 4);
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(cur + 2 * ROWS + rt * 16 + fq * 4);     // (0 past L)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                on[rt][r] = m4[r] != 0.f;
                cval[rt][r] = rt * 16 + fq * 4 + r < d.L ? -1e8f * LOG2E : -INFINITY;
                anyf += on[rt][r] ? 1.f : 0.f;
            }
        }
        anyf = col_sum(anyf) > 0.f ? 1.f : 0.f;
        // ---- d(logits), tile by tile in registers -> LDS (bf16, transposed); pooling-path dX; db3
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int f = wave + 4 * j;
            if (f < d.F) {
                // xfa: this f's column of X for the lane's rows (x any); dxs: this lane's share of the pooling path's
                // dX[l, f] = sum_k sm dp.  That gradient only feeds the time-embedding parameters, which are linear in it: the lane
                // adds dxs * (cos, cos * t) to per-lane sums that are reduced across lanes once, after the last patch -- no
                // per-row 16-lane reductions, no dX tile
                f32x2 xfa[RT][2], dxs[RT][2];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const f32x4 x4 = *reinterpret_cast<const f32x4*>(Xf + f * XP + rt * 16 + fq * 4);
                    xfa[rt][0] = f32x2{x4[0], x4[1]} * anyf;
                    xfa[rt][1] = f32x2{x4[2], x4[3]} * anyf;
                    dxs[rt][0] = dxs[rt][1] = f32x2{0.f, 0.f};
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int k = half * 16 + fr, c = f * 32 + k;
                    const bf16x8 b = frag_row(W3s, PT, f * 32 + half * 16, 0, fr, fq);
                    const float bias2 = b3s[c];
                    f32x2 v[RT][2];
                    float m = -INFINITY;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const f32x4 acc = mfma(a[rt], b, zero4());
                        v[rt][0] = f32x2{acc[0], acc[1]} * LOG2E + bias2;       // log2 units: one fma per pair, no multiply inside exp
                        v[rt][1] = f32x2{acc[2], acc[3]} * LOG2E + bias2;
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                v[rt][h][e] = on[rt][2 * h + e] ? v[rt][h][e] : cval[rt][2 * h + e];
                                m = fmaxf(m, v[rt][h][e]);
                            }
                    }
                    m = col_max(m);
                    f32x2 s2{0.f, 0.f};
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            v[rt][h] -= m;
                            v[rt][h][0] = __builtin_amdgcn_exp2f(v[rt][h][0]);   // exp2(-inf) = 0 for the padded rows; bf16 mode: v_exp_f32 is ample
                            v[rt][h][1] = __builtin_amdgcn_exp2f(v[rt][h][1]);
                            s2 += v[rt][h];
                        }
                    const float inv = __builtin_amdgcn_rcpf(col_sum(s2[0] + s2[1]));    // (v_rcp_f32: 1 ulp)
                    const float dpk = dp[k] * inv, cta = cts[c] * anyf;   // (dp is zero past K; the softmax's 1 / sum rides on it)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        bf16x4 o;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const f32x2 smd = v[rt][h] * dpk;                    // 0 past L (sm) and past K (dpk)
                            const f32x2 ds = smd * (xfa[rt][h] - cta);           // (x the mask: see `any`)
                            dxs[rt][h] += smd;
                            accB3[j][half] += ds;
                            o[2 * h] = (bf16_t)ds[0];
                            o[2 * h + 1] = (bf16_t)ds[1];
                        }
                        *reinterpret_cast<bf16x4*>(dST + c * PTT + rt * 16 + fq * 4) = o;
                    }
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(Xc + f * XP + rt * 16 + fq * 4);
                    const f32x2 g0 = dxs[rt][0] * f32x2{c4[0], c4[1]}, g1 = dxs[rt][1] * f32x2{c4[2], c4[3]};
                    tpb[j] += g0 + g1;
                    tpw[j] += g0 * f32x2{t4[rt][0], t4[rt][1]} + g1 * f32x2{t4[rt][2], t4[rt][3]};
                }
            }
        }
        __syncthreads();
        // ---- dz2 = (dS W3) * [h2 > 0] -> LDS tile (+ db2 from the accumulators); dW3 += dS^T h2
        for (int t = wave; t < RT * 2; t += 4) {
            const int rt = t >> 1;      // (column tile t & 1 = wave & 1 = tnt: t advances by 4)
            f32x4 acc = zero4();
#pragma unroll 4
            for (int kk = 0; kk < d.NCq; kk += 32)
                acc = mfma(frag_kmajor(dST, PTT, rt * 16, kk, fr, fq), frag_kmajor(W3s, PT, tnt * 16, kk, fr, fq), acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (rt * 16 + fq * 4 + r) * PT + tnt * 16 + fr;
                const float g = (float)h2s[o] > 0.f ? acc[r] : 0.f;
                dz2s[o] = (bf16_t)g;
                sb2 += g;
            }
        }
#pragma unroll
        for (int j = 0; j < MF; ++j) {
            const int f = wave + 4 * j;
            if (f < d.F) {
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int kk = 0; kk < ROWS; kk += 32) {
                        const bf16x8 af = frag_row(dST, PTT, f * 32 + half * 16, kk, fr, fq);
                        accW3[j][half][0] = mfma(af, frag_kmajor(h2s, PT, 0, kk, fr, fq), accW3[j][half][0]);
                        accW3[j][half][1] = mfma(af, frag_kmajor(h2s, PT, 16, kk, fr, fq), accW3[j][half][1]);
                    }
            }
        }
        __syncthreads();
        // ---- layer 2 backward: dW2 += dz2^T h1 (tile (tmt, tnt)); dz1 = (dz2 W2) * [h1 > 0] (+ db1 from the accumulators)
#pragma unroll
        for (int kk = 0; kk < ROWS; kk += 32)
            accW2 = mfma(frag_kmajor(dz2s, PT, tmt * 16, kk, fr, fq), frag_kmajor(h1s, PT, tnt * 16, kk, fr, fq), accW2);
        for (int t = wave; t < RT * 2; t += 4) {
            const int rt = t >> 1;      // nt = t & 1 = wave & 1 = tnt (t advances by 4)
            const f32x4 acc = mfma(frag_row(dz2s, PT, rt * 16, 0, fr, fq), frag_kmajor(W2s, PT, tnt * 16, 0, fr, fq), zero4());     // dh1 = dz2 W2, k1 tile tnt
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (rt * 16 + fq * 4 + r) * PT + tnt * 16 + fr;
                const float g = (float)h1s[o] > 0.f ? acc[r] : 0.f;
                dz1s[o] = (bf16_t)g;
                sb1 += g;
            }
        }
        __syncthreads();
        // ---- layer 1 backward: dW1 += dz1^T X (f tile 0 only: F <= 16); dX = dz1 W1 -> time-embedding grads (the pooling path's share: above)
        if (wave < 2) {
#pragma unroll
            for (int kk = 0; kk < ROWS; kk += 32)
                accW1 = mfma(frag_kmajor(dz1s, PT, wave * 16, kk, fr, fq), frag_kmajor(Xb, PT, 0, kk, fr, fq), accW1);
        }
        for (int rt = (RT == 2 ? wave - 2 : wave); rt >= 0 && rt < RT; rt += 4) {      // RT = 2: waves 2, 3; RT = 4: all four
            const f32x4 acc = mfma(frag_row(dz1s, PT, rt * 16, 0, fr, fq), frag_kmajor(W1s, PT, 0, 0, fr, fq), zero4());            // dXm = dz1 W1, f tile 0
            // d X / d(pre-activation) is in LDS (Xc: zero for the value column, the padding and rows past L)
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(Xc + fr * XP + rt * 16 + fq * 4);
            const f32x4 tr = *reinterpret_cast<const f32x4*>(cur + rt * 16 + fq * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g = acc[r] * c4[r];
                te_w = fmaf(g, tr[r], te_w);
                te_b += g;
            }
        }
        if (pn < d.P) nx.park(d, pin + ((it + 1) & 1) * PIN);     // (that half was last read in the previous patch)
        __syncthreads();     // every LDS tile is rewritten by the next patch
    }

    // ---- this workgroup's sums -> its own slab (plain stores; ttcn_unpack_kernel adds the slabs up).  As atomics into one slab
    // the ~11 k adds of each workgroup queued up per address: 67 us of an 88 us kernel at P = 1024 / 384 workgroups.
    float* my = slab + (size_t)blockIdx.x * sl.total;
#pragma unroll
    for (int j = 0; j < MF; ++j) {
        const int f = wave + 4 * j;
        if (f < d.F) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        my[sl.W3 + (size_t)(f * 32 + half * 16 + fq * 4 + r) * KP + nt * 16 + fr] = accW3[j][half][nt][r];
                const float b3sum = col_sum(accB3[j][half][0] + accB3[j][half][1]);
                if (fq == 0) my[sl.b3 + f * 32 + half * 16 + fr] = b3sum;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        my[sl.W2 + (tmt * 16 + fq * 4 + r) * KP + tnt * 16 + fr] = accW2[r];
        if (wave < 2) my[sl.W1 + (wave * 16 + fq * 4 + r) * KP + fr] = accW1[r];
    }
    if (tid >= 128 && tid < 128 + KP) my[sl.Tb + tid - 128] = accT;
    // bias and time-embedding gradients: several waves hold shares of the same entry -> summed in LDS first
    float* tes = pin;                 // [te_w 32 | te_b 32 | db2 32 | db1 32] (every tile of the last patch is dead: the loop ends with a barrier)
    if (tid < 4 * KP) tes[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MF; ++j) {
        const int f = wave + 4 * j;
        const float sw = col_sum(row_sum16(tpw[j][0] + tpw[j][1])), sb = col_sum(row_sum16(tpb[j][0] + tpb[j][1]));
        if (lane == 0 && f >= 1 && f < d.F) {
            atomicAdd(tes + f, sw);
            atomicAdd(tes + KP + f, sb);
        }
    }
    te_w = col_sum(te_w);
    te_b = col_sum(te_b);
    sb2 = col_sum(sb2);
    sb1 = col_sum(sb1);
    if (fq == 0) {
        if (RT == 4 || wave >= 2) {
            atomicAdd(tes + fr, te_w);
            atomicAdd(tes + KP + fr, te_b);
        }
        atomicAdd(tes + 2 * KP + tnt * 16 + fr, sb2);
        atomicAdd(tes + 3 * KP + tnt * 16 + fr, sb1);
    }
    __syncthreads();
    if (tid < 2 * KP) my[sl.te + tid] = tes[tid];
    else if (tid < 3 * KP) my[sl.b2 + tid - 2 * KP] = tes[tid];
    else if (tid < 4 * KP) my[sl.b1 + tid - 3 * KP] = tes[tid];
}

// ---- packing: padded fp32 weights (+ the f-major W3), transposed bf16 copies; unpacking of the gradient slab
struct PackIn { const float *W1, *b1, *W2, *b2, *W3, *b3; };
__global__ __launch_bounds__(256) void ttcn_pack_kernel(int F, int K, PackIn q, float* W1p, float* b1p, float* W2p, float* b2p,
                                                         float* W3q, float* b3q, bf16_t* W3T, bf16_t* W2T, bf16_t* W1T, bf16_t* W3h) {
    const int i = blockIdx.x * 256 + threadIdx.x, NCq = F * 32;
    if (i < KP * KP) {
        const int r = i >> 5, c = i & 31;
        const float v1 = (r < K && c < F) ? q.W1[r * F + c] : 0.f, v2 = (r < K && c < K) ? q.W2[r * K + c] : 0.f;
        W1p[i] = v1;                      // [k1][f]
        W2p[i] = v2;                      // [k2][k1]
        W1T[c * KP + r] = (bf16_t)v1;     // [f][k1]
        W2T[c * KP + r] = (bf16_t)v2;     // [k1][k2]
    }
    if (i < KP) { b1p[i] = i < K ? q.b1[i] : 0.f; b2p[i] = i < K ? q.b2[i] : 0.f; }
    if (i < NCq * KP) {
        const int c = i >> 5, kk = i & 31, f = c >> 5, k = c & 31;
        const float v = (k < K && kk < K) ? q.W3[(size_t)(k * F + f) * K + kk] : 0.f;
        W3q[i] = v;
        W3T[(size_t)kk * NCq + c] = (bf16_t)v;
        W3h[i] = (bf16_t)v;
    }
    if (i < NCq) { const int f = i >> 5, k = i & 31; b3q[i] = k < K ? q.b3[k * F + f] : 0.f; }
}

struct UnpackOut { float *W1, *b1, *W2, *b2, *W3, *b3, *ws, *bs, *wp, *bp, *Tb; };
// grid ceil(outputs / 32), 256 threads = 32 outputs x 8 slab lanes: out = sum over the nslab per-workgroup slabs of the backward
__global__ __launch_bounds__(256) void ttcn_unpack_kernel(int F, int K, const float* __restrict__ slab, int nslab, UnpackOut g, int te_acc) {
    __shared__ float red[8][33];
    const Slab sl = slab_of(F);
    const int slot = threadIdx.x & 31, part = threadIdx.x >> 5;
    int i = blockIdx.x * 32 + slot, n;
    float* dst = nullptr;
    int off = 0;
    bool acc = false;      // te_acc: the time-embedding parameters are shared with the decoder's LearnableTE, whose backward adds into the same buffers
    if (i < (n = K * F)) { dst = g.W1 + i; off = sl.W1 + (i / F) * KP + i % F; }
    else if ((i -= n) < (n = K * K)) { dst = g.W2 + i; off = sl.W2 + (i / K) * KP + i % K; }
    else if ((i -= n) < (n = F * K * K)) { const int c = i / K, kk = i % K, k = c / F, f = c % F; dst = g.W3 + i; off = sl.W3 + (f * 32 + k) * KP + kk; }
    else if ((i -= n) < (n = F * K)) { const int k = i / F, f = i % F; dst = g.b3 + i; off = sl.b3 + f * 32 + k; }
    else if ((i -= n) < K) { dst = g.b1 + i; off = sl.b1 + i; }
    else if ((i -= K) < K) { dst = g.b2 + i; off = sl.b2 + i; }
    else if ((i -= K) < K) { dst = g.Tb + i; off = sl.Tb + i; }
    else if ((i -= K) < 1) { dst = g.ws; off = sl.te + 1; acc = te_acc != 0; }
    else if ((i -= 1) < 1) { dst = g.bs; off = sl.te + KP + 1; acc = te_acc != 0; }
    else if ((i -= 1) < F - 2) { dst = g.wp + i; off = sl.te + 2 + i; acc = te_acc != 0; }
    else if ((i -= F - 2) < F - 2) { dst = g.bp + i; off = sl.te + KP + 2 + i; acc = te_acc != 0; }
    float s = 0.f;
    if (dst) {
#pragma unroll 8
        for (int w = part; w < nslab; w += 8) s += slab[(size_t)w * sl.total + off];
    }
    red[part][slot] = s;
    __syncthreads();
    if (part == 0 && dst) {
#pragma unroll
        for (int q = 1; q < 8; ++q) s += red[q][slot];
        *dst = (acc ? *dst : 0.f) + s;
    }
}
inline int ttcn_unpack_outputs(int F, int K) { return K * F + K * K + F * K * K + F * K + 3 * K + 2 + 2 * (F - 2); }
constexpr int kMaxBwdGrid = 512;       // slabs in the caller's scratch

size_t fwd_lds(int RT, int NCq) { return (size_t)RT * 16 * PT * 2 * 3 + (size_t)16 * (RT * 16 + 4) * 4 + (size_t)NCq * 4 + (size_t)2 * 3 * RT * 16 * 4 + 64; }
size_t bwd_lds(int RT, int NCq) {
    const size_t ROWS = RT * 16;
    return ROWS * PT * 2 * 4 + (size_t)NCq * (ROWS + 8) * 2 + 16 * (ROWS + 4) * 4 * 2 + 2 * (3 * ROWS + KP + (size_t)NCq) * 4 + (size_t)(NCq + 2 * KP) * PT * 2 + (size_t)NCq * 4;
}

struct PackPtrs { float *W1p, *b1p, *W2p, *b2p, *W3q, *b3q; bf16_t *W3T, *W2T, *W1T, *W3h; };
PackPtrs pack_ptrs(float* base, int F) {
    PackPtrs q;
    float* o = base;
    q.W1p = o; o += KP * KP;
    q.W2p = o; o += KP * KP;
    q.b1p = o; o += KP;
    q.b2p = o; o += KP;
    q.W3q = o; o += (size_t)F * 32 * KP;
    q.b3q = o; o += F * 32;
    q.W3T = reinterpret_cast<bf16_t*>(o); o += (size_t)F * 32 * KP / 2;      // every block above is a multiple of 32 floats
    q.W2T = reinterpret_cast<bf16_t*>(o); o += KP * KP / 2;
    q.W1T = reinterpret_cast<bf16_t*>(o); o += KP * KP / 2;
    q.W3h = reinterpret_cast<bf16_t*>(o);
    return q;
}

}  // namespace

bool ttcn_full_supported(int precision, int L, int F, int K) {
    return precision == 1 && K >= 1 && K <= 32 && F >= 2 && F <= 16 && L >= 1 && L <= 64;
}
size_t ttcn_full_pack_floats(int F) { return (size_t)2 * KP * KP + 2 * KP + (size_t)F * 32 * KP + F * 32 + (size_t)F * 32 * KP + KP * KP + 64; }
size_t ttcn_full_slab_floats(int F) { return (size_t)kMaxBwdGrid * slab_of(F).total; }     // one slab per workgroup of the backward

int launch_ttcn_full_fwd(int P, int L, int F, int K, const float* x, const float* tt, const float* mask, const immtsf_ttcn_params* p,
                         float* pack, float* ctr, float* out, int out_ld, int flag_col, hipStream_t s) {
    const FD d{P, L, F, K, F * 32};
    const PackPtrs q = pack_ptrs(pack, F);
    hipLaunchKernelGGL(ttcn_pack_kernel, dim3(cdiv(F * 32 * KP, 256)), dim3(256), 0, s, F, K, PackIn{p->W1, p->b1, p->W2, p->b2, p->W3, p->b3},
                       q.W1p, q.b1p, q.W2p, q.b2p, q.W3q, q.b3q, q.W3T, q.W2T, q.W1T, q.W3h);
    IMMTSF_LAUNCH_CHECK();
    const TEp te{p->te_scale_w, p->te_scale_b, p->te_per_w, p->te_per_b};
    const Wts w{q.W1p, q.b1p, q.W2p, q.b2p, q.W3q, q.b3q};
    const int mf = (F + 3) / 4;
#define TTCN_FWD(RT, MF)                                                                                                                     \
    hipLaunchKernelGGL((ttcn_full_fwd_kernel<RT, MF>), dim3(P < 2048 ? P : 2048), dim3(256), fwd_lds(RT, d.NCq), s, d, x, tt, mask, te, w, q.W3h, p->T_bias, ctr, \
                       out, out_ld, flag_col)
    if (L <= 32) {
        if (mf == 1) TTCN_FWD(2, 1); else if (mf == 2) TTCN_FWD(2, 2); else if (mf == 3) TTCN_FWD(2, 3); else TTCN_FWD(2, 4);
    } else {
        if (mf == 1) TTCN_FWD(4, 1); else if (mf == 2) TTCN_FWD(4, 2); else if (mf == 3) TTCN_FWD(4, 3); else TTCN_FWD(4, 4);
    }
#undef TTCN_FWD
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_ttcn_full_bwd(int P, int L, int F, int K, const float* x, const float* tt, const float* mask, const immtsf_ttcn_params* p,
                         const float* pack, const float* ctr, const float* out, const float* dout, int out_ld, float* slab,
                         const immtsf_ttcn_params* gr, hipStream_t s, int te_acc) {
    const FD d{P, L, F, K, F * 32, 0};
    const PackPtrs q = pack_ptrs(const_cast<float*>(pack), F);
    const TEp te{p->te_scale_w, p->te_scale_b, p->te_per_w, p->te_per_b};
    const Wts w{q.W1p, q.b1p, q.W2p, q.b2p, q.W3q, q.b3q};
    const WtsT wt{q.W3T, q.W2T, q.W1T, q.W3h};
    // persistent workgroups; the per-patch work is a chain of small dependent MFMA / LDS phases (latency-, not throughput-bound), so a
    // second workgroup per CU hides it: the L <= 32 instances are held to 256 registers / lane (two waves per SIMD).  Measured inside the
    // cfg2 step, beside the text-side backward GEMMs on the other stream: one wave / SIMD at 160 / 192 / 224 / 256 workgroups ->
    // 0.891 / 0.882 / 0.874 / 0.887 ms (later 0.850 at 224); two waves / SIMD at 256 / 342 / 384 / 448 / 512 -> 0.856 / 0.847 / 0.839 /
    // 0.841 / 0.852.  Alone (P = 1024): 103 us at 224 x 1 wave, 80 us at 448 x 2 waves.
    const int genv = g_immtsf_ttcn_bwd_grid;      // tool switch (immtsf_debug_gemm_config bits 17..26), 0 = the rule below
    int gmax = genv > 0 ? genv : (L <= 32 ? (P >= 8192 ? 512 : 384) : 224);     // many patches: exactly two workgroups per CU
    gmax = gmax > kMaxBwdGrid ? kMaxBwdGrid : gmax;
    const int grid = P < gmax ? P : gmax;
    const size_t lds = bwd_lds(L <= 32 ? 2 : 4, d.NCq);
    const int mf = (F + 3) / 4;
#define TTCN_BWD(RT, MF)                                                                                                                  \
    do {                                                                                                                                  \
        if (lds > 64 * 1024)                                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ttcn_full_bwd_kernel<RT, MF>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                                          \
        hipLaunchKernelGGL((ttcn_full_bwd_kernel<RT, MF>), dim3(grid), dim3(256), lds, s, d, x, tt, mask, te, w, wt, ctr, out, dout, out_ld, \
                           slab);                                                                                                         \
    } while (0)
    if (L <= 32) {
        if (mf == 1) TTCN_BWD(2, 1); else if (mf == 2) TTCN_BWD(2, 2); else if (mf == 3) TTCN_BWD(2, 3); else TTCN_BWD(2, 4);
    } else {
        if (mf == 1) TTCN_BWD(4, 1); else if (mf == 2) TTCN_BWD(4, 2); else if (mf == 3) TTCN_BWD(4, 3); else TTCN_BWD(4, 4);
    }
#undef TTCN_BWD
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL(ttcn_unpack_kernel, dim3(cdiv(ttcn_unpack_outputs(F, K), 32)), dim3(256), 0, s, F, K, slab, grid,
                       UnpackOut{gr->W1, gr->b1, gr->W2, gr->b2, gr->W3, gr->b3, gr->te_scale_w, gr->te_scale_b, gr->te_per_w, gr->te_per_b,
                                 gr->T_bias}, te_acc);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
