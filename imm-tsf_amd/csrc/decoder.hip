// tPatchGNN's forecast decoder (models/tPatchGNN.py:168-174 applied at :283-291) as one kernel per direction.
//
// The reference repeats the encoder state h (B, N, D) over the Lp prediction steps and the time embedding te (B, Lp, E)
// over the N variables, concatenates them into a (B*N*Lp, D+E) matrix and runs Linear -> ReLU -> Linear -> ReLU ->
// Linear(H, 1) on it.  As GEMMs that is 3 launches forward and 6 backward over 16 k rows of 32-42 columns (~120 us of
// launch latency per step at the benchmark shape for 50 MFLOP).  The first layer is separable,
//     W1 [h ; te] = W1[:, :D] h[b, n] + W1[:, D:] te[b, lp],
// so a workgroup per window computes u[n] and v[lp] once (N + Lp small mat-vecs), and every (n, lp) row is then one
// thread: h1 = relu(u[n] + v[lp]), 32 x 32 FMAs against W2 broadcast from LDS, a dot with w3.  Exact fp32 (plain
// v_fma), in both precision modes -- the whole decoder is 35 MFLOP.  The backward recomputes the forward per row, runs
// dW2 = dH2^T H1 as register-blocked FMAs over an LDS image of the chunk's rows, reduces the first layer's gradient
// over lp (-> du[n]) and over n (-> dv[lp]) in LDS; a workgroup walks several windows (<= one resident workgroup per CU
// slot) with the parameter gradients in registers / LDS and adds them to global memory with one atomic per element at the end.
#include "../../include/immtsf.h"
#include "common.hpp"
#include <stdlib.h>

namespace {

struct DecDims { int B, N, Lp, D, E; };
struct DecP { const float *W1, *b1, *W2, *b2, *W3, *b3; };
struct DecG { float *W1, *b1, *W2, *b2, *W3, *b3; };
// LearnableTE of the prediction times inside the decoder (models/tPatchGNN.py:176-180 applied at :283-285): te[b, lp, :] =
// [w0 t + b0 ; sin(w t + b)] is built from t (B, Lp) while the window is staged, and the backward turns its d te rows into the four
// parameter gradients on the spot -- the separate Time2Vec launches (forward 4 us; backward: 14 us of ONE workgroup on the backbone's
// dependent chain) and the (B, Lp, E) te / d te tensors disappear.  t == nullptr: te / dte are the caller's tensors.
struct DecTE { const float *t, *w0, *b0, *w, *b; float *gw0, *gb0, *gw, *gb; };
__device__ __forceinline__ float dec_te_value(const DecTE& q, float t, int e) {
    return e == 0 ? fmaf(q.w0[0], t, q.b0[0]) : sinf(fmaf(q.w[e - 1], t, q.b[e - 1]));
}
// d te[b, lp, e] = a  ->  d w_e += g t, d b_e += g with g = a (e = 0) or a cos(w t + b): LDS sums [0..16) d w | [16..32) d b (E <= 16)
__device__ __forceinline__ void dec_te_grad(const DecTE& q, float t, int e, float a, float* gtes) {
    const float g = e == 0 ? a : a * cosf(fmaf(q.w[e - 1], t, q.b[e - 1]));
    atomicAdd(gtes + e, g * t);
    atomicAdd(gtes + 16 + e, g);
}
__device__ __forceinline__ void dec_te_flush(const DecTE& q, int i, int E, const float* gtes) {      // i < 2 E: one global atomic each
    const int e = i < E ? i : i - E;
    float* dst = i < E ? (e == 0 ? q.gw0 : q.gw + e - 1) : (e == 0 ? q.gb0 : q.gb + e - 1);
    atomicAdd(dst, gtes[(i < E ? 0 : 16) + e]);
}

// chunk geometry: 256 rows = NC variables x LPC steps, LPC = smallest power of two >= min(Lp, 256)
__host__ __device__ inline int lpc_of(int Lp) { int c = 1; while (c < Lp && c < 256) c <<= 1; return c; }

template <int H> struct Geo {
    static constexpr int P1 = H + 1;          // pitch of the per-variable / per-step vectors (conflict-free column walks)
    static constexpr int PR = H + 4;          // pitch of the per-row chunk images (16-byte aligned rows)
    static constexpr int KPT = H * H / 256;   // dW2 entries per thread: KPT consecutive k of one j
    static_assert(KPT >= 1 && H % KPT == 0 && KPT % 4 == 0, "hidden width: 32 (64 would need 128-row chunk images to fit LDS)");
};

// staged copies of W1 (pitch D+E+1), h[b] and te[b]: the first layer and the last phase of the backward read them many
// times with per-thread strides (from global memory that was 13 of the forward's 17 us)
__host__ __device__ inline size_t stage_floats(int H, int N, int Lp, int D, int E) { return (size_t)H * (D + E + 1) + (size_t)N * D + (size_t)Lp * E; }
template <int H>
size_t fwd_lds(int N, int Lp, int D, int E) {
    return (size_t)(H * H + 3 * H + (N + Lp) * Geo<H>::P1 + stage_floats(H, N, Lp, D, E)) * sizeof(float);
}
template <int H>
size_t bwd_lds(int N, int Lp, int D, int E) {       // ... + the workgroup's running dW1 / db1 (it walks several windows)
    return (size_t)(H * H + 3 * H + 2 * (N + Lp) * Geo<H>::P1 + 2 * 256 * Geo<H>::PR + 2 * H + 8 + H * (D + E) + H + 32) * sizeof(float);
}

// coalesced copies: W1s[k][D+E+1], hs[n][D], tes[lp][E] (contiguous, in this order, at `st`)
template <int H>
__device__ __forceinline__ void stage(const DecDims& d, const DecP& p, const float* __restrict__ h, const float* __restrict__ te, int b,
                                      float* st, const DecTE& tq, int nthreads = 256, bool with_w1 = true) {
    const int ld = d.D + d.E, pw = ld + 1;
    float* hs = st + H * pw;
    float* tes = hs + d.N * d.D;
    if (with_w1)        // (a persistent workgroup whose image of W1 survives its windows stages it once)
        for (int i = threadIdx.x; i < H * ld; i += nthreads) st[(i / ld) * pw + i % ld] = p.W1[i];
    for (int i = threadIdx.x; i < d.N * d.D; i += nthreads) hs[i] = h[(size_t)b * d.N * d.D + i];
    if (tq.t) {
        for (int i = threadIdx.x; i < d.Lp * d.E; i += nthreads) tes[i] = dec_te_value(tq, tq.t[(size_t)b * d.Lp + i / d.E], i % d.E);
    } else {
        for (int i = threadIdx.x; i < d.Lp * d.E; i += nthreads) tes[i] = te[(size_t)b * d.Lp * d.E + i];
    }
}
// u[n][k] = b1[k] + sum_d W1[k][d] h[b, n, d];  v[lp][k] = sum_e W1[k][D + e] te[b, lp, e]   (operands staged by stage())
template <int H>
__device__ __forceinline__ void first_layer(const DecDims& d, const float* __restrict__ b1, const float* st, float* u, float* v, int nthreads = 256) {
    constexpr int P1 = Geo<H>::P1;
    const int pw = d.D + d.E + 1;
    const float* hs = st + H * pw;
    const float* tes = hs + d.N * d.D;
    for (int i = threadIdx.x; i < (d.N + d.Lp) * H; i += nthreads) {
        const int r = i / H, k = i - r * H;
        const float* w = st + k * pw;
        float a;
        if (r < d.N) {
            a = b1[k];
            const float* x = hs + r * d.D;
#pragma unroll 8
            for (int q = 0; q < d.D; ++q) a = fmaf(w[q], x[q], a);
            u[r * P1 + k] = a;
        } else {
            a = 0.f;
            const float* x = tes + (r - d.N) * d.E;
#pragma unroll 8
            for (int q = 0; q < d.E; ++q) a = fmaf(w[d.D + q], x[q], a);
            v[(r - d.N) * P1 + k] = a;
        }
    }
}

// h2 = relu(W2 h1 + b2) for one row; W2s is the row-major LDS copy (every lane reads the same address: broadcast)
template <int H>
__device__ __forceinline__ void second_layer(const float* W2s, const float* b2s, const float (&h1)[H], float (&h2)[H]) {
    // row j + 1 is read while row j is used (one wave per SIMD here: nothing else hides the LDS latency); the scheduling
    // barrier keeps it at ONE row ahead -- left alone the scheduler front-loads all H*H reads into registers
    float4 wn[H / 4];
#pragma unroll
    for (int k = 0; k < H; k += 4) wn[k / 4] = *reinterpret_cast<const float4*>(W2s + k);
#pragma unroll
    for (int j = 0; j < H; ++j) {
        float4 w[H / 4];
#pragma unroll
        for (int q = 0; q < H / 4; ++q) w[q] = wn[q];
        if (j + 1 < H) {
#pragma unroll
            for (int k = 0; k < H; k += 4) wn[k / 4] = *reinterpret_cast<const float4*>(W2s + (j + 1) * H + k);
        }
        float a = b2s[j];
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            a = fmaf(w[k / 4].x, h1[k], a); a = fmaf(w[k / 4].y, h1[k + 1], a);
            a = fmaf(w[k / 4].z, h1[k + 2], a); a = fmaf(w[k / 4].w, h1[k + 3], a);
        }
        h2[j] = fmaxf(a, 0.f);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int H>
__global__ __launch_bounds__(256) void dec_fwd_kernel(DecDims d, DecP p, const float* __restrict__ h, const float* __restrict__ te,
                                                       float* __restrict__ out, DecTE tq) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int P1 = Geo<H>::P1;
    float* W2s = sm;
    float* b2s = W2s + H * H;
    float* w3s = b2s + H;
    float* pad = w3s + H;           // keeps u 16-byte aligned whatever H
    float* u = pad + H;
    float* v = u + d.N * P1;
    float* st = v + d.Lp * P1;
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < H * H; i += 256) W2s[i] = p.W2[i];
    if (threadIdx.x < H) { b2s[threadIdx.x] = p.b2[threadIdx.x]; w3s[threadIdx.x] = p.W3[threadIdx.x]; }
    stage<H>(d, p, h, te, b, st, tq);
    __syncthreads();
    first_layer<H>(d, p.b1, st, u, v);
    __syncthreads();
    const float b3 = p.b3[0];
    for (int r = threadIdx.x; r < d.N * d.Lp; r += 256) {
        asm volatile("" ::: "memory");      // keeps the H*H loop-invariant LDS reads of W2 inside the loop (registers!)
        const int n = r / d.Lp, lp = r - n * d.Lp;
        float h1[H], h2[H];
#pragma unroll
        for (int k = 0; k < H; ++k) h1[k] = fmaxf(u[n * P1 + k] + v[lp * P1 + k], 0.f);
        second_layer<H>(W2s, b2s, h1, h2);
        float y = b3;
#pragma unroll
        for (int j = 0; j < H; ++j) y = fmaf(w3s[j], h2[j], y);
        out[((size_t)b * d.Lp + lp) * d.N + n] = y;
    }
}

template <int H>
__global__ __launch_bounds__(256) void dec_bwd_kernel(DecDims d, DecP p, const float* __restrict__ h, const float* __restrict__ te,
                                                       const float* __restrict__ dout, float* __restrict__ dh,
                                                       float* __restrict__ dte, DecG g, DecTE tq) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int P1 = Geo<H>::P1, PR = Geo<H>::PR, KPT = Geo<H>::KPT;
    float* W2s = sm;
    float* b2s = W2s + H * H;
    float* w3s = b2s + H;
    float* pad = w3s + H;
    float* Xs = pad + H;                      // chunk image 1: h1, later the masked first-layer gradient
    float* Ys = Xs + 256 * PR;                // chunk image 2: dh2, later dy * h2
    float* u = Ys + 256 * PR;
    float* v = u + d.N * P1;
    float* du = v + d.Lp * P1;
    float* dv = du + d.N * P1;
    float* dw3s = dv + d.Lp * P1;             // [H]
    float* db2s = dw3s + H;                   // [H]
    float* db3s = db2s + H;                   // [1] (+ 7 of padding)
    float* gW1s = db3s + 8;                   // [H][D + E] running dW1 of this workgroup's windows
    float* gb1s = gW1s + H * (d.D + d.E);     // [H]
    float* gtes = gb1s + H;                   // [2][16]: running d w | d b of the fused LearnableTE
    const int tid = threadIdx.x;
    if (tid < 32) gtes[tid] = 0.f;
    for (int i = tid; i < H * H; i += 256) W2s[i] = p.W2[i];
    if (tid < H) { b2s[tid] = p.b2[tid]; w3s[tid] = p.W3[tid]; dw3s[tid] = 0.f; db2s[tid] = 0.f; gb1s[tid] = 0.f; }
    if (tid == 0) db3s[0] = 0.f;
    for (int i = tid; i < H * (d.D + d.E); i += 256) gW1s[i] = 0.f;
    const int LPC = lpc_of(d.Lp), NC = 256 / LPC;
    const int ln = tid / LPC, llp = tid - ln * LPC;
    const int wj = tid / (H / KPT), wk = (tid % (H / KPT)) * KPT;      // this thread's strip of dW2
    float accW[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) accW[i] = 0.f;
    float accb2 = 0.f;

    // a workgroup walks windows b, b + grid, ...: the parameter gradients stay in registers / LDS across them and reach global
    // memory once per workgroup (with one workgroup per window the ~2.5 k atomics of each of 4096 windows queue up per address)
    for (int b = blockIdx.x; b < d.B; b += gridDim.x) {
    for (int i = tid; i < (d.N + d.Lp) * P1; i += 256) du[i] = 0.f;      // du and dv are contiguous
    stage<H>(d, p, h, te, b, Xs, tq);       // the staged operands borrow the first chunk image (launcher checks they fit)
    __syncthreads();
    first_layer<H>(d, p.b1, Xs, u, v);
    __syncthreads();

    for (int n0 = 0; n0 < d.N; n0 += NC)
        for (int lp0 = 0; lp0 < d.Lp; lp0 += LPC) {
            asm volatile("" ::: "memory");      // as in the forward: no hoisting of the W2 reads out of the chunk loop
            const int n = n0 + ln, lp = lp0 + llp;
            const bool live = n < d.N && lp < d.Lp;
            float h1[H], h2[H], z[H];
            float dy = 0.f;
            if (live) {
                dy = dout[((size_t)b * d.Lp + lp) * d.N + n];
#pragma unroll
                for (int k = 0; k < H; ++k) h1[k] = fmaxf(u[n * P1 + k] + v[lp * P1 + k], 0.f);
            } else {
#pragma unroll
                for (int k = 0; k < H; ++k) h1[k] = 0.f;
            }
            second_layer<H>(W2s, b2s, h1, h2);
            asm volatile("" ::: "memory");      // W2 is read again below: re-read it rather than keep H*H values in registers
            // dh2 = dy w3 [h2 > 0] -> image 2; h2 <- dy * h2 (this row's contribution to dw3); dh1 = W2^T dh2
            float g2v[H];
#pragma unroll
            for (int j = 0; j < H; ++j) {
                const float t = dy * w3s[j];
                g2v[j] = h2[j] > 0.f ? t : 0.f;
                h2[j] *= dy;
            }
#pragma unroll
            for (int j = 0; j < H; j += 4)
                *reinterpret_cast<float4*>(Ys + tid * PR + j) = make_float4(g2v[j], g2v[j + 1], g2v[j + 2], g2v[j + 3]);
#pragma unroll
            for (int k = 0; k < H; ++k) z[k] = 0.f;
            float4 wn[H / 4];
#pragma unroll
            for (int k = 0; k < H; k += 4) wn[k / 4] = *reinterpret_cast<const float4*>(W2s + k);
#pragma unroll
            for (int j = 0; j < H; ++j) {       // one row ahead, as in second_layer
                float4 w[H / 4];
#pragma unroll
                for (int q = 0; q < H / 4; ++q) w[q] = wn[q];
                if (j + 1 < H) {
#pragma unroll
                    for (int k = 0; k < H; k += 4) wn[k / 4] = *reinterpret_cast<const float4*>(W2s + (j + 1) * H + k);
                }
#pragma unroll
                for (int k = 0; k < H; k += 4) {
                    z[k] = fmaf(w[k / 4].x, g2v[j], z[k]); z[k + 1] = fmaf(w[k / 4].y, g2v[j], z[k + 1]);
                    z[k + 2] = fmaf(w[k / 4].z, g2v[j], z[k + 2]); z[k + 3] = fmaf(w[k / 4].w, g2v[j], z[k + 3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < H; k += 4)
                *reinterpret_cast<float4*>(Xs + tid * PR + k) = make_float4(h1[k], h1[k + 1], h1[k + 2], h1[k + 3]);
            __syncthreads();
            // dW2[wj][wk ..] += sum_rows dh2[row][wj] * h1[row][wk ..]; db2[wj] += sum_rows dh2[row][wj]
#pragma unroll 8
            for (int r = 0; r < 256; ++r) {
                const float g2 = Ys[r * PR + wj];
#pragma unroll
                for (int i = 0; i < KPT; i += 4) {
                    const float4 x = *reinterpret_cast<const float4*>(Xs + r * PR + wk + i);
                    accW[i] = fmaf(g2, x.x, accW[i]); accW[i + 1] = fmaf(g2, x.y, accW[i + 1]);
                    accW[i + 2] = fmaf(g2, x.z, accW[i + 2]); accW[i + 3] = fmaf(g2, x.w, accW[i + 3]);
                }
                if (wk == 0) accb2 += g2;
            }
            __syncthreads();
            // second image: masked first-layer gradient and dy * h2
#pragma unroll
            for (int k = 0; k < H; k += 4) {
                *reinterpret_cast<float4*>(Xs + tid * PR + k) = make_float4(h1[k] > 0.f ? z[k] : 0.f, h1[k + 1] > 0.f ? z[k + 1] : 0.f,
                                                                            h1[k + 2] > 0.f ? z[k + 2] : 0.f, h1[k + 3] > 0.f ? z[k + 3] : 0.f);
                *reinterpret_cast<float4*>(Ys + tid * PR + k) = make_float4(h2[k], h2[k + 1], h2[k + 2], h2[k + 3]);
            }
            __syncthreads();
            // du[n][k] += sum over the chunk's steps; dv[lp][k] += sum over the chunk's variables (one owner per entry)
            for (int i = tid; i < NC * H; i += 256) {
                const int cn = i / H, k = i - cn * H;
                if (n0 + cn < d.N) {
                    float a = 0.f;
#pragma unroll 8
                    for (int q = 0; q < LPC; ++q) a += Xs[(cn * LPC + q) * PR + k];
                    du[(n0 + cn) * P1 + k] += a;
                }
            }
            for (int i = tid; i < LPC * H; i += 256) {
                const int cl = i / H, k = i - cl * H;
                if (lp0 + cl < d.Lp) {
                    float a = 0.f;
#pragma unroll 8
                    for (int q = 0; q < NC; ++q) a += Xs[(q * LPC + cl) * PR + k];
                    dv[(lp0 + cl) * P1 + k] += a;
                }
            }
            {   // dw3[j] += sum_rows dy h2[j]: 256 / H row groups per j, combined with LDS atomics; db3 += sum dy
                const int j = tid % H, part = tid / H, per = 256 / (256 / H);
                float a = 0.f;
#pragma unroll 8
                for (int q = part * per; q < (part + 1) * per; ++q) a += Ys[q * PR + j];
                atomicAdd(dw3s + j, a);
                const float s = wave_sum(dy);
                if ((tid & 63) == 0) atomicAdd(db3s, s);
            }
            __syncthreads();
        }

    // ---- this window's first-layer parameter gradients -> the running sums, data gradients dh / dte
    if (tid < H) {
        float s = 0.f;
        for (int n = 0; n < d.N; ++n) s += du[n * P1 + tid];
        gb1s[tid] += s;
    }
    stage<H>(d, p, h, te, b, Xs, tq);       // the chunk images are dead: W1 / h / te again for the first layer's gradients
    __syncthreads();
    const int ld = d.D + d.E, pw = ld + 1;
    const float* hs = Xs + H * pw;
    const float* tes = hs + d.N * d.D;
    for (int i = tid; i < H * ld; i += 256) {          // dW1[k][c]
        const int k = i / ld, c = i - k * ld;
        float a = 0.f;
        if (c < d.D) {
#pragma unroll 8
            for (int n = 0; n < d.N; ++n) a = fmaf(du[n * P1 + k], hs[n * d.D + c], a);
        } else {
#pragma unroll 8
            for (int lp = 0; lp < d.Lp; ++lp) a = fmaf(dv[lp * P1 + k], tes[lp * d.E + c - d.D], a);
        }
        gW1s[i] += a;
    }
    for (int i = tid; i < d.N * d.D; i += 256) {        // dh[b, n, c] = sum_k W1[k][c] du[n][k]
        const int n = i / d.D, c = i - n * d.D;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) a = fmaf(Xs[k * pw + c], du[n * P1 + k], a);
        dh[(size_t)b * d.N * d.D + i] = a;
    }
    for (int i = tid; i < d.Lp * d.E; i += 256) {       // dte[b, lp, e] = sum_k W1[k][D + e] dv[lp][k]
        const int lp = i / d.E, e = i - lp * d.E;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) a = fmaf(Xs[k * pw + d.D + e], dv[lp * P1 + k], a);
        if (tq.t) dec_te_grad(tq, tq.t[(size_t)b * d.Lp + lp], e, a, gtes);
        else dte[(size_t)b * d.Lp * d.E + i] = a;
    }
    __syncthreads();        // the next window restages over Xs and clears du / dv
    }   // windows
    if (tq.t && tid < 2 * d.E) dec_te_flush(tq, tid, d.E, gtes);

    // ---- parameter gradients of this workgroup's windows -> global (one atomic per element)
#pragma unroll
    for (int i = 0; i < KPT; ++i) atomicAdd(g.W2 + wj * H + wk + i, accW[i]);
    if (wk == 0) atomicAdd(g.b2 + wj, accb2);
    if (tid < H) {
        atomicAdd(g.W3 + tid, dw3s[tid]);
        atomicAdd(g.b1 + tid, gb1s[tid]);
    }
    if (tid == 0) atomicAdd(g.b3, db3s[0]);
    for (int i = tid; i < H * (d.D + d.E); i += 256) atomicAdd(g.W1 + i, gW1s[i]);
}

// ---------------------------------------------------------------------------------------------------- bf16 MFMA variants
// precision 1 (bf16 operands, fp32 accumulation -- the mode of every other product of the step).  The (n, lp) rows of a window
// are 16-row MFMA tiles; H = 32 is one K-step of v_mfma_f32_16x16x32_bf16.  h1 = relu(u[n] + v[lp]) is cheap to form from the LDS
// vectors in ANY operand layout, so nothing is ever transposed through memory:
//   forward   z2^T = W2 h1^T (A = W2 rows, B = h1: lane = row, 4 consecutive j per register quad) -> bias, ReLU, dot with w3 over
//             the lane's 8 j + two xor shuffles;
//   backward  the same product gives g2 = dy w3 [z2 > 0] in the A layout of dh1 = g2 W2 (B = W2 with its j taken in the
//             accumulator order), whose result (lane = k, rows in registers) is masked by [h1 > 0] and added to du[n] / dv[lp]
//             in LDS; the product with the operands swapped gives g2 with lane = j -- the B layout of dW2^T = h1^T g2 over a
//             pair of row tiles (rows in the accumulator order on both sides, h1^T formed directly from u / v).
// dW2 / db2 / dW3 / db3 live in registers across the windows of a persistent workgroup; the first layer's gradients and dh / dte
// are the fp32 code of the exact kernel.  ~16 MFMAs + ~400 VALU instructions per 32 rows instead of ~3000 FMAs per row.
typedef float dec_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dec_f32x4 dec_mfma(bf16x8 a, bf16x8 b, dec_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ bf16x8 dec_load8(const float* __restrict__ src) {
    const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
    bf16x8 r;
    r[0] = (bf16_t)a.x; r[1] = (bf16_t)a.y; r[2] = (bf16_t)a.z; r[3] = (bf16_t)a.w;
    r[4] = (bf16_t)b.x; r[5] = (bf16_t)b.y; r[6] = (bf16_t)b.z; r[7] = (bf16_t)b.w;
    return r;
}
// h1 fragment of row r (operand index = lane & 15): k = 8 fq .. + 7
__device__ __forceinline__ bf16x8 dec_h1_frag(const float* u, const float* v, int P1, int n, int lp, int fq) {
    bf16x8 r;
#pragma unroll
    for (int s = 0; s < 8; ++s) r[s] = (bf16_t)fmaxf(u[n * P1 + fq * 8 + s] + v[lp * P1 + fq * 8 + s], 0.f);
    return r;
}

template <int H>
size_t fwd_lds_mfma(int N, int Lp, int D, int E) { return (size_t)((N + Lp) * Geo<H>::P1 + stage_floats(H, N, Lp, D, E) + 8) * sizeof(float); }
template <int H>
size_t bwd_lds_mfma(int N, int Lp, int D, int E, bool with_slabs) {
    const size_t red = 8 * (H * H + 2 * H + 4), slabs = with_slabs ? (size_t)8 * (Lp + 1) * Geo<H>::P1 : 0;      // the waves' parameter sums | their dv slabs (same region)
    return (size_t)(2 * (N + Lp) * Geo<H>::P1 + stage_floats(H, N, Lp, D, E) + ((N * Lp + 31) & ~31) + H * (D + E) + H + 32 + (red > slabs ? red : slabs) + 16) * sizeof(float);
}

// grid B, 256 threads
__global__ __launch_bounds__(256) void dec_fwd_mfma_kernel(DecDims d, DecP p, const float* __restrict__ h, const float* __restrict__ te,
                                                            float* __restrict__ out, DecTE tq) {
    constexpr int H = 32, P1 = Geo<H>::P1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* u = sm;
    float* v = u + d.N * P1;
    float* st = v + d.Lp * P1;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const bf16x8 wA0 = dec_load8(p.W2 + (size_t)fr * H + fq * 8), wA1 = dec_load8(p.W2 + (size_t)(16 + fr) * H + fq * 8);
    float b2v[2][4], w3v[2][4];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e) { b2v[jt][e] = p.b2[jt * 16 + fq * 4 + e]; w3v[jt][e] = p.W3[jt * 16 + fq * 4 + e]; }
    const float b3 = p.b3[0];
    stage<H>(d, p, h, te, b, st, tq);
    __syncthreads();
    first_layer<H>(d, p.b1, st, u, v);
    __syncthreads();
    const int rows = d.N * d.Lp, tiles = (rows + 15) >> 4;
    for (int t = wave; t < tiles; t += 4) {
        const int r = t * 16 + fr, rc = r < rows ? r : rows - 1, n = rc / d.Lp, lp = rc - n * d.Lp;
        const bf16x8 hb = dec_h1_frag(u, v, P1, n, lp, fq);
        const dec_f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const dec_f32x4 z0 = dec_mfma(wA0, hb, z), z1 = dec_mfma(wA1, hb, z);
        float y = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y = fmaf(w3v[0][e], fmaxf(z0[e] + b2v[0][e], 0.f), y);
            y = fmaf(w3v[1][e], fmaxf(z1[e] + b2v[1][e], 0.f), y);
        }
        y = xor32_sum(xor16_sum(y));
        if (fq == 0 && r < rows) out[((size_t)b * d.Lp + lp) * d.N + n] = y + b3;
    }
}

// persistent workgroups (grid <= B), 512 threads (eight waves share a window's 16-row tiles: the per-tile chain of dependent LDS
// reads / MFMAs / LDS atomics is latency-bound, more waves per window is what shortens it)
__global__ __launch_bounds__(512) void dec_bwd_mfma_kernel(DecDims d, DecP p, const float* __restrict__ h, const float* __restrict__ te,
                                                            const float* __restrict__ dout, float* __restrict__ dh,
                                                            float* __restrict__ dte, DecG g, int use_slabs, DecTE tq) {
    constexpr int H = 32, P1 = Geo<H>::P1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* u = sm;
    float* v = u + d.N * P1;
    float* du = v + d.Lp * P1;
    float* dv = du + d.N * P1;
    float* Xs = dv + d.Lp * P1;                        // staged W1 | h[b] | te[b]
    float* dys = Xs + stage_floats(H, d.N, d.Lp, d.D, d.E);     // dy of the window's rows, r = n * Lp + lp
    float* gW1s = dys + ((d.N * d.Lp + 31) & ~31);
    float* gb1s = gW1s + H * (d.D + d.E);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    constexpr int NT = 512, NW = NT / 64;
    // W2 in the three operand forms (registers for the whole kernel)
    bf16x8 wA[2], wP[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        wA[t] = dec_load8(p.W2 + (size_t)(t * 16 + fr) * H + fq * 8);           // [j = 16 t + fr][k = 8 fq ..]: A of z2^T, B of z2
#pragma unroll
        for (int s = 0; s < 8; ++s)                                             // B of dh1: [j in accumulator order][k = 16 t + fr]
            wP[t][s] = (bf16_t)p.W2[(size_t)((s >> 2) * 16 + fq * 4 + (s & 3)) * H + t * 16 + fr];
    }
    float b2o1[2][4], w3o1[2][4], b2o2[2], w3o2[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { b2o1[jt][e] = p.b2[jt * 16 + fq * 4 + e]; w3o1[jt][e] = p.W3[jt * 16 + fq * 4 + e]; }
        b2o2[jt] = p.b2[jt * 16 + fr];
        w3o2[jt] = p.W3[jt * 16 + fr];
    }
    dec_f32x4 accW2[2][2];            // [kt][jt]: dW2[j = 16 jt + fr][k = 16 kt + 4 fq + e]
    float accw3[2][4], accb2[2][4], accb3 = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) accW2[a][c] = dec_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) accw3[a][e] = accb2[a][e] = 0.f;
    for (int i = tid; i < H * (d.D + d.E); i += NT) gW1s[i] = 0.f;
    if (tid < H) gb1s[tid] = 0.f;
    const int rows = d.N * d.Lp, tiles = (rows + 15) >> 4, pairs = (tiles + 1) >> 1;
    // dv of a wave's tiles goes to a slab of its own, [Lp + 1][P1] (the last row takes the padding rows), by plain read-modify-write:
    // fp32 LDS atomics retire about one lane per clock for the whole CU -- eight waves' 10 k lane-adds were 5 us per tile, 10 of the
    // tile loop's 12.6 us.  One ds instruction touches rows 4 fq + e (e fixed): distinct steps lp whenever Lp > 12.  The slabs share
    // the region of the final parameter sums (`red`), are added up in wave order after the loop: the sum is deterministic, too
    float* gtes = gb1s + H;                   // [2][16]: running d w | d b of the fused LearnableTE
    if (tid < 32) gtes[tid] = 0.f;
    float* red = gtes + 32;
    float te_gw = 0.f, te_gb = 0.f;          // this thread's share of the fused LearnableTE's d w[e] / d b[e], e = tid % 16
    // (its frequency and phase are loaded once, its step's time at the top of a window: nothing of that phase waits for memory)
    const int te_e = tid & 15;
    const bool te_on = tq.t != nullptr && te_e < d.E;
    const float te_w = (te_on && te_e > 0) ? tq.w[te_e - 1] : 0.f, te_b = (te_on && te_e > 0) ? tq.b[te_e - 1] : 0.f;
    const bool priv = use_slabs != 0;       // (the host: Lp > 12 and the slabs fit)
    const int slab_floats = (d.Lp + 1) * P1;
    float* slab = red + (size_t)wave * slab_floats;

    for (int b = blockIdx.x; b < d.B; b += gridDim.x) {
        const float te_t0 = (te_on && (tid >> 4) < d.Lp) ? tq.t[(size_t)b * d.Lp + (tid >> 4)] : 0.f;
        for (int i = tid; i < (d.N + d.Lp) * P1; i += NT) du[i] = 0.f;      // du and dv are contiguous
        for (int r = tid; r < pairs * 32; r += NT) {            // (padded to whole tile pairs with zeros)
            const int n = r / d.Lp, lp = r - n * d.Lp;
            dys[r] = r < rows ? dout[((size_t)b * d.Lp + lp) * d.N + n] : 0.f;
        }
        if (priv)
            for (int i = lane; i < slab_floats; i += 64) slab[i] = 0.f;
        stage<H>(d, p, h, te, b, Xs, tq, NT, b == (int)blockIdx.x);      // (nothing else is written to Xs in this kernel)
        __syncthreads();
        first_layer<H>(d, p.b1, Xs, u, v, NT);      // (every thread once: with the default stride the upper half of the 512 repeated the lower's rows)
        __syncthreads();
        for (int pr = wave; pr < pairs; pr += NW) {
            bf16x8 hb[2];
            dec_f32x4 g2o2[2][2];          // [t][jt]: g2 with lane = j (16 jt + fr), rows 4 fq + e of tile t
            int ou[2][4], ov[2][4];        // u / v offsets (n * P1, lp * P1) of rows 4 fq + e of tile t (clamped to the last row)
            int os[2][4];                  // the rows' offsets in the wave's dv slab (padding rows: the extra row)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int r0 = (pr * 2 + t) * 16;                  // wave-uniform: one division per tile, the lanes walk on from it
                const int r0c = r0 < rows ? r0 : rows - 1, n0 = r0c / d.Lp, lp0 = r0c - n0 * d.Lp;
                auto locate = [&](int o, int& un, int& vl) __attribute__((always_inline)) {
                    int rr = r0 + o;
                    rr = rr < rows ? rr : rows - 1;
                    int n = n0, lp = lp0 + (rr - r0c);
                    if (d.Lp >= 16) {                    // a tile spans at most two variables: one wrap
                        if (lp >= d.Lp) { lp -= d.Lp; ++n; }
                    } else {
                        n = rr / d.Lp;
                        lp = rr - n * d.Lp;
                    }
                    un = n * P1; vl = lp * P1;
                };
                int un1, vl1;
                locate(fr, un1, vl1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    locate(fq * 4 + e, ou[t][e], ov[t][e]);
                    os[t][e] = r0 + fq * 4 + e < rows ? ov[t][e] : d.Lp * P1;
                }
                const float dy1 = dys[r0 + fr];                     // (zero past the window's rows)
                {
                    bf16x8 r;
#pragma unroll
                    for (int q = 0; q < 8; ++q) r[q] = (bf16_t)fmaxf(u[un1 + fq * 8 + q] + v[vl1 + fq * 8 + q], 0.f);
                    hb[t] = r;
                }
                const dec_f32x4 z = {0.f, 0.f, 0.f, 0.f};
                dec_f32x4 gq[2];
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    const dec_f32x4 z1 = dec_mfma(wA[jt], hb[t], z);             // lane = row fr, j = 16 jt + 4 fq + e
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = z1[e] + b2o1[jt][e];
                        const float gv = a > 0.f ? dy1 * w3o1[jt][e] : 0.f;
                        accw3[jt][e] = fmaf(dy1, fmaxf(a, 0.f), accw3[jt][e]);
                        accb2[jt][e] += gv;
                        gq[jt][e] = gv;
                    }
                }
                if (fq == 0) accb3 += dy1;
                bf16x8 gA;
#pragma unroll
                for (int e = 0; e < 4; ++e) { gA[e] = (bf16_t)gq[0][e]; gA[4 + e] = (bf16_t)gq[1][e]; }
                // dh1 = g2 W2 -> [h1 > 0] -> du[n] / dv[lp]: lane = k (16 kt + fr), rows 4 fq + e (rows past the end: g2 = 0)
                // (rows grow with fq and e: first row in lane 0's e = 0, last in lane 63's e = 3 -- a wave-uniform test)
                const bool one_n = __builtin_amdgcn_readfirstlane(ou[t][0]) == __builtin_amdgcn_readlane(ou[t][3], 63);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const dec_f32x4 dh1 = dec_mfma(gA, wP[kt], z);
                    const int k = kt * 16 + fr;
                    float m[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[e] = u[ou[t][e] + k] + v[ov[t][e] + k] > 0.f ? dh1[e] : 0.f;
                    if (priv) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) slab[os[t][e] + k] += m[e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(dv + ov[t][e] + k, m[e]);
                    }
                    if (one_n) {      // the whole tile is one variable (Lp a multiple of 16): one add per k
                        const float sacc = xor32_sum(xor16_sum((m[0] + m[1]) + (m[2] + m[3])));
                        if (fq == 0) atomicAdd(du + ou[t][0] + k, sacc);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(du + ou[t][e] + k, m[e]);
                    }
                }
                // the same product with the operands swapped: lane = j
                const float4 dy4 = *reinterpret_cast<const float4*>(dys + r0 + fq * 4);
                const float dy2[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    const dec_f32x4 z2 = dec_mfma(hb[t], wA[jt], z);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g2o2[t][jt][e] = z2[e] + b2o2[jt] > 0.f ? dy2[e] * w3o2[jt] : 0.f;
                }
            }
            // dW2^T += h1^T g2 over the pair's 32 rows (rows in the accumulator order (4 fq + s | 16 + 4 fq + s - 4) on both sides)
            bf16x8 gB[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int e = 0; e < 4; ++e) { gB[jt][e] = (bf16_t)g2o2[0][jt][e]; gB[jt][4 + e] = (bf16_t)g2o2[1][jt][e]; }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                bf16x8 hT;
                const int k = kt * 16 + fr;
#pragma unroll
                for (int q = 0; q < 8; ++q) hT[q] = (bf16_t)fmaxf(u[ou[q >> 2][q & 3] + k] + v[ov[q >> 2][q & 3] + k], 0.f);
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) accW2[kt][jt] = dec_mfma(hT, gB[jt], accW2[kt][jt]);
            }
        }
        __syncthreads();
        if (priv) {
            for (int i = tid; i < d.Lp * P1; i += NT) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) a += red[w * slab_floats + i];
                dv[i] = a;
            }
            __syncthreads();
        }
        // ---- first layer (fp32, as in the exact kernel): db1, dW1 -> the running sums; dh, dte
        if (tid < H) {
            float s = 0.f;
            for (int n = 0; n < d.N; ++n) s += du[n * P1 + tid];
            gb1s[tid] += s;
        }
        const int ld = d.D + d.E, pw = ld + 1;
        const float* hs = Xs + H * pw;
        const float* tes = hs + d.N * d.D;
        // dW1 in two loops of uniform trip count (as one loop over (k, c) a wave walked BOTH sums -- N terms for the h columns, Lp for the
        // time-embedding columns -- for every output: 5.1 us of a window's 24 on the device clock)
        for (int i = tid; i < H * d.D; i += NT) {
            const int k = i / d.D, c = i - k * d.D;
            float a = 0.f;
#pragma unroll 8
            for (int n = 0; n < d.N; ++n) a = fmaf(du[n * P1 + k], hs[n * d.D + c], a);
            gW1s[k * ld + c] += a;
        }
        for (int i = tid; i < H * d.E; i += NT) {
            const int k = i / d.E, e = i - k * d.E;
            float a = 0.f;
#pragma unroll 8
            for (int lp = 0; lp < d.Lp; ++lp) a = fmaf(dv[lp * P1 + k], tes[lp * d.E + e], a);
            gW1s[k * ld + d.D + e] += a;
        }
        for (int i = tid; i < d.N * d.D; i += NT) {
            const int n = i / d.D, c = i - n * d.D;
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < H; ++k) a = fmaf(Xs[k * pw + c], du[n * P1 + k], a);
            dh[(size_t)b * d.N * d.D + i] = a;
        }
        if (tq.t) {
            // the fused LearnableTE's gradients: a thread = (step lp, element e = tid % 16), its sums stay in registers over the
            // workgroup's windows and meet once behind the loop.  (As two LDS atomics per (lp, e) onto 2 E addresses this phase was
            // 8.7 us of a window's 24 -- the atomics queue per address.)
            const int e = te_e;
            if (te_on) {
                for (int lp = tid >> 4; lp < d.Lp; lp += NT / 16) {
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < H; ++k) a = fmaf(Xs[k * pw + d.D + e], dv[lp * P1 + k], a);
                    const float t = lp == (tid >> 4) ? te_t0 : tq.t[(size_t)b * d.Lp + lp];
                    const float gv = e == 0 ? a : a * cosf(fmaf(te_w, t, te_b));
                    te_gw = fmaf(gv, t, te_gw);
                    te_gb += gv;
                }
            }
        } else {
            for (int i = tid; i < d.Lp * d.E; i += NT) {
                const int lp = i / d.E, e = i - lp * d.E;
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < H; ++k) a = fmaf(Xs[k * pw + d.D + e], dv[lp * P1 + k], a);
                dte[(size_t)b * d.Lp * d.E + i] = a;
            }
        }
        __syncthreads();
    }
    if (tq.t) {     // the waves' LearnableTE sums -> gtes (lanes with the same e: lane ^ 16, lane ^ 32; then the eight waves in order)
        te_gw = xor32_sum(xor16_sum(te_gw));
        te_gb = xor32_sum(xor16_sum(te_gb));
        if (lane < 16) { red[wave * 32 + lane] = te_gw; red[wave * 32 + 16 + lane] = te_gb; }
        __syncthreads();
        if (tid < 32) {
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sacc += red[w * 32 + tid];
            gtes[tid] = sacc;
        }
        __syncthreads();
    }
    // ---- this workgroup's parameter gradients -> global: the four waves' sums are added up in LDS first (one atomic per element
    // per WORKGROUP: the atomics queue per address, their count is what the flush costs)
    // red: [NW][RED]: W2 (1024) | b2 (32) | W3 (32) | b3 (1)
    constexpr int RED = H * H + 2 * H + 4;
    float* mine = red + wave * RED;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 4; ++e) mine[(jt * 16 + fr) * H + kt * 16 + fq * 4 + e] = accW2[kt][jt][e];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sb = row16_sum(accb2[jt][e]), sw = row16_sum(accw3[jt][e]);
            if (fr == 0) { mine[H * H + jt * 16 + fq * 4 + e] = sb; mine[H * H + H + jt * 16 + fq * 4 + e] = sw; }
        }
    accb3 = wave_sum(accb3);
    if (lane == 0) mine[H * H + 2 * H] = accb3;
    __syncthreads();
    for (int i = tid; i < H * H + 2 * H + 1; i += NT) {
        float v4 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v4 += red[w * RED + i];
        float* dst = i < H * H ? g.W2 + i : i < H * H + H ? g.b2 + (i - H * H) : i < H * H + 2 * H ? g.W3 + (i - H * H - H) : g.b3;
        atomicAdd(dst, v4);
    }
    if (tid < H) atomicAdd(g.b1 + tid, gb1s[tid]);
    for (int i = tid; i < H * (d.D + d.E); i += NT) atomicAdd(g.W1 + i, gW1s[i]);
    if (tq.t && tid < 2 * d.E) dec_te_flush(tq, tid, d.E, gtes);
}

constexpr size_t kLdsMax = 150 * 1024;

bool dims_ok(int B, int N, int Lp, int D, int E, int H) {
    return B >= 0 && N > 0 && Lp > 0 && D > 0 && E > 0 && H == 32;
}

}  // namespace

extern "C" {

size_t immtsf_tpatchgnn_decoder_lds_bytes(int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H) {
    if (!dims_ok(1, N, Lp, D, E, H)) return 0;
    const size_t b = bwd_lds<32>(N, Lp, D, E), f = fwd_lds<32>(N, Lp, D, E);
    if (stage_floats(32, N, Lp, D, E) > (size_t)2 * 256 * Geo<32>::PR) return 0;      // staged operands borrow the chunk images
    return b <= kLdsMax && f <= kLdsMax ? (b > f ? b : f) : 0;
}

int immtsf_tpatchgnn_decoder_forward(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, const float* h,
                                     const float* te, const immtsf_decoder_params* p, float* out, immtsf_stream_t stream) {
    return immtsf_tpatchgnn_decoder_forward_p(B, N, Lp, D, E, H, 0, h, te, p, out, stream);
}

static bool dec_mfma_ok(const immtsf_decoder_params* p) {
    constexpr bool on = true;
    return on && (reinterpret_cast<uintptr_t>(p->W2) & 15) == 0;
}

static int dec_forward(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, int32_t precision, const float* h,
                       const float* te, const DecTE& tq, const immtsf_decoder_params* p, float* out, immtsf_stream_t stream) {
    if (!dims_ok(B, N, Lp, D, E, H) || !h || (!te && !tq.t) || !p || !out || precision < 0 || precision > 1) return IMMTSF_EINVAL;
    if (immtsf_tpatchgnn_decoder_lds_bytes(N, Lp, D, E, H) == 0) return IMMTSF_EUNSUPPORTED;
    if (B == 0) return IMMTSF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DecDims d{B, N, Lp, D, E};
    const DecP q{p->W1, p->b1, p->W2, p->b2, p->W3, p->b3};
    if (precision == 1 && dec_mfma_ok(p)) {
        hipLaunchKernelGGL(dec_fwd_mfma_kernel, dim3(B), dim3(256), fwd_lds_mfma<32>(N, Lp, D, E), s, d, q, h, te, out, tq);
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    const size_t lds = fwd_lds<32>(N, Lp, D, E);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dec_fwd_kernel<32>, dim3(B), dim3(256), lds, s, d, q, h, te, out, tq);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int immtsf_tpatchgnn_decoder_forward_p(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, int32_t precision, const float* h,
                                       const float* te, const immtsf_decoder_params* p, float* out, immtsf_stream_t stream) {
    return dec_forward(B, N, Lp, D, E, H, precision, h, te, DecTE{}, p, out, stream);
}
/* LearnableTE inside the decoder: t (B, Lp) prediction times, tp = its four parameters (te_scale.weight / .bias (1), te_periodic.weight /
 * .bias (E - 1)); E <= 16 */
int immtsf_tpatchgnn_decoder_forward_te(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, int32_t precision, const float* h,
                                        const float* t, const immtsf_time2vec_params* tp, const immtsf_decoder_params* p, float* out,
                                        immtsf_stream_t stream) {
    if (!t || !tp || !tp->w0 || !tp->b0 || E < 1 || E > 16 || (E > 1 && (!tp->w || !tp->b))) return IMMTSF_EINVAL;
    return dec_forward(B, N, Lp, D, E, H, precision, h, nullptr, DecTE{t, tp->w0, tp->b0, tp->w, tp->b, nullptr, nullptr, nullptr, nullptr}, p, out,
                       stream);
}

int immtsf_tpatchgnn_decoder_backward(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, const float* h,
                                      const float* te, const immtsf_decoder_params* p, const float* dout, float* dh,
                                      float* dte, const immtsf_decoder_params* grads, immtsf_stream_t stream) {
    return immtsf_tpatchgnn_decoder_backward_p(B, N, Lp, D, E, H, 0, h, te, p, dout, dh, dte, grads, stream);
}

static int dec_backward(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, int32_t precision, const float* h,
                        const float* te, const DecTE& tq, const immtsf_decoder_params* p, const float* dout, float* dh,
                        float* dte, const immtsf_decoder_params* grads, immtsf_stream_t stream) {
    if (!dims_ok(B, N, Lp, D, E, H) || !h || (!tq.t && (!te || !dte)) || !p || !dout || !dh || !grads || precision < 0 || precision > 1)
        return IMMTSF_EINVAL;
    if (immtsf_tpatchgnn_decoder_lds_bytes(N, Lp, D, E, H) == 0) return IMMTSF_EUNSUPPORTED;
    if (B == 0) return IMMTSF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DecDims d{B, N, Lp, D, E};
    const DecP q{p->W1, p->b1, p->W2, p->b2, p->W3, p->b3};
    const DecG gq{grads->W1, grads->b1, grads->W2, grads->b2, grads->W3, grads->b3};
    if (precision == 1 && dec_mfma_ok(p)) {
        const bool slabs = Lp > 12 && bwd_lds_mfma<32>(N, Lp, D, E, true) <= kLdsMax;      // else: LDS atomics on the one dv image
        const size_t lm = bwd_lds_mfma<32>(N, Lp, D, E, slabs);
        int per = (int)(160 * 1024 / lm);
        per = per < 1 ? 1 : per > 2 ? 2 : per;          // (the grid is the fan-in of the final atomics)
        const int grid = B < 256 * per ? B : 256 * per;
        hipLaunchKernelGGL(dec_bwd_mfma_kernel, dim3(grid), dim3(512), lm, s, d, q, h, te, dout, dh, dte, gq, slabs ? 1 : 0, tq);
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    const size_t lds = bwd_lds<32>(N, Lp, D, E);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_bwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int per_cu = (int)(160 * 1024 / lds) > 0 ? (int)(160 * 1024 / lds) : 1;      // resident workgroups per CU by LDS
    const int grid = B < 256 * per_cu ? B : 256 * per_cu;
    hipLaunchKernelGGL(dec_bwd_kernel<32>, dim3(grid), dim3(256), lds, s, d, q, h, te, dout, dh, dte, gq, tq);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int immtsf_tpatchgnn_decoder_backward_p(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, int32_t precision, const float* h,
                                        const float* te, const immtsf_decoder_params* p, const float* dout, float* dh,
                                        float* dte, const immtsf_decoder_params* grads, immtsf_stream_t stream) {
    return dec_backward(B, N, Lp, D, E, H, precision, h, te, DecTE{}, p, dout, dh, dte, grads, stream);
}
/* ... and its backward: the four LearnableTE gradients are ACCUMULATED (atomics) into tgrads like the decoder's own into `grads` */
int immtsf_tpatchgnn_decoder_backward_te(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, int32_t precision, const float* h,
                                         const float* t, const immtsf_time2vec_params* tp, const immtsf_decoder_params* p, const float* dout,
                                         float* dh, const immtsf_decoder_params* grads, const immtsf_time2vec_params* tgrads,
                                         immtsf_stream_t stream) {
    if (!t || !tp || !tgrads || !tp->w0 || !tp->b0 || !tgrads->w0 || !tgrads->b0 || E < 1 || E > 16 ||
        (E > 1 && (!tp->w || !tp->b || !tgrads->w || !tgrads->b)))
        return IMMTSF_EINVAL;
    return dec_backward(B, N, Lp, D, E, H, precision, h, nullptr, DecTE{t, tp->w0, tp->b0, tp->w, tp->b, tgrads->w0, tgrads->b0, tgrads->w, tgrads->b},
                        p, dout, dh, nullptr, grads, stream);
}

}  // extern "C"
