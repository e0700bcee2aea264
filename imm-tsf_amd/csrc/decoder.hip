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

namespace {

struct DecDims { int B, N, Lp, D, E; };
struct DecP { const float *W1, *b1, *W2, *b2, *W3, *b3; };
struct DecG { float *W1, *b1, *W2, *b2, *W3, *b3; };

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
inline size_t stage_floats(int H, int N, int Lp, int D, int E) { return (size_t)H * (D + E + 1) + (size_t)N * D + (size_t)Lp * E; }
template <int H>
size_t fwd_lds(int N, int Lp, int D, int E) {
    return (size_t)(H * H + 3 * H + (N + Lp) * Geo<H>::P1 + stage_floats(H, N, Lp, D, E)) * sizeof(float);
}
template <int H>
size_t bwd_lds(int N, int Lp, int D, int E) {       // ... + the workgroup's running dW1 / db1 (it walks several windows)
    return (size_t)(H * H + 3 * H + 2 * (N + Lp) * Geo<H>::P1 + 2 * 256 * Geo<H>::PR + 2 * H + 8 + H * (D + E) + H) * sizeof(float);
}

// coalesced copies: W1s[k][D+E+1], hs[n][D], tes[lp][E] (contiguous, in this order, at `st`)
template <int H>
__device__ __forceinline__ void stage(const DecDims& d, const DecP& p, const float* __restrict__ h, const float* __restrict__ te, int b,
                                      float* st) {
    const int ld = d.D + d.E, pw = ld + 1;
    float* hs = st + H * pw;
    float* tes = hs + d.N * d.D;
    for (int i = threadIdx.x; i < H * ld; i += 256) st[(i / ld) * pw + i % ld] = p.W1[i];
    for (int i = threadIdx.x; i < d.N * d.D; i += 256) hs[i] = h[(size_t)b * d.N * d.D + i];
    for (int i = threadIdx.x; i < d.Lp * d.E; i += 256) tes[i] = te[(size_t)b * d.Lp * d.E + i];
}
// u[n][k] = b1[k] + sum_d W1[k][d] h[b, n, d];  v[lp][k] = sum_e W1[k][D + e] te[b, lp, e]   (operands staged by stage())
template <int H>
__device__ __forceinline__ void first_layer(const DecDims& d, const float* __restrict__ b1, const float* st, float* u, float* v) {
    constexpr int P1 = Geo<H>::P1;
    const int pw = d.D + d.E + 1;
    const float* hs = st + H * pw;
    const float* tes = hs + d.N * d.D;
    for (int i = threadIdx.x; i < (d.N + d.Lp) * H; i += 256) {
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
                                                       float* __restrict__ out) {
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
    stage<H>(d, p, h, te, b, st);
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
                                                       float* __restrict__ dte, DecG g) {
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
    const int tid = threadIdx.x;
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
    stage<H>(d, p, h, te, b, Xs);       // the staged operands borrow the first chunk image (launcher checks they fit)
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
    stage<H>(d, p, h, te, b, Xs);       // the chunk images are dead: W1 / h / te again for the first layer's gradients
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
        dte[(size_t)b * d.Lp * d.E + i] = a;
    }
    __syncthreads();        // the next window restages over Xs and clears du / dv
    }   // windows

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
    if (!dims_ok(B, N, Lp, D, E, H) || !h || !te || !p || !out) return IMMTSF_EINVAL;
    if (immtsf_tpatchgnn_decoder_lds_bytes(N, Lp, D, E, H) == 0) return IMMTSF_EUNSUPPORTED;
    if (B == 0) return IMMTSF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DecDims d{B, N, Lp, D, E};
    const DecP q{p->W1, p->b1, p->W2, p->b2, p->W3, p->b3};
    const size_t lds = fwd_lds<32>(N, Lp, D, E);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_fwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dec_fwd_kernel<32>, dim3(B), dim3(256), lds, s, d, q, h, te, out);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int immtsf_tpatchgnn_decoder_backward(int32_t B, int32_t N, int32_t Lp, int32_t D, int32_t E, int32_t H, const float* h,
                                      const float* te, const immtsf_decoder_params* p, const float* dout, float* dh,
                                      float* dte, const immtsf_decoder_params* grads, immtsf_stream_t stream) {
    if (!dims_ok(B, N, Lp, D, E, H) || !h || !te || !p || !dout || !dh || !dte || !grads) return IMMTSF_EINVAL;
    if (immtsf_tpatchgnn_decoder_lds_bytes(N, Lp, D, E, H) == 0) return IMMTSF_EUNSUPPORTED;
    if (B == 0) return IMMTSF_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DecDims d{B, N, Lp, D, E};
    const DecP q{p->W1, p->b1, p->W2, p->b2, p->W3, p->b3};
    const DecG gq{grads->W1, grads->b1, grads->W2, grads->b2, grads->W3, grads->b3};
    const size_t lds = bwd_lds<32>(N, Lp, D, E);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_bwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int per_cu = (int)(160 * 1024 / lds) > 0 ? (int)(160 * 1024 / lds) : 1;      // resident workgroups per CU by LDS
    const int grid = B < 256 * per_cu ? B : 256 * per_cu;
    hipLaunchKernelGGL(dec_bwd_kernel<32>, dim3(grid), dim3(256), lds, s, d, q, h, te, dout, dh, dte, gq);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

}  // extern "C"
