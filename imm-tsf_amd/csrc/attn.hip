// Ragged single-query cross-attention (TTF_T2V_XAttn) and row softmax (+attention-weight dropout) kernels.
//
// TTF_T2V_XAttn's query is one learned vector (fusions/TTF_T2V_XAttn.py:91,143), so per window b and head h the
// scores are a mat-vec of the window's packed keys with the scaled query, the softmax runs over that window's
// n_b notes only (offsets[b]..offsets[b+1]) -- no -inf padding, no T-fold copy of K/V (the reference materialises
// T copies, :150-159) -- and only the attention-weight dropout makes the T output rows differ.
#include "attn.hpp"

namespace {

constexpr int TT = 32;   // forecast steps accumulated per pass in registers

// grid (B, H, ceil(hd/256)), 256 threads; LDS: sc[N] | atile[TT*64] | red[16]
__global__ __launch_bounds__(256) void ragged_attn_fwd_kernel(RaggedAttnDims dm, const int* __restrict__ offsets,
                                                               const int* __restrict__ rowmap,
                                                               const float* __restrict__ KVp, const float* __restrict__ qs,
                                                               float* __restrict__ P, float* __restrict__ ctx, DropCfg drop,
                                                               uint64_t site) {
    extern __shared__ float lds[];
    float* sc = lds;
    float* atile = lds + dm.N;
    float* red = atile + TT * 64;
    const int b = blockIdx.x, h = blockIdx.y, ez = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hd = dm.hd, d = dm.H * hd, ld = 2 * d, T = dm.T;
    const int o0 = offsets[b], n = offsets[b + 1] - o0;
    const int e = ez * 256 + tid;
    const bool valid = e < hd;
    if (n == 0) {   // no notes: the caller zeroes the row after out_proj anyway (M_txt); keep ctx defined
        if (valid) for (int t = 0; t < T; ++t) ctx[(size_t)(b * T + t) * d + h * hd + e] = 0.f;
        return;
    }
    for (int i = wave; i < n; i += 4) {
        const float* kr = KVp + (size_t)(o0 + i) * ld + h * hd;
        float a = 0.f;
#pragma unroll 4
        for (int c = lane; c < hd; c += 64) a = fmaf(qs[h * hd + c], kr[c], a);
        a = wave_sum(a);
        if (lane == 0) sc[i] = a;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int i = tid; i < n; i += 256) m = fmaxf(m, sc[i]);
    m = block_max(m, red);
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) { const float p = expf(sc[i] - m); sc[i] = p; sum += p; }
    sum = block_sum(sum, red);
    const float inv = 1.f / sum;
    for (int i = tid; i < n; i += 256) {
        const float p = sc[i] * inv;
        sc[i] = p;
        if (ez == 0) P[(size_t)(o0 + i) * dm.H + h] = p;
    }
    __syncthreads();

    const float* vbase = KVp + (size_t)o0 * ld + d + h * hd + e;
    if (drop.p <= 0.f) {   // every forecast step sees the same weights
        if (!valid) return;
        float acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < n; ++i) acc = fmaf(sc[i], vbase[(size_t)i * ld], acc);
        for (int t = 0; t < T; ++t) ctx[(size_t)(b * T + t) * d + h * hd + e] = acc;
        return;
    }
    for (int t0 = 0; t0 < T; t0 += TT) {
        float acc[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[tt] = 0.f;
        for (int i0 = 0; i0 < n; i0 += 64) {
            __syncthreads();
            for (int x = tid; x < TT * 64; x += 256) {
                const int tt = x >> 6, ii = x & 63, t = t0 + tt, i = i0 + ii;
                float a = 0.f;
                if (t < T && i < n) {
                    const int n_orig = rowmap[o0 + i] - b * dm.N;
                    const uint64_t idx = ((uint64_t)(b * T + t) * dm.H + h) * dm.N + n_orig;
                    a = sc[i] * dropout_scale(drop, site, idx);
                }
                atile[x] = a;
            }
            __syncthreads();
            if (valid) {
                const int cnt = min(64, n - i0);
                for (int ii = 0; ii < cnt; ii += 4) {     // four notes per step: their V loads are in flight together
                    float v[4];                           // (one load -> 32 FMAs -> next load serialises on latency)
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = (ii + u < cnt) ? vbase[(size_t)(i0 + ii + u) * ld] : 0.f;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int tt = 0; tt < TT; ++tt) acc[tt] = fmaf(atile[tt * 64 + ii + u], v[u], acc[tt]);   // tile is 0 past n
                }
            }
        }
        if (valid) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
                if (t0 + tt < T) ctx[(size_t)(b * T + t0 + tt) * d + h * hd + e] = acc[tt];
        }
    }
}

constexpr int MT = 4;   // backward keeps the dropout scales of up to MT*64 forecast steps in registers

// Backward, part 1.  grid (B, H, ceil(hd/64)), 256 threads: every workgroup owns 64 head columns of one window,
// its 4 waves take the window's notes round-robin, one column per lane:
//   g[c]      = sum_t m[t,i] * dctx[b,t,h,c]            (m = dropout scale of the attention weight, 1 if p = 0)
//   dv[i,c]   = p[i] * g[c]
//   dp[i]    += sum_c g[c] * v[i,c]                     (fp32 atomic: hd/64 partial sums per note)
__global__ __launch_bounds__(256) void ragged_attn_bwd_dv_kernel(RaggedAttnDims dm, const int* __restrict__ offsets,
                                                                  const int* __restrict__ rowmap,
                                                                  const float* __restrict__ KVp, const float* __restrict__ P,
                                                                  const float* __restrict__ dctx, float* __restrict__ dKVp,
                                                                  float* __restrict__ dp_buf, DropCfg drop, uint64_t site) {
    const int b = blockIdx.x, h = blockIdx.y, c = blockIdx.z * 64 + (threadIdx.x & 63);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hd = dm.hd, d = dm.H * hd, ld = 2 * d, T = dm.T;
    const int o0 = offsets[b], n = offsets[b + 1] - o0;
    if (n == 0) return;
    const bool valid = c < hd;
    const float* dc = dctx + (size_t)b * T * d + h * hd + c;   // row t at dc + t*d
    float gsum = 0.f;
    if (drop.p <= 0.f && valid)
        for (int t = 0; t < T; ++t) gsum += dc[(size_t)t * d];
    // the upstream gradient column is the same for every note of the window: up to DCR forecast steps stay in registers
    // (re-reading them per note made every note wait on T dependent loads)
    constexpr int DCR = 32;
    float dcv[DCR];
#pragma unroll
    for (int t = 0; t < DCR; ++t) dcv[t] = (valid && drop.p > 0.f && t < T) ? dc[(size_t)t * d] : 0.f;
    for (int i = wave; i < n; i += 4) {
        float g = gsum;
        if (drop.p > 0.f) {
            const int n_orig = rowmap[o0 + i] - b * dm.N;
            float mreg[MT];
#pragma unroll
            for (int k = 0; k < MT; ++k) {
                const int t = k * 64 + lane;
                mreg[k] = 0.f;
                if (t < T) {
                    const uint64_t idx = ((uint64_t)(b * T + t) * dm.H + h) * dm.N + n_orig;
                    mreg[k] = dropout_scale(drop, site, idx);
                }
            }
            g = 0.f;
#pragma unroll
            for (int tt = 0; tt < DCR; ++tt) g = fmaf(__shfl(mreg[0], tt, 64), dcv[tt], g);      // steps 0..DCR-1 (dcv is 0 past T)
#pragma unroll
            for (int k = 0; k < MT; ++k) {
                const int tcnt = min(64, T - k * 64);
                for (int tt = (k == 0 ? DCR : 0); tt < tcnt; ++tt) {
                    const float mt = __shfl(mreg[k], tt, 64);
                    if (valid) g = fmaf(mt, dc[(size_t)(k * 64 + tt) * d], g);
                }
            }
        }
        float a = 0.f;
        if (valid) {
            const size_t off = (size_t)(o0 + i) * ld + d + h * hd + c;
            a = g * KVp[off];
            dKVp[off] = P[(size_t)(o0 + i) * dm.H + h] * g;
        }
        a = wave_sum(a);
        if (lane == 0) atomicAdd(dp_buf + (size_t)(o0 + i) * dm.H + h, a);
    }
}

// Backward, part 2.  grid (B, H), 256 threads; LDS: ds[N] | red[16].
//   ds[i] = p[i] (dp[i] - sum_j p[j] dp[j]);  dk[i,:] = ds[i] * qs_h;  dqs_part[b, h, :] = sum_i ds[i] k[i,:]
__global__ __launch_bounds__(256) void ragged_attn_bwd_ds_kernel(RaggedAttnDims dm, const int* __restrict__ offsets,
                                                                  const float* __restrict__ KVp, const float* __restrict__ qs,
                                                                  const float* __restrict__ P, const float* __restrict__ dp_buf,
                                                                  float* __restrict__ dKVp, float* __restrict__ dqs_part) {
    extern __shared__ float lds[];
    float* ds = lds;
    float* red = lds + dm.N;
    const int b = blockIdx.x, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hd = dm.hd, d = dm.H * hd, ld = 2 * d;
    const int o0 = offsets[b], n = offsets[b + 1] - o0;
    if (n == 0) {
        for (int c = tid; c < hd; c += 256) dqs_part[(size_t)b * d + h * hd + c] = 0.f;
        return;
    }
    float part = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float p = P[(size_t)(o0 + i) * dm.H + h], g = dp_buf[(size_t)(o0 + i) * dm.H + h];
        ds[i] = g;
        part = fmaf(p, g, part);
    }
    const float dot = block_sum(part, red);
    __syncthreads();
    for (int i = tid; i < n; i += 256) ds[i] = P[(size_t)(o0 + i) * dm.H + h] * (ds[i] - dot);
    __syncthreads();
    for (int i = wave; i < n; i += 4) {
        float* dkr = dKVp + (size_t)(o0 + i) * ld + h * hd;
        const float dsi = ds[i];
        for (int c = lane; c < hd; c += 64) dkr[c] = dsi * qs[h * hd + c];
    }
    for (int c = tid; c < hd; c += 256) {
        const float* kc = KVp + (size_t)o0 * ld + h * hd + c;
        float a = 0.f;
#pragma unroll 8
        for (int i = 0; i < n; ++i) a = fmaf(ds[i], kc[(size_t)i * ld], a);
        dqs_part[(size_t)b * d + h * hd + c] = a;
    }
}

// ---- dense attention rows: one wave per (b,h,l) row of length S ---------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(float* __restrict__ sc, float* __restrict__ A, int rows, int HL,
                                                                int S, const unsigned char* __restrict__ live, DropCfg drop,
                                                                uint64_t site, int causal_L) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = sc + (size_t)row * S;
    float* a = A + (size_t)row * S;
    if (live && !live[row / HL]) {
        for (int i = lane; i < S; i += 64) { p[i] = 0.f; a[i] = 0.f; }
        return;
    }
    // causal_L > 0: TriangularCausalMask (utils/masking.py): query l = row % L may only see keys s <= l
    const int Sv = causal_L > 0 ? min(S, (row % causal_L) + 1) : S;
    float m = -INFINITY;
    for (int i = lane; i < Sv; i += 64) m = fmaxf(m, p[i]);
    m = wave_max(m);
    float sum = 0.f;
    for (int i = lane; i < Sv; i += 64) sum += expf(p[i] - m);
    const float inv = 1.f / wave_sum(sum);
    for (int i = lane; i < S; i += 64) {
        const float v = (i < Sv) ? expf(p[i] - m) * inv : 0.f;
        p[i] = v;
        a[i] = v * dropout_scale(drop, site, (uint64_t)row * S + i);
    }
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(float* __restrict__ dA, const float* __restrict__ P, int rows,
                                                                int S, DropCfg drop, uint64_t site) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* g = dA + (size_t)row * S;
    const float* p = P + (size_t)row * S;
    float dot = 0.f;
    for (int i = lane; i < S; i += 64) {
        const float dpv = g[i] * dropout_scale(drop, site, (uint64_t)row * S + i);
        g[i] = dpv;
        dot = fmaf(p[i], dpv, dot);
    }
    dot = wave_sum(dot);
    for (int i = lane; i < S; i += 64) g[i] = p[i] * (g[i] - dot);
}

}  // namespace

int launch_ragged_attn_fwd(RaggedAttnDims dm, const int* offsets, const int* rowmap, const float* KVp, const float* qs,
                           float* P, float* ctx, DropCfg drop, uint64_t site, hipStream_t s) {
    if (dm.B <= 0) return IMMTSF_OK;
    const size_t lds = (size_t)(dm.N + TT * 64 + 16) * sizeof(float);
    if (lds > 160 * 1024) return IMMTSF_EUNSUPPORTED;
    hipLaunchKernelGGL(ragged_attn_fwd_kernel, dim3(dm.B, dm.H, cdiv(dm.hd, 256)), dim3(256), lds, s, dm, offsets, rowmap, KVp, qs,
                       P, ctx, drop, site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_ragged_attn_bwd(RaggedAttnDims dm, const int* offsets, const int* rowmap, const float* KVp, const float* qs,
                           const float* P, const float* dctx, float* dKVp, float* dqs_part, float* dp_buf, DropCfg drop,
                           uint64_t site, hipStream_t s) {
    if (dm.B <= 0) return IMMTSF_OK;
    if (drop.p > 0.f && dm.T > MT * 64) return IMMTSF_EUNSUPPORTED;
    const size_t lds = (size_t)(dm.N + 16) * sizeof(float);
    if (lds > 160 * 1024) return IMMTSF_EUNSUPPORTED;
    hipError_t e = hipMemsetAsync(dp_buf, 0, (size_t)dm.B * dm.N * dm.H * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(ragged_attn_bwd_dv_kernel, dim3(dm.B, dm.H, cdiv(dm.hd, 64)), dim3(256), 0, s, dm, offsets, rowmap, KVp, P,
                       dctx, dKVp, dp_buf, drop, site);
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL(ragged_attn_bwd_ds_kernel, dim3(dm.B, dm.H), dim3(256), lds, s, dm, offsets, KVp, qs, P, dp_buf, dKVp,
                       dqs_part);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_softmax_rows_fwd(float* sc, float* A, int B, int H, int L, int S, const unsigned char* live, DropCfg drop,
                            uint64_t site, int causal, hipStream_t s) {
    const int rows = B * H * L;
    if (rows <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, sc, A, rows, H * L, S, live, drop, site,
                       causal ? L : 0);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_softmax_rows_bwd(float* dA, const float* P, int B, int H, int L, int S, DropCfg drop, uint64_t site,
                            hipStream_t s) {
    const int rows = B * H * L;
    if (rows <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, dA, P, rows, S, drop, site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
