// HBM-bound row kernels: ragged index (note mask -> lengths/offsets/rowmap), Time2Vec, LayerNorm(+dropout),
// column reductions and a few tiny vector helpers.  64-lane wavefronts: a "row" is normally owned by one wave,
// loads are 16 B per lane whenever the row is 16-byte aligned.
#include "rowops.hpp"

namespace {

constexpr int kSlabs = 32;   // row slabs of the two-stage column reductions (deterministic, no atomics)

// ------------------------------------------------------------------------------------------- note mask
// reference: note_mask = (V.abs().sum(dim=2) > 0)  fusions/TTF_T2V_XAttn.py:107, fusions/TTF_RecAvg.py:69
__global__ __launch_bounds__(256) void note_mask_kernel(const float* __restrict__ V, int rows, int d_m,
                                                         unsigned char* __restrict__ mask, int* nan_flag, int vec) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = V + (size_t)row * d_m;
    float s = 0.f;
    bool bad = false;
    if (vec) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (int i = lane; i < d_m / 4; i += 64) {
            const float4 v = p4[i];
            s += fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w);
            bad |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
        }
    } else {
        for (int i = lane; i < d_m; i += 64) {
            const float v = p[i];
            s += fabsf(v);
            bad |= (v != v);
        }
    }
    s = wave_sum(s);
    const bool anybad = __any(bad);
    if (lane == 0) {
        mask[row] = (s > 0.f) ? 1 : 0;   // NaN sum compares false, like torch
        if (anybad && nan_flag) atomicOr(nan_flag, 1);
    }
}

// one block (1024 threads = 16 waves).  Phase 1: a wave per window counts its notes (ballot/popcount);
// phase 2: block scan of the lengths; phase 3: a wave per window writes the compacted row indices.
__global__ __launch_bounds__(256) void mask_from_lengths_kernel(const int* __restrict__ lengths, int B, int N,
                                                                unsigned char* __restrict__ mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B * N) mask[i] = (i % N) < lengths[i / N] ? 1 : 0;
}

__global__ __launch_bounds__(1024) void ragged_index_kernel(const unsigned char* __restrict__ mask, int B, int N,
                                                            int* __restrict__ lengths, int* __restrict__ offsets,
                                                            int* __restrict__ rowmap, int* __restrict__ seg,
                                                            unsigned char* __restrict__ mtxt, unsigned char* __restrict__ mtxt2) {
    __shared__ int sh[1024];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = wave; b < B; b += 16) {
        int cnt = 0;
        for (int n0 = 0; n0 < N; n0 += 64) {
            const int n = n0 + lane;
            const bool m = (n < N) && mask[(size_t)b * N + n] != 0;
            cnt += __popcll(__ballot(m));
        }
        if (lane == 0) { lengths[b] = cnt; mtxt[b] = cnt > 0 ? 1 : 0; if (mtxt2) mtxt2[b] = cnt > 0 ? 1 : 0; }
    }
    if (tid == 0) carry = 0;
    __syncthreads();   // lengths[] written by this block are visible after the barrier (same CU)
    for (int base = 0; base < B; base += 1024) {
        const int b = base + tid;
        const int len = (b < B) ? lengths[b] : 0;
        sh[tid] = len;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int v = (tid >= off) ? sh[tid - off] : 0;
            __syncthreads();
            sh[tid] += v;
            __syncthreads();
        }
        if (b < B) offsets[b] = carry + sh[tid] - len;
        __syncthreads();
        if (tid == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (tid == 0) offsets[B] = carry;
    __syncthreads();
    for (int b = wave; b < B; b += 16) {
        int pos = offsets[b];
        for (int n0 = 0; n0 < N; n0 += 64) {
            const int n = n0 + lane;
            const bool m = (n < N) && mask[(size_t)b * N + n] != 0;
            const unsigned long long bal = __ballot(m);
            if (m) {
                const int k = pos + __popcll(bal & ((1ull << lane) - 1ull));
                rowmap[k] = b * N + n;
                seg[k] = b;
            }
            pos += __popcll(bal);
        }
    }
}

// many windows (B > 256): the same three phases as three launches -- a wave per window over the whole chip for the counting and the
// compaction, one workgroup only for the scan of the B lengths (the one-workgroup kernel above walks B * N mask bytes alone:
// 0.36 ms at 4096 windows).  Outputs identical.
// (256-thread workgroups: a 1024-thread one needs four free wave slots on EVERY SIMD of a CU at once and starved behind the other
// stream's TTCN forward, whose small workgroups keep refilling the slots -- 470 us for a 5 us kernel at 4096 windows)
__global__ __launch_bounds__(256) void ragged_count_kernel(const unsigned char* __restrict__ mask, int B, int N, int* __restrict__ lengths,
                                                           unsigned char* __restrict__ mtxt, unsigned char* __restrict__ mtxt2) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int cnt = 0;
    for (int n0 = 0; n0 < N; n0 += 64) {
        const int n = n0 + lane;
        const bool m = (n < N) && mask[(size_t)b * N + n] != 0;
        cnt += __popcll(__ballot(m));
    }
    if (lane == 0) { lengths[b] = cnt; mtxt[b] = cnt > 0 ? 1 : 0; if (mtxt2) mtxt2[b] = cnt > 0 ? 1 : 0; }
}
// exclusive scan of the B window lengths by ONE workgroup: a thread sums its `per` consecutive windows, the 64 lanes of a wave
// scan those sums by shuffles, the 16 waves meet in LDS -- one barrier (the first version walked 1024 windows per pass through a
// 10-step LDS scan with two barriers per step: 190 us at 4096 windows)
__global__ __launch_bounds__(1024) void ragged_scan_kernel(const int* __restrict__ lengths, int B, int* __restrict__ offsets) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (B + 1023) / 1024, b0 = tid * per;
    int loc = 0;
    for (int k = 0; k < per; ++k)
        if (b0 + k < B) loc += lengths[b0 + k];
    int v = loc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w_ = 0; w_ < 16; ++w_) {
        if (w_ < wave) base += wsum[w_];
        tot += wsum[w_];
    }
    int run = base + v - loc;
    for (int k = 0; k < per; ++k)
        if (b0 + k < B) { offsets[b0 + k] = run; run += lengths[b0 + k]; }
    if (tid == 0) offsets[B] = tot;
}
__global__ __launch_bounds__(256) void ragged_fill_kernel(const unsigned char* __restrict__ mask, int B, int N, const int* __restrict__ offsets,
                                                          int* __restrict__ rowmap, int* __restrict__ seg) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int pos = offsets[b];
    for (int n0 = 0; n0 < N; n0 += 64) {
        const int n = n0 + lane;
        const bool m = (n < N) && mask[(size_t)b * N + n] != 0;
        const unsigned long long bal = __ballot(m);
        if (m) {
            const int k = pos + __popcll(bal & ((1ull << lane) - 1ull));
            rowmap[k] = b * N + n;
            seg[k] = b;
        }
        pos += __popcll(bal);
    }
}

__device__ __forceinline__ void gather_rows_body(int r, const float* __restrict__ src, int ld_src,
                                                 const int* __restrict__ rowmap, const int* __restrict__ total,
                                                 int width, float* __restrict__ dst, int ld_dst,
                                                 bf16_t* __restrict__ dst_h, int vec) {
    if (r >= *total) return;
    const float* s = src + (size_t)rowmap[r] * ld_src;
    if (vec) {      // width, both pitches multiples of 4 and 16-byte aligned bases: 16-byte loads, 8-byte bf16 stores
        const float4* s4 = reinterpret_cast<const float4*>(s);
        for (int i = threadIdx.x; i < (width >> 2); i += blockDim.x) {
            const float4 v = s4[i];
            if (dst) reinterpret_cast<float4*>(dst + (size_t)r * ld_dst)[i] = v;
            if (dst_h) {
                bf16x4 h;
                h[0] = (bf16_t)v.x; h[1] = (bf16_t)v.y; h[2] = (bf16_t)v.z; h[3] = (bf16_t)v.w;
                reinterpret_cast<bf16x4*>(dst_h + (size_t)r * ld_dst)[i] = h;
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < width; i += blockDim.x) {
        const float v = s[i];
        if (dst) dst[(size_t)r * ld_dst + i] = v;
        if (dst_h) dst_h[(size_t)r * ld_dst + i] = (bf16_t)v;
    }
}
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int ld_src,
                                                           const int* __restrict__ rowmap, const int* __restrict__ total,
                                                           int width, float* __restrict__ dst, int ld_dst,
                                                           bf16_t* __restrict__ dst_h, int vec) {
    gather_rows_body(blockIdx.x, src, ld_src, rowmap, total, width, dst, ld_dst, dst_h, vec);
}

// ------------------------------------------------------------------------------------------- Time2Vec
// reference: fusions/TTF_T2V_XAttn.py:7-24  (applied to RAW tau, :136)
__device__ __forceinline__ void time2vec_fwd_body(int bid, const float* __restrict__ tau_pad, const int* __restrict__ rowmap,
                                                  const int* __restrict__ total, int d_tau,
                                                  const float* __restrict__ w0, const float* __restrict__ b0,
                                                  const float* __restrict__ w, const float* __restrict__ b,
                                                  float* __restrict__ dst, int ld_dst, int max_rows,
                                                  bf16_t* __restrict__ dst_h) {
    const long idx = (long)bid * 256 + threadIdx.x;
    const int r = (int)(idx / d_tau), j = (int)(idx % d_tau);
    if (r >= (total ? *total : max_rows)) return;
    const float t = tau_pad[rowmap ? rowmap[r] : r];
    const float v = (j == 0) ? fmaf(w0[0], t, b0[0]) : sinf(fmaf(w[j - 1], t, b[j - 1]));
    if (dst) dst[(size_t)r * ld_dst + j] = v;
    if (dst_h) dst_h[(size_t)r * ld_dst + j] = (bf16_t)v;
}
__global__ __launch_bounds__(256) void time2vec_fwd_kernel(const float* __restrict__ tau_pad, const int* __restrict__ rowmap,
                                                            const int* __restrict__ total, int d_tau,
                                                            const float* __restrict__ w0, const float* __restrict__ b0,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float* __restrict__ dst, int ld_dst, int max_rows,
                                                            bf16_t* __restrict__ dst_h) {
    time2vec_fwd_body(blockIdx.x, tau_pad, rowmap, total, d_tau, w0, b0, w, b, dst, ld_dst, max_rows, dst_h);
}

// partial[slab][0][j] = sum_r g*tau, partial[slab][1][j] = sum_r g,  g = dfeat[r,j] * (j ? cos(w tau + b) : 1)
__device__ __forceinline__ void time2vec_bwd_body(int bx, int slab, int nsl, float (*red)[4][64], const float* __restrict__ tau_pad,
                                                  const int* __restrict__ rowmap, const int* __restrict__ total, int d_tau,
                                                  const float* __restrict__ w, const float* __restrict__ b,
                                                  const float* __restrict__ dfeat, int ld,
                                                  float* __restrict__ partial, int max_rows) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = bx * 64 + tx;
    const int M = total ? *total : max_rows, rps = (M + nsl - 1) / nsl;
    const int r0 = slab * rps, r1 = min(M, r0 + rps);
    float aw = 0.f, ab = 0.f;
    if (j < d_tau) {
        const float wj = j ? w[j - 1] : 0.f, bj = j ? b[j - 1] : 0.f;
        int r = r0 + ty;
        for (; r + 12 < r1; r += 16) {      // four rows per pass: their (rowmap -> tau, dfeat) load chains are in flight together
            float t[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                t[u] = tau_pad[rowmap ? rowmap[r + 4 * u] : r + 4 * u];
                g[u] = dfeat[(size_t)(r + 4 * u) * ld + j];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j) g[u] *= cosf(fmaf(wj, t[u], bj));
                aw = fmaf(g[u], t[u], aw);
                ab += g[u];
            }
        }
        for (; r < r1; r += 4) {
            const float t = tau_pad[rowmap ? rowmap[r] : r];
            float g = dfeat[(size_t)r * ld + j];
            if (j) g *= cosf(fmaf(wj, t, bj));
            aw = fmaf(g, t, aw);
            ab += g;
        }
    }
    red[0][ty][tx] = aw;
    red[1][ty][tx] = ab;
    __syncthreads();
    if (ty == 0 && j < d_tau) {
        partial[((size_t)slab * 2 + 0) * d_tau + j] = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
        partial[((size_t)slab * 2 + 1) * d_tau + j] = red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
    }
}
__global__ __launch_bounds__(256) void time2vec_bwd_kernel(const float* __restrict__ tau_pad, const int* __restrict__ rowmap,
                                                            const int* __restrict__ total, int d_tau,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ dfeat, int ld,
                                                            float* __restrict__ partial, int max_rows) {
    __shared__ float red[2][4][64];
    time2vec_bwd_body(blockIdx.x, blockIdx.y, gridDim.y, red, tau_pad, rowmap, total, d_tau, w, b, dfeat, ld, partial, max_rows);
}

// narrow embeddings (d_tau <= 32: tPatchGNN's LearnableTE of the prediction times, 10 columns x 2048 rows) in ONE workgroup:
// CT column lanes x 1024 / CT row lanes, xor shuffles over the row lanes of a wave, one LDS round over the 16 waves
__global__ __launch_bounds__(1024) void time2vec_bwd_small_kernel(const float* __restrict__ tau, int rows, int d_tau, int CT,
                                                                   const float* __restrict__ w, const float* __restrict__ b,
                                                                   const float* __restrict__ dfeat, int ld, float* dw0, float* db0, float* dw,
                                                                   float* db, int accumulate) {
    __shared__ float ra[16 * 32], rb[16 * 32];
    const int RT = 1024 / CT, j = threadIdx.x % CT, ty = threadIdx.x / CT;
    float aw = 0.f, ab = 0.f;
    if (j < d_tau) {
        const float wj = j ? w[j - 1] : 0.f, bj = j ? b[j - 1] : 0.f;
        // 16 rows of loads in flight per thread: one workgroup does all the work, and at 4 the 2048 rows of the benchmark were eight
        // dependent global round trips (10 of the kernel's 15 us)
        constexpr int U = 16;
        for (int r0 = ty; r0 < rows; r0 += RT * U) {
            float tt[U], gg[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * RT;
                tt[u] = 0.f;
                gg[u] = 0.f;
                if (r < rows) { tt[u] = tau[r]; gg[u] = dfeat[(size_t)r * ld + j]; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float g = gg[u];
                if (j) g *= cosf(fmaf(wj, tt[u], bj));
                aw = fmaf(g, tt[u], aw);
                ab += g;
            }
        }
    }
    aw = coset_sum(aw, CT);
    ab = coset_sum(ab, CT);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < CT) { ra[wave * CT + lane] = aw; rb[wave * CT + lane] = ab; }
    __syncthreads();
    if (threadIdx.x < d_tau) {
        float sw = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { sw += ra[k * CT + threadIdx.x]; sb += rb[k * CT + threadIdx.x]; }
        const int jj = threadIdx.x;
        if (jj == 0) { dw0[0] = accumulate ? dw0[0] + sw : sw; db0[0] = accumulate ? db0[0] + sb : sb; }
        else { dw[jj - 1] = accumulate ? dw[jj - 1] + sw : sw; db[jj - 1] = accumulate ? db[jj - 1] + sb : sb; }
    }
}

// one wave per feature column: lanes stride over the slabs, shuffle-reduce
__global__ __launch_bounds__(256) void time2vec_bwd_final_kernel(const float* __restrict__ partial, int d_tau,
                                                                  float* dw0, float* db0, float* dw, float* db, int nsl,
                                                                  int accumulate) {
    const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= d_tau) return;
    float aw = 0.f, ab = 0.f;
    for (int s = lane; s < nsl; s += 64) {
        aw += partial[((size_t)s * 2 + 0) * d_tau + j];
        ab += partial[((size_t)s * 2 + 1) * d_tau + j];
    }
    aw = wave_sum(aw);
    ab = wave_sum(ab);
    if (lane == 0) {        // accumulate: the parameters have a second user this step whose backward adds into the same buffers
        if (j == 0) { dw0[0] = accumulate ? dw0[0] + aw : aw; db0[0] = accumulate ? db0[0] + ab : ab; }
        else { dw[j - 1] = accumulate ? dw[j - 1] + aw : aw; db[j - 1] = accumulate ? db[j - 1] + ab : ab; }
    }
}

// ------------------------------------------------------------------------------------------- column sums
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                              int M, const int* __restrict__ dyn, int N, int ld,
                                                              float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx, slab = blockIdx.y;
    if (dyn) M = *dyn;
    const int rps = (M + kSlabs - 1) / kSlabs;
    const int r0 = slab * rps, r1 = min(M, r0 + rps);
    float a = 0.f;
    if (n < N) {
        if (Y) for (int r = r0 + ty; r < r1; r += 4) a = fmaf(X[(size_t)r * ld + n], Y[(size_t)r * ld + n], a);
        else   for (int r = r0 + ty; r < r1; r += 4) a += X[(size_t)r * ld + n];
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && n < N) partial[(size_t)slab * N + n] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

// both LayerNorm parameter gradients in one pass: partial[slab][0][n] = sum_r X*Y, partial[slab][1][n] = sum_r X
__global__ __launch_bounds__(256) void colsum2_partial_kernel(const float* __restrict__ X, const float* __restrict__ Y, int M,
                                                               int N, int ld, float* __restrict__ partial) {
    __shared__ float red[2][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx, slab = blockIdx.y;
    const int rps = (M + kSlabs - 1) / kSlabs;
    const int r0 = slab * rps, r1 = min(M, r0 + rps);
    float a = 0.f, b = 0.f;
    if (n < N)
        for (int r = r0 + ty; r < r1; r += 4) {
            const float x = X[(size_t)r * ld + n];
            a = fmaf(x, Y[(size_t)r * ld + n], a);
            b += x;
        }
    red[0][ty][tx] = a;
    red[1][ty][tx] = b;
    __syncthreads();
    if (ty == 0 && n < N) {
        partial[((size_t)slab * 2 + 0) * N + n] = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
        partial[((size_t)slab * 2 + 1) * N + n] = red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
    }
}

// LayerNorm backward's three column reductions in one pass over the rows -- sum X*Y (dgamma), sum X (dbeta), sum Z (the
// residual vector's gradient) -- which also zeroes Z's rows whose row_flag[r / flag_div] == 0 and writes Z's bf16 image
// (what the next GEMM reads): colsum2 + colsum + mask_rows = 5 launches otherwise.  partial[slab][3][N].
__global__ __launch_bounds__(256) void colsum3_partial_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                               float* __restrict__ Z, int M, int N, int ld, float* __restrict__ partial,
                                                               const unsigned char* __restrict__ row_flag, int flag_div,
                                                               bf16_t* __restrict__ Zh) {
    __shared__ float red[3][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx, slab = blockIdx.y, nsl = gridDim.y;
    const int rps = (M + nsl - 1) / nsl;
    const int r0 = slab * rps, r1 = min(M, r0 + rps);
    float a = 0.f, b = 0.f, c = 0.f;
    if (n < N)
        for (int r = r0 + ty; r < r1; r += 4) {
            const size_t o = (size_t)r * ld + n;
            const float x = X[o];
            float z = Z[o];
            a = fmaf(x, Y[o], a);
            b += x;
            c += z;
            if (row_flag && !row_flag[r / flag_div]) { z = 0.f; Z[o] = 0.f; }
            if (Zh) Zh[o] = (bf16_t)z;
        }
    red[0][ty][tx] = a;
    red[1][ty][tx] = b;
    red[2][ty][tx] = c;
    __syncthreads();
    if (ty == 0 && n < N)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            partial[((size_t)slab * 3 + k) * N + n] = red[k][0][tx] + red[k][1][tx] + red[k][2][tx] + red[k][3][tx];
}

// grid ceil(N / 64): 64 columns x 4 slab lanes per workgroup (a thread that walks all 64 slabs alone is a 192-load chain: 19 us)
__global__ __launch_bounds__(256) void colsum3_final_kernel(const float* __restrict__ partial, int N, float* __restrict__ out_xy,
                                                             float* __restrict__ out_x, float* __restrict__ out_z, int nsl) {
    __shared__ float red[3][4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, n = blockIdx.x * 64 + tx;
    float a = 0.f, b = 0.f, c = 0.f;
    if (n < N)
#pragma unroll 4
        for (int s = ty; s < nsl; s += 4) {
            a += partial[((size_t)s * 3 + 0) * N + n];
            b += partial[((size_t)s * 3 + 1) * N + n];
            c += partial[((size_t)s * 3 + 2) * N + n];
        }
    red[0][ty][tx] = a;
    red[1][ty][tx] = b;
    red[2][ty][tx] = c;
    __syncthreads();
    if (ty == 0 && n < N) {
        out_xy[n] = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
        out_x[n] = red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
        out_z[n] = red[2][0][tx] + red[2][1][tx] + red[2][2][tx] + red[2][3][tx];
    }
}

// narrow matrices (N <= 32 columns, e.g. LayerNorm over the C series variables): ONE workgroup, CT column lanes x
// 1024/CT row lanes, tree over the row lanes -- one launch instead of the two-stage pair (whose 64-wide column tiling
// would leave 7/8 of every wave idle)
__global__ __launch_bounds__(1024) void colsum2_small_kernel(const float* __restrict__ X, const float* __restrict__ Y, int M, int N,
                                                              int ld, int CT, float* __restrict__ out_xy, float* __restrict__ out_x) {
    __shared__ float ra[1024], rb[1024];
    const int RT = 1024 / CT, tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    float a = 0.f, b = 0.f;
    if (tx < N) {
#pragma unroll 8
        for (int r = ty; r < M; r += RT) {
            const float x = X[(size_t)r * ld + tx];
            a = fmaf(x, Y[(size_t)r * ld + tx], a);
            b += x;
        }
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    for (int st = RT >> 1; st > 0; st >>= 1) {
        if (ty < st) { ra[threadIdx.x] += ra[threadIdx.x + st * CT]; rb[threadIdx.x] += rb[threadIdx.x + st * CT]; }
        __syncthreads();
    }
    if (ty == 0 && tx < N) { out_xy[tx] = ra[tx]; out_x[tx] = rb[tx]; }
}

// the same for N % 4 == 0 with 16-byte row chunks: N / 4 column lanes x (1024 / (N/4)) row lanes -- a quarter of the loads
// per thread in flight as float4s (M = 1024, N = 32: 8 rows per thread instead of 32; 17 -> ~6 us)
__global__ __launch_bounds__(1024) void colsum2_small_vec_kernel(const float* __restrict__ X, const float* __restrict__ Y, int M, int N,
                                                                  int ld, int CT, float* __restrict__ out_xy, float* __restrict__ out_x) {
    __shared__ float4 ra[1024], rb[1024];
    const int RT = 1024 / CT, tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (tx * 4 < N) {
#pragma unroll 8
        for (int r = ty; r < M; r += RT) {
            const float4 x = *reinterpret_cast<const float4*>(X + (size_t)r * ld + tx * 4);
            const float4 y = *reinterpret_cast<const float4*>(Y + (size_t)r * ld + tx * 4);
            a.x = fmaf(x.x, y.x, a.x); a.y = fmaf(x.y, y.y, a.y); a.z = fmaf(x.z, y.z, a.z); a.w = fmaf(x.w, y.w, a.w);
            b.x += x.x; b.y += x.y; b.z += x.z; b.w += x.w;
        }
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    for (int st = RT >> 1; st > 0; st >>= 1) {
        if (ty < st) {
            const float4 p = ra[threadIdx.x + st * CT], q = rb[threadIdx.x + st * CT];
            float4& u = ra[threadIdx.x];
            float4& v = rb[threadIdx.x];
            u.x += p.x; u.y += p.y; u.z += p.z; u.w += p.w;
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        __syncthreads();
    }
    if (ty == 0 && tx * 4 < N) {
        *reinterpret_cast<float4*>(out_xy + tx * 4) = ra[tx];
        *reinterpret_cast<float4*>(out_x + tx * 4) = rb[tx];
    }
}

// one sum (of X, or of X*Y) over the rows of a narrow matrix (N <= 32) in ONE workgroup: row lanes sit CT lanes apart inside a
// wave (xor shuffles first), one LDS round over the 16 waves -- instead of the two-stage pair
__global__ __launch_bounds__(1024) void colsum_small_kernel(const float* __restrict__ X, const float* __restrict__ Y, int M, int N, int ld,
                                                             int CT, float* __restrict__ out, int accumulate) {
    __shared__ float red[16 * 32];
    const int RT = 1024 / CT, tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    float a = 0.f;
    if (tx < N) {
#pragma unroll 8
        for (int r = ty; r < M; r += RT) {
            const float x = X[(size_t)r * ld + tx];
            a = Y ? fmaf(x, Y[(size_t)r * ld + tx], a) : a + x;
        }
    }
    a = coset_sum(a, CT);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < CT) red[wave * CT + lane] = a;
    __syncthreads();
    if (threadIdx.x < N) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w * CT + threadIdx.x];
        out[threadIdx.x] = accumulate ? out[threadIdx.x] + t : t;
    }
}

// a LayerNorm's two parameter gradients (sums of X * Y and of X) and a bias gradient (sums of Z) over the same few rows in ONE launch of
// two workgroups -- the encoder layer's feed-forward half at <= 4096 rows of d_model 32: two single-workgroup launches on the
// backbone's dependent backward chain before
__global__ __launch_bounds__(1024) void colsum_small_pair_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                                  const float* __restrict__ Z, int M, int N, int ld, int CT,
                                                                  float* __restrict__ out_xy, float* __restrict__ out_x,
                                                                  float* __restrict__ out_z) {
    __shared__ float red[3][16 * 32];
    const int RT = 1024 / CT, tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    const bool first = blockIdx.x == 0;
    float a = 0.f, b = 0.f;
    if (tx < N) {
#pragma unroll 8
        for (int r = ty; r < M; r += RT) {
            if (first) {
                const float x = X[(size_t)r * ld + tx];
                a = fmaf(x, Y[(size_t)r * ld + tx], a);
                b += x;
            } else {
                a += Z[(size_t)r * ld + tx];
            }
        }
    }
    a = coset_sum(a, CT);
    b = coset_sum(b, CT);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < CT) { red[0][wave * CT + lane] = a; red[1][wave * CT + lane] = b; }
    __syncthreads();
    if (threadIdx.x < N) {
        float s = 0.f, t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { s += red[0][w * CT + threadIdx.x]; t += red[1][w * CT + threadIdx.x]; }
        if (first) { out_xy[threadIdx.x] = s; out_x[threadIdx.x] = t; }
        else out_z[threadIdx.x] = s;
    }
}

// the same three sums spread over the chip, for outputs that read ZERO (a trainer's pre-zeroed gradient sinks): a workgroup of 256
// threads takes 256 / CT rows per pass over its row slab and adds its N partial sums to the outputs with one atomic each -- the pair
// kernel above is two workgroups walking 1024 rows in 32 dependent steps, 18 us on the backbone's backward chain at the benchmark shape
__global__ __launch_bounds__(256) void colsum_small_atomic_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                                   const float* __restrict__ Z, int M, int N, int ld, int CT, int rows_per_wg,
                                                                   float* __restrict__ out_xy, float* __restrict__ out_x,
                                                                   float* __restrict__ out_z) {
    __shared__ float red[3][4 * 32];
    const int RT = 256 / CT, tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
    float a = 0.f, b = 0.f, c = 0.f;
    if (tx < N) {
#pragma unroll 4
        for (int r = r0 + ty; r < r1; r += RT) {
            const float x = X[(size_t)r * ld + tx];
            a = fmaf(x, Y[(size_t)r * ld + tx], a);
            b += x;
            c += Z[(size_t)r * ld + tx];
        }
    }
    a = coset_sum(a, CT);
    b = coset_sum(b, CT);
    c = coset_sum(c, CT);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < CT) { red[0][wave * CT + lane] = a; red[1][wave * CT + lane] = b; red[2][wave * CT + lane] = c; }
    __syncthreads();
    if ((int)threadIdx.x < N) {
        float s = 0.f, t = 0.f, u = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { s += red[0][w * CT + threadIdx.x]; t += red[1][w * CT + threadIdx.x]; u += red[2][w * CT + threadIdx.x]; }
        atomicAdd(out_xy + threadIdx.x, s);
        atomicAdd(out_x + threadIdx.x, t);
        atomicAdd(out_z + threadIdx.x, u);
    }
}

__global__ __launch_bounds__(256) void colsum2_final_kernel(const float* __restrict__ partial, int N, float* __restrict__ out_xy,
                                                             float* __restrict__ out_x) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float a = 0.f, b = 0.f;
    for (int s = 0; s < kSlabs; ++s) {
        a += partial[((size_t)s * 2 + 0) * N + n];
        b += partial[((size_t)s * 2 + 1) * N + n];
    }
    out_xy[n] = a;
    out_x[n] = b;
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int N, float* __restrict__ out,
                                                            int accumulate) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float a = 0.f;
    for (int s = 0; s < kSlabs; ++s) a += partial[(size_t)s * N + n];
    out[n] = accumulate ? out[n] + a : a;
}

// ---- many rows (M >= 4096, the >= 256-window regime): 16 bytes per lane instead of 4, CL column lanes x 256 / CL row lanes per
// workgroup (CL = min(64, N / 4 rounded up to a power of two): narrow matrices keep every lane busy), up to 256 row slabs so that
// the grid covers the chip.  partial[slab][K][N].  K sums: 1: X (* Y); 2: X * Y and X; 3: X * Y, X and Z (+ Z's row zeroing and
// bf16 image, as colsum3_partial_kernel).  The one-dword-per-lane kernels above read 131072 x 768 fp32 at ~1 TB/s.
constexpr int kSlabsBig = 256;
template <int K>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ Z,
                                                          int M, int N, int ld, float* __restrict__ partial, int cl_shift,
                                                          const unsigned char* __restrict__ row_flag, int flag_div, bf16_t* __restrict__ Zh) {
    __shared__ float4 red[K][256];
    const int CL = 1 << cl_shift, RL = 256 >> cl_shift;
    const int tx = threadIdx.x & (CL - 1), ty = threadIdx.x >> cl_shift;
    const int n = (blockIdx.x * CL + tx) * 4, slab = blockIdx.y, nsl = gridDim.y;
    const int rps = (M + nsl - 1) / nsl;
    const int r0 = slab * rps, r1 = min(M, r0 + rps);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
    if (n < N)
        for (int r = r0 + ty; r < r1; r += RL) {
            const size_t o = (size_t)r * ld + n;
            const float4 x = *reinterpret_cast<const float4*>(X + o);
            if (K >= 2 || Y) {
                const float4 y = Y ? *reinterpret_cast<const float4*>(Y + o) : make_float4(1.f, 1.f, 1.f, 1.f);
                a.x = fmaf(x.x, y.x, a.x); a.y = fmaf(x.y, y.y, a.y); a.z = fmaf(x.z, y.z, a.z); a.w = fmaf(x.w, y.w, a.w);
            } else {
                a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
            }
            if (K >= 2) { b.x += x.x; b.y += x.y; b.z += x.z; b.w += x.w; }
            if (K >= 3) {
                float4 z = *reinterpret_cast<const float4*>(Z + o);
                c.x += z.x; c.y += z.y; c.z += z.z; c.w += z.w;
                if (row_flag && !row_flag[r / flag_div]) {
                    z = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(Z + o) = z;
                }
                if (Zh) {
                    bf16x4 h;
                    h[0] = (bf16_t)z.x; h[1] = (bf16_t)z.y; h[2] = (bf16_t)z.z; h[3] = (bf16_t)z.w;
                    *reinterpret_cast<bf16x4*>(Zh + o) = h;
                }
            }
        }
    red[0][threadIdx.x] = a;
    if (K >= 2) red[1][threadIdx.x] = b;
    if (K >= 3) red[2][threadIdx.x] = c;
    __syncthreads();
    for (int off = RL >> 1; off > 0; off >>= 1) {
        if (ty < off) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float4 u = red[k][threadIdx.x];
                const float4 v = red[k][threadIdx.x + (off << cl_shift)];
                u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
                red[k][threadIdx.x] = u;
            }
        }
        __syncthreads();
    }
    if (ty == 0 && n < N) {
#pragma unroll
        for (int k = 0; k < K; ++k) *reinterpret_cast<float4*>(partial + ((size_t)slab * K + k) * N + n) = red[k][tx];
    }
}
// grid (ceil(N / 32), K): 32 columns x 8 slab lanes per workgroup, one of the K sums per grid row (a thread walks nsl / 8 slabs:
// with 64 columns x 4 lanes x all K sums per thread this was 104 us behind a 400 us first pass at 256 slabs)
template <int K>
__global__ __launch_bounds__(256) void colsum_vec_final_kernel(const float* __restrict__ partial, int N, int nsl, float* __restrict__ o0,
                                                                float* __restrict__ o1, float* __restrict__ o2, int accumulate) {
    __shared__ float red[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, n = blockIdx.x * 32 + tx, k = blockIdx.y;
    float acc = 0.f;
    if (n < N)
#pragma unroll 8
        for (int s = ty; s < nsl; s += 8) acc += partial[((size_t)s * K + k) * N + n];
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && n < N) {
        float* out = k == 0 ? o0 : k == 1 ? o1 : o2;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += red[q][tx];
        out[n] = accumulate ? out[n] + v : v;
    }
}
// the many-row path applies: vectorisable and worth the bigger grid
static bool colsum_vec_ok(const void* X, const void* Y, const void* Z, const void* Zh, int M, int N, int ld, bool big_scratch) {
    if (!big_scratch || M < 4096 || (N & 3) || (ld & 3)) return false;
    const uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(Z);
    return (al & 15) == 0 && (reinterpret_cast<uintptr_t>(Zh) & 7) == 0;
}
template <int K>
static int launch_colsum_vec(const float* X, const float* Y, float* Z, int M, int N, int ld, float* o0, float* o1, float* o2, int accumulate,
                             float* scratch, const unsigned char* row_flag, int flag_div, void* Zh, hipStream_t s) {
    int sh = 0;
    while ((1 << sh) < cdiv(N, 4) && sh < 6) ++sh;
    const int CL = 1 << sh, RL = 256 / CL;
    int nsl = M / (RL * 8);                      // >= 8 rows per row lane
    nsl = nsl < 32 ? 32 : (nsl > kSlabsBig ? kSlabsBig : nsl);
    hipLaunchKernelGGL((colsum_vec_kernel<K>), dim3(cdiv(N, CL * 4), nsl), dim3(256), 0, s, X, Y, Z, M, N, ld, scratch, sh, row_flag,
                       flag_div > 0 ? flag_div : 1, static_cast<bf16_t*>(Zh));
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL((colsum_vec_final_kernel<K>), dim3(cdiv(N, 32), K), dim3(256), 0, s, scratch, N, nsl, o0, o1, o2, accumulate);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// ------------------------------------------------------------------------------------------- LayerNorm
// one wave per row; two-pass statistics like torch (mean, then mean of squared deviations), eps inside sqrt.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int rows, int d,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, float* __restrict__ xhat, float* __restrict__ rstd,
                                                             float* __restrict__ z, DropCfg drop, uint64_t site,
                                                             bf16_t* __restrict__ zh) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + (size_t)row * d;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) s += p[i];
    const float mu = wave_sum(s) / (float)d;
    float v = 0.f;
    for (int i = lane; i < d; i += 64) { const float c = p[i] - mu; v = fmaf(c, c, v); }
    const float rs = 1.0f / sqrtf(wave_sum(v) / (float)d + eps);
    if (lane == 0 && rstd) rstd[row] = rs;
    for (int i = lane; i < d; i += 64) {
        const float h = (p[i] - mu) * rs;
        if (xhat) xhat[(size_t)row * d + i] = h;
        const float y = fmaf(h, gamma[i], beta[i]);
        const float zv = y * dropout_scale(drop, site, (uint64_t)row * d + i);
        if (z) z[(size_t)row * d + i] = zv;
        if (zh) zh[(size_t)row * d + i] = (bf16_t)zv;
    }
}

// d % 4 == 0 and d <= 1024: the row lives in registers as float4 chunks (one read of x instead of three), 16-byte
// accesses, and one Philox call covers the four dropout bits of a chunk (the scalar kernels call it per element: at
// 2048 x 768 that is 1.5 M calls, a third of the kernel)
constexpr int LN_DV = 4;
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const float* __restrict__ x, int rows, int d,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float eps, float* __restrict__ xhat, float* __restrict__ rstd,
                                                                 float* __restrict__ z, DropCfg drop, uint64_t site,
                                                                 bf16_t* __restrict__ zh, const float* __restrict__ res,
                                                                 DropCfg pre, uint64_t presite, bf16_t* __restrict__ xhat_h) {
    // res != null: the normalised row is res + dropout_pre(x) (post-norm residual block: LayerNorm(x_in + Dropout(branch)))
    // xhat_h: x_hat as a bf16 image (with xhat == null: INSTEAD of the fp32 one -- its only reader, the backward, can take either)
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6), d4 = d >> 2;
    if (row >= rows) return;
    const float4* p = reinterpret_cast<const float4*>(x + (size_t)row * d);
    float4 xv[LN_DV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_DV; ++j) {
        const int q = lane + 64 * j;
        xv[j] = q < d4 ? p[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (res && q < d4) {
            float sc[4];
            dropout_scale4(pre, presite, (uint64_t)row * d + (uint64_t)q * 4, sc);
            const float4 r = reinterpret_cast<const float4*>(res + (size_t)row * d)[q];
            xv[j] = make_float4(fmaf(xv[j].x, sc[0], r.x), fmaf(xv[j].y, sc[1], r.y), fmaf(xv[j].z, sc[2], r.z), fmaf(xv[j].w, sc[3], r.w));
        }
        s += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
    }
    const float mu = wave_sum(s) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < LN_DV; ++j)
        if (lane + 64 * j < d4) {
            const float a = xv[j].x - mu, b = xv[j].y - mu, c = xv[j].z - mu, e = xv[j].w - mu;
            v = fmaf(a, a, fmaf(b, b, fmaf(c, c, fmaf(e, e, v))));
        }
    const float rs = 1.0f / sqrtf(wave_sum(v) / (float)d + eps);
    if (lane == 0 && rstd) rstd[row] = rs;
#pragma unroll
    for (int j = 0; j < LN_DV; ++j) {
        const int q = lane + 64 * j;
        if (q < d4) {
            const float4 h = make_float4((xv[j].x - mu) * rs, (xv[j].y - mu) * rs, (xv[j].z - mu) * rs, (xv[j].w - mu) * rs);
            if (xhat) reinterpret_cast<float4*>(xhat + (size_t)row * d)[q] = h;
            if (xhat_h) {
                const bf16x4 hh = {(bf16_t)h.x, (bf16_t)h.y, (bf16_t)h.z, (bf16_t)h.w};
                reinterpret_cast<bf16x4*>(xhat_h + (size_t)row * d)[q] = hh;
            }
            const float4 g = reinterpret_cast<const float4*>(gamma)[q], b = reinterpret_cast<const float4*>(beta)[q];
            float sc[4];
            dropout_scale4(drop, site, (uint64_t)row * d + (uint64_t)q * 4, sc);
            const float4 zv = make_float4(fmaf(h.x, g.x, b.x) * sc[0], fmaf(h.y, g.y, b.y) * sc[1], fmaf(h.z, g.z, b.z) * sc[2], fmaf(h.w, g.w, b.w) * sc[3]);
            if (z) reinterpret_cast<float4*>(z + (size_t)row * d)[q] = zv;
            if (zh) {
                bf16x4 hv;
                hv[0] = (bf16_t)zv.x; hv[1] = (bf16_t)zv.y; hv[2] = (bf16_t)zv.z; hv[3] = (bf16_t)zv.w;
                reinterpret_cast<bf16x4*>(zh + (size_t)row * d)[q] = hv;
            }
        }
    }
}

__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(float* __restrict__ dz_dy, int rows, int d,
                                                                 const float* __restrict__ gamma, const float* __restrict__ xhat,
                                                                 const float* __restrict__ rstd, float* __restrict__ dx,
                                                                 DropCfg drop, uint64_t site, float* __restrict__ dbranch,
                                                                 DropCfg pre, uint64_t presite) {
    // dbranch != null: also the gradient of the branch input of the residual form, dx * dropout_pre mask
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6), d4 = d >> 2;
    if (row >= rows) return;
    float4* g = reinterpret_cast<float4*>(dz_dy + (size_t)row * d);
    const float4* h = reinterpret_cast<const float4*>(xhat + (size_t)row * d);
    float4 tv[LN_DV], hv[LN_DV];      // dy * gamma, xhat
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_DV; ++j) {
        const int q = lane + 64 * j;
        tv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        hv[j] = tv[j];
        if (q < d4) {
            float sc[4];
            dropout_scale4(drop, site, (uint64_t)row * d + (uint64_t)q * 4, sc);
            float4 dy = g[q];
            dy = make_float4(dy.x * sc[0], dy.y * sc[1], dy.z * sc[2], dy.w * sc[3]);
            if (drop.p > 0.f) g[q] = dy;      // without output dropout dz is left as it is (callers may pass a read-only tensor)
            const float4 gm = reinterpret_cast<const float4*>(gamma)[q];
            hv[j] = h[q];
            tv[j] = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
            c1 += (tv[j].x + tv[j].y) + (tv[j].z + tv[j].w);
            c2 = fmaf(tv[j].x, hv[j].x, fmaf(tv[j].y, hv[j].y, fmaf(tv[j].z, hv[j].z, fmaf(tv[j].w, hv[j].w, c2))));
        }
    }
    c1 = wave_sum(c1) / (float)d;
    c2 = wave_sum(c2) / (float)d;
    const float rs = rstd[row];
#pragma unroll
    for (int j = 0; j < LN_DV; ++j) {
        const int q = lane + 64 * j;
        if (q < d4) {
            const float4 o = make_float4(rs * (tv[j].x - c1 - hv[j].x * c2), rs * (tv[j].y - c1 - hv[j].y * c2),
                                         rs * (tv[j].z - c1 - hv[j].z * c2), rs * (tv[j].w - c1 - hv[j].w * c2));
            reinterpret_cast<float4*>(dx + (size_t)row * d)[q] = o;
            if (dbranch) {
                float sc[4];
                dropout_scale4(pre, presite, (uint64_t)row * d + (uint64_t)q * 4, sc);
                reinterpret_cast<float4*>(dbranch + (size_t)row * d)[q] = make_float4(o.x * sc[0], o.y * sc[1], o.z * sc[2], o.w * sc[3]);
            }
        }
    }
}

// LayerNorm backward WITH the parameter-gradient sums of its rows: a workgroup walks rows blockIdx.x * 4 + wave, + 4 gridDim.x, ...
// and every lane keeps the column sums of dy * xhat (-> d gamma), dy (-> d beta) and, K == 3, of dx (the residual's share: the
// gradient of a parameter that is added to every row) for its own columns; the four waves are added up in LDS and the workgroup
// writes ONE partial row set, partial[(blockIdx.x * K + k) * d + n] -- the layout colsum_vec_final_kernel<K> sums.  K == 3 also does
// what the text side's colsum3 pass did on its way: rows whose row_flag[row / flag_div] == 0 are zeroed in dx (AFTER their
// contribution to the third sum) and dx's bf16 image is written.  Saves the separate pass over dz, xhat and dx (3 x rows x d x 4
// bytes: 1.3 GB at 4096 windows) and a launch.
// XH: x_hat comes as the bf16 image the forward wrote instead of the fp32 one (launch_layernorm_fwd's xhat_h); dx may be null when
// its bf16 image dxh is all the consumer reads: 0.8 GB in + 0.4 GB out per launch at 4096 windows become 0.6 + 0.2
template <int K, bool XH>
__global__ __launch_bounds__(256) void layernorm_bwd_sums_kernel(const float* __restrict__ dz_dy, int rows, int d, const float* __restrict__ gamma,
                                                                  const void* __restrict__ xhat_any, const float* __restrict__ rstd,
                                                                  float* __restrict__ dx, DropCfg drop, uint64_t site,
                                                                  float* __restrict__ partial, const unsigned char* __restrict__ row_flag,
                                                                  int flag_div, bf16_t* __restrict__ dxh) {
    const float* xhat = XH ? nullptr : static_cast<const float*>(xhat_any);
    const bf16_t* xhat_h = XH ? static_cast<const bf16_t*>(xhat_any) : nullptr;
    extern __shared__ __attribute__((aligned(16))) float ln_red[];       // [3 waves][K][d]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d4 = d >> 2;
    float4 sw[LN_DV], sb[LN_DV], sq[K == 3 ? LN_DV : 1];
#pragma unroll
    for (int j = 0; j < LN_DV; ++j) {
        sw[j] = sb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (K == 3) sq[j] = sw[j];
    }
    float4 gm[LN_DV];
#pragma unroll
    for (int j = 0; j < LN_DV; ++j) {
        const int q = lane + 64 * j;
        gm[j] = q < d4 ? reinterpret_cast<const float4*>(gamma)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int row = blockIdx.x * 4 + wave; row < rows; row += 4 * gridDim.x) {
        const float4* g = reinterpret_cast<const float4*>(dz_dy + (size_t)row * d);
        const float4* h = XH ? nullptr : reinterpret_cast<const float4*>(xhat + (size_t)row * d);
        const bf16x4* hh = XH ? reinterpret_cast<const bf16x4*>(xhat_h + (size_t)row * d) : nullptr;
        float4 tv[LN_DV], hv[LN_DV];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_DV; ++j) {
            const int q = lane + 64 * j;
            tv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            hv[j] = tv[j];
            if (q < d4) {
                float sc[4];
                dropout_scale4(drop, site, (uint64_t)row * d + (uint64_t)q * 4, sc);
                float4 dy = g[q];
                dy = make_float4(dy.x * sc[0], dy.y * sc[1], dy.z * sc[2], dy.w * sc[3]);
                // (dy is NOT written back: its only readers were the column sums, which are formed right here -- 0.4 GB of the kernel's
                // 1.6 GB at 4096 windows with dropout on)
                if (XH) { const bf16x4 t4 = hh[q]; hv[j] = make_float4((float)t4[0], (float)t4[1], (float)t4[2], (float)t4[3]); }
                else hv[j] = h[q];
                sw[j].x = fmaf(dy.x, hv[j].x, sw[j].x); sw[j].y = fmaf(dy.y, hv[j].y, sw[j].y);
                sw[j].z = fmaf(dy.z, hv[j].z, sw[j].z); sw[j].w = fmaf(dy.w, hv[j].w, sw[j].w);
                sb[j].x += dy.x; sb[j].y += dy.y; sb[j].z += dy.z; sb[j].w += dy.w;
                tv[j] = make_float4(dy.x * gm[j].x, dy.y * gm[j].y, dy.z * gm[j].z, dy.w * gm[j].w);
                c1 += (tv[j].x + tv[j].y) + (tv[j].z + tv[j].w);
                c2 = fmaf(tv[j].x, hv[j].x, fmaf(tv[j].y, hv[j].y, fmaf(tv[j].z, hv[j].z, fmaf(tv[j].w, hv[j].w, c2))));
            }
        }
        c1 = wave_sum(c1) / (float)d;
        c2 = wave_sum(c2) / (float)d;
        const float rs = rstd[row];
        const bool keep = !(K == 3 && row_flag) || row_flag[row / flag_div] != 0;
#pragma unroll
        for (int j = 0; j < LN_DV; ++j) {
            const int q = lane + 64 * j;
            if (q < d4) {
                float4 o = make_float4(rs * (tv[j].x - c1 - hv[j].x * c2), rs * (tv[j].y - c1 - hv[j].y * c2),
                                       rs * (tv[j].z - c1 - hv[j].z * c2), rs * (tv[j].w - c1 - hv[j].w * c2));
                if (K == 3) { sq[j].x += o.x; sq[j].y += o.y; sq[j].z += o.z; sq[j].w += o.w; }
                if (!keep) o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (dx) reinterpret_cast<float4*>(dx + (size_t)row * d)[q] = o;
                if (K == 3 && dxh) {
                    const bf16x4 hvv = {(bf16_t)o.x, (bf16_t)o.y, (bf16_t)o.z, (bf16_t)o.w};
                    reinterpret_cast<bf16x4*>(dxh + (size_t)row * d)[q] = hvv;
                }
            }
        }
    }
    // waves 1..3 park their sums, wave 0 adds them and writes the workgroup's partial rows
    float4* red4 = reinterpret_cast<float4*>(ln_red);
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < LN_DV; ++j) {
            const int q = lane + 64 * j;
            if (q < d4) {
                red4[((wave - 1) * K + 0) * d4 + q] = sw[j];
                red4[((wave - 1) * K + 1) * d4 + q] = sb[j];
                if (K == 3) red4[((wave - 1) * K + 2) * d4 + q] = sq[j];
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < LN_DV; ++j) {
            const int q = lane + 64 * j;
            if (q < d4) {
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const float4 a = red4[(w * K + 0) * d4 + q], b = red4[(w * K + 1) * d4 + q];
                    sw[j].x += a.x; sw[j].y += a.y; sw[j].z += a.z; sw[j].w += a.w;
                    sb[j].x += b.x; sb[j].y += b.y; sb[j].z += b.z; sb[j].w += b.w;
                    if (K == 3) {
                        const float4 c = red4[(w * K + 2) * d4 + q];
                        sq[j].x += c.x; sq[j].y += c.y; sq[j].z += c.z; sq[j].w += c.w;
                    }
                }
                float4* out = reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * K * d);
                out[0 * d4 + q] = sw[j];
                out[1 * d4 + q] = sb[j];
                if (K == 3) out[2 * d4 + q] = sq[j];
            }
        }
    }
}

// LayerNorm backward + sums (the K == 3, bf16-x_hat form above) whose upstream gradient arrives LOW-RANK: dy[row, :] = G[row, :PW] Wb
// (Wb: PW x d) -- the gradient a rank-PW projection behind the LayerNorm sends back (MMF_XAttn_Add's composed projection: dZ = dP Wc,
// csrc/xrank.hip).  The product is formed in registers from Wb held in LDS, so the (rows x d) fp32 dZ that rank_expand_kernel wrote and
// this kernel re-read (0.8 GB of the step's traffic at 4096 windows, one launch on the text chain at 64) never exists.  512 threads:
// eight waves share the LDS copy of Wb; a wave takes R consecutive rows at a time and every Wb read from LDS serves all R (R = 4 for
// many rows: the LDS reads, PW x d x 4 bytes per row at R = 1, would otherwise cost as much as the HBM traffic).
template <int R, int DV, bool XH>
__global__ __launch_bounds__(512) void layernorm_bwd_lr_kernel(const float* __restrict__ G, int ldg, int PW, const float* __restrict__ Wb,
                                                                int rows, int d, const float* __restrict__ gamma,
                                                                const void* __restrict__ xhat_any, const float* __restrict__ rstd,
                                                                float* __restrict__ dx, bf16_t* __restrict__ dxh, DropCfg drop, uint64_t site,
                                                                float* __restrict__ partial, const unsigned char* __restrict__ row_flag,
                                                                int flag_div, const unsigned long long* __restrict__ keepw, int T) {
    // keepw (optional): the forward's keep bits of this dropout site, word ((row / T) * 2 + (row % T) / 16) * 256 + q holds four bits per
    // step (row % T) % 16 of column chunk q (t2v_mix_ln_fwd_kernel): one 8-byte load per chunk instead of a Philox call per (row, chunk)
    // -- the generator was the largest share of this kernel's instructions
    extern __shared__ __attribute__((aligned(16))) float lr_lds[];      // Wb [PW][d], then the workgroup's sums [3][d]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d4 = d >> 2;
    float4* wl4 = reinterpret_cast<float4*>(lr_lds);
    float4* red4 = wl4 + (size_t)PW * d4;
    for (int x = tid; x < PW * d4; x += 512) wl4[x] = reinterpret_cast<const float4*>(Wb)[x];
    float4 sw[DV], sb[DV], sq[DV], gm[DV];
#pragma unroll
    for (int j = 0; j < DV; ++j) {
        const int q = lane + 64 * j;
        sw[j] = sb[j] = sq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        gm[j] = q < d4 ? reinterpret_cast<const float4*>(gamma)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // the next rows' operands (x_hat as stored, the coefficient rows) are fetched while the current ones are worked on: with one
    // workgroup per CU (the LDS copy of Wb) there are two waves per SIMD to hide a load behind, not eight
    typedef typename std::conditional<XH, uint2, float4>::type Raw;
    Raw hr[R][DV];
    float gn[R];
    const int stride = gridDim.x * 8 * R;
    auto fetch = [&](int row0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            const bool ok = row < rows;
            gn[r] = (ok && lane < PW) ? G[(size_t)row * ldg + lane] : 0.f;
#pragma unroll
            for (int j = 0; j < DV; ++j) {
                const int q = lane + 64 * j;
                if (ok && q < d4) {
                    if constexpr (XH) hr[r][j] = reinterpret_cast<const uint2*>(static_cast<const bf16_t*>(xhat_any) + (size_t)row * d)[q];
                    else hr[r][j] = reinterpret_cast<const float4*>(static_cast<const float*>(xhat_any) + (size_t)row * d)[q];
                } else {
                    if constexpr (XH) hr[r][j] = make_uint2(0u, 0u);
                    else hr[r][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    };
    const int first = (blockIdx.x * 8 + wave) * R;
    if (first < rows) fetch(first);
    for (int row0 = first; row0 < rows; row0 += stride) {
        float4 dy[R][DV], hv[R][DV];
        float gl[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            gl[r] = gn[r];
#pragma unroll
            for (int j = 0; j < DV; ++j) {
                dy[r][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (XH) {
                    const uint2 t = hr[r][j];
                    hv[r][j] = make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16),
                                           __uint_as_float(t.y & 0xffff0000u));
                } else {
                    hv[r][j] = hr[r][j];
                }
            }
        }
        if (row0 + stride < rows) fetch(row0 + stride);
#pragma unroll 2
        for (int k = 0; k < PW; ++k) {
            float gk[R];
#pragma unroll
            for (int r = 0; r < R; ++r) gk[r] = lane_bcast(gl[r], k);
#pragma unroll
            for (int j = 0; j < DV; ++j) {
                const int q = lane + 64 * j;
                if (q < d4) {
                    const float4 w4 = wl4[k * d4 + q];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        dy[r][j].x = fmaf(gk[r], w4.x, dy[r][j].x); dy[r][j].y = fmaf(gk[r], w4.y, dy[r][j].y);
                        dy[r][j].z = fmaf(gk[r], w4.z, dy[r][j].z); dy[r][j].w = fmaf(gk[r], w4.w, dy[r][j].w);
                    }
                }
            }
        }
        // output-dropout keep bits of the wave's R x DV chunks in a rolled loop (one Philox call each), four bits per chunk
        uint64_t kb = ~0ull;
        if (drop.p > 0.f && keepw) {
            kb = 0ull;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = min(row0 + r, rows - 1), bw = row / T, t = row - bw * T;
#pragma unroll
                for (int j = 0; j < DV; ++j) {
                    const int q = lane + 64 * j;
                    const unsigned long long wv = q < d4 ? keepw[((size_t)bw * 2 + (t >> 4)) * 256 + q] : 0ull;
                    kb |= ((wv >> (4 * (t & 15))) & 15ull) << (4 * (r * DV + j));
                }
            }
        } else if (drop.p > 0.f) {
            kb = 0ull;
#pragma unroll 1
            for (int x = 0; x < R * DV; ++x) {
                const int r = x / DV, j = x - r * DV, q = lane + 64 * j, row = row0 + r;
                float sc[4];
                dropout_scale4(drop, site, (uint64_t)row * d + (uint64_t)q * 4, sc);
                const uint64_t bits = (sc[0] != 0.f ? 1ull : 0ull) | (sc[1] != 0.f ? 2ull : 0ull) | (sc[2] != 0.f ? 4ull : 0ull) | (sc[3] != 0.f ? 8ull : 0ull);
                kb |= bits << (4 * x);
            }
        }
        const float keepv = drop.p > 0.f ? drop.inv_keep : 1.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r;
            const bool ok = row < rows;
            float4 tv[DV];
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int j = 0; j < DV; ++j) {
                const uint32_t b4 = (uint32_t)(kb >> (4 * (r * DV + j))) & 15u;
                float4 y = dy[r][j];
                y = make_float4((b4 & 1u) ? y.x * keepv : 0.f, (b4 & 2u) ? y.y * keepv : 0.f, (b4 & 4u) ? y.z * keepv : 0.f, (b4 & 8u) ? y.w * keepv : 0.f);
                const float4 h = hv[r][j];
                sw[j].x = fmaf(y.x, h.x, sw[j].x); sw[j].y = fmaf(y.y, h.y, sw[j].y); sw[j].z = fmaf(y.z, h.z, sw[j].z); sw[j].w = fmaf(y.w, h.w, sw[j].w);
                sb[j].x += y.x; sb[j].y += y.y; sb[j].z += y.z; sb[j].w += y.w;
                tv[j] = make_float4(y.x * gm[j].x, y.y * gm[j].y, y.z * gm[j].z, y.w * gm[j].w);
                c1 += (tv[j].x + tv[j].y) + (tv[j].z + tv[j].w);
                c2 = fmaf(tv[j].x, h.x, fmaf(tv[j].y, h.y, fmaf(tv[j].z, h.z, fmaf(tv[j].w, h.w, c2))));
            }
            c1 = wave_sum(c1) / (float)d;
            c2 = wave_sum(c2) / (float)d;
            const float rs = ok ? rstd[row] : 0.f;
            const bool keep = !row_flag || !ok || row_flag[row / flag_div] != 0;
#pragma unroll
            for (int j = 0; j < DV; ++j) {
                const int q = lane + 64 * j;
                if (ok && q < d4) {
                    const float4 h = hv[r][j];
                    float4 o = make_float4(rs * (tv[j].x - c1 - h.x * c2), rs * (tv[j].y - c1 - h.y * c2), rs * (tv[j].z - c1 - h.z * c2),
                                           rs * (tv[j].w - c1 - h.w * c2));
                    sq[j].x += o.x; sq[j].y += o.y; sq[j].z += o.z; sq[j].w += o.w;
                    if (!keep) o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (dx) reinterpret_cast<float4*>(dx + (size_t)row * d)[q] = o;
                    if (dxh) {
                        const bf16x4 hvv = {(bf16_t)o.x, (bf16_t)o.y, (bf16_t)o.z, (bf16_t)o.w};
                        reinterpret_cast<bf16x4*>(dxh + (size_t)row * d)[q] = hvv;
                    }
                }
            }
        }
    }
    // the eight waves add up in wave order (deterministic), then the workgroup writes its partial row set
    for (int w = 0; w < 8; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < DV; ++j) {
                const int q = lane + 64 * j;
                if (q < d4) {
                    if (w == 0) { red4[0 * d4 + q] = sw[j]; red4[1 * d4 + q] = sb[j]; red4[2 * d4 + q] = sq[j]; }
                    else {
                        float4 a = red4[0 * d4 + q], b = red4[1 * d4 + q], c = red4[2 * d4 + q];
                        a.x += sw[j].x; a.y += sw[j].y; a.z += sw[j].z; a.w += sw[j].w;
                        b.x += sb[j].x; b.y += sb[j].y; b.z += sb[j].z; b.w += sb[j].w;
                        c.x += sq[j].x; c.y += sq[j].y; c.z += sq[j].z; c.w += sq[j].w;
                        red4[0 * d4 + q] = a; red4[1 * d4 + q] = b; red4[2 * d4 + q] = c;
                    }
                }
            }
        }
        __syncthreads();
    }
    float4* out = reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * 3 * d);
    for (int x = tid; x < 3 * d4; x += 512) out[x] = red4[x];
}

// Non-stationary-Transformer normalisation of the history (reference: models/PatchTST.py:104-109, models/TimesNet.py:113-117 -- mean over
// time, biased variance, eps 1e-5 inside the root): a thread per series (b, c) walks its L values twice; x (B, L, C) -> xn, means (B, C),
// stdev (B, C).  The six eager launches of the expression as written were 37 us at the head of the backbone's branch.
__global__ __launch_bounds__(256) void instance_norm_kernel(const float* __restrict__ x, int B, int L, int C, float* __restrict__ xn,
                                                             float* __restrict__ means, float* __restrict__ stdev) {
    // one workgroup per window: its L x C values through LDS (coalesced in and out), a thread per variable for the two statistics
    // (a thread per series walking global memory with stride C was 15 us of exposed load latency)
    extern __shared__ float in_lds[];       // [L * C] values, then [C] mean, [C] 1 / stdev
    const int b = blockIdx.x, n = L * C;
    const float* p = x + (size_t)b * n;
    float* mu_s = in_lds + n;
    float* rs_s = mu_s + C;
    for (int i = threadIdx.x; i < n; i += 256) in_lds[i] = p[i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int t = 0; t < L; ++t) s += in_lds[t * C + c];
        const float mu = s / (float)L;
        float v = 0.f;
        for (int t = 0; t < L; ++t) { const float q = in_lds[t * C + c] - mu; v = fmaf(q, q, v); }
        const float sd = sqrtf(v / (float)L + 1e-5f);
        means[(size_t)b * C + c] = mu;
        stdev[(size_t)b * C + c] = sd;
        mu_s[c] = mu;
        rs_s[c] = sd;
    }
    __syncthreads();
    float* o = xn + (size_t)b * n;
    for (int i = threadIdx.x; i < n; i += 256) { const int c = i % C; o[i] = (in_lds[i] - mu_s[c]) / rs_s[c]; }
}

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(float* __restrict__ dz_dy, int rows, int d,
                                                             const float* __restrict__ gamma, const float* __restrict__ xhat,
                                                             const float* __restrict__ rstd, float* __restrict__ dx,
                                                             DropCfg drop, uint64_t site) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* g = dz_dy + (size_t)row * d;
    const float* h = xhat + (size_t)row * d;
    float c1 = 0.f, c2 = 0.f;
    for (int i = lane; i < d; i += 64) {
        const float dy = g[i] * dropout_scale(drop, site, (uint64_t)row * d + i);
        g[i] = dy;
        const float t = dy * gamma[i];
        c1 += t;
        c2 = fmaf(t, h[i], c2);
    }
    c1 = wave_sum(c1) / (float)d;
    c2 = wave_sum(c2) / (float)d;
    const float rs = rstd[row];
    for (int i = lane; i < d; i += 64) dx[(size_t)row * d + i] = rs * (g[i] * gamma[i] - c1 - h[i] * c2);
}

// ------------------------------------------------------------------------------------------- tiny vector ops
__device__ __forceinline__ void matvec_body(int bid, const float* __restrict__ W, int ldw, const float* __restrict__ x,
                                            const float* __restrict__ b, int rows, int cols,
                                            float* __restrict__ y, float* __restrict__ ys, float scale,
                                            float* __restrict__ y_nobias) {
    const int lane = threadIdx.x & 63, row = bid * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float a = 0.f;
    for (int j = lane; j < cols; j += 64) a = fmaf(W[(size_t)row * ldw + j], x[j], a);
    a = wave_sum(a);
    if (lane == 0) {
        if (y_nobias) y_nobias[row] = a;
        if (b) a += b[row];
        if (y) y[row] = a;
        if (ys) ys[row] = a * scale;
    }
}
__global__ __launch_bounds__(256) void matvec_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ x,
                                                      const float* __restrict__ b, int rows, int cols,
                                                      float* __restrict__ y, float* __restrict__ ys, float scale,
                                                      float* __restrict__ y_nobias) {
    matvec_body(blockIdx.x, W, ldw, x, b, rows, cols, y, ys, scale, y_nobias);
}
// Three independent row jobs at the head of TTF_T2V_XAttn's forward -- pack + cast the notes, Time2Vec of their time stamps, the
// learned query's in-projection (parameters only) -- as ONE launch: workgroups [0, na) gather, [na, na + nb) Time2Vec, the rest
// the mat-vec.  (Each was a launch of its own on the text side's forward chain; a dependent launch costs ~5 us whatever it does.)
struct GatherJob { const float* src; int ld_src; const int* rowmap; const int* total; int width; float* dst; int ld_dst; bf16_t* dst_h; int vec;
                   const float* score_u; float* score_out; };      // vec == 2, optional: score_out[r] = (the bf16 row | its Time2Vec columns) . score_u
struct T2VJob { const float* tau_pad; const int* rowmap; const int* total; int d_tau; const float *w0, *b0, *w, *b; float* dst; int ld_dst, max_rows; bf16_t* dst_h; };
struct MatvecJob { const float* W; int ldw; const float *x, *b; int rows, cols; float *y, *ys; float scale; };
// a packed row per WAVE (vec == 2: 16-byte-aligned rows, bf16 destination): the row's gather + cast with up to four 16-byte loads per
// lane in flight, and its Time2Vec columns from one load of the time stamp.  (A workgroup per row and a thread per Time2Vec element
// were 131 072 + 196 608 workgroups at 4096 windows, half of them empty: 290 us for 330 MB.)
__device__ __forceinline__ void notes_row_wave_body(int r, const GatherJob& a, const T2VJob& t) {
    if (r >= *a.total) return;
    const int lane = threadIdx.x & 63;
    const float4* s4 = reinterpret_cast<const float4*>(a.src + (size_t)a.rowmap[r] * a.ld_src);
    bf16x4* d4 = reinterpret_cast<bf16x4*>(a.dst_h + (size_t)r * a.ld_dst);
    const float tv = t.tau_pad[t.rowmap ? t.rowmap[r] : r];
    const int n4 = a.width >> 2;
    const float4* u4 = reinterpret_cast<const float4*>(a.score_u);
    float dot = 0.f;
    for (int i0 = lane; i0 < n4; i0 += 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u;
            v[u] = i < n4 ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u;
            if (i < n4) {
                bf16x4 h;
                h[0] = (bf16_t)v[u].x; h[1] = (bf16_t)v[u].y; h[2] = (bf16_t)v[u].z; h[3] = (bf16_t)v[u].w;
                d4[i] = h;
                if (u4) {       // the score reads the values the other passes will read: the bf16 image
                    const float4 w = u4[i];
                    dot = fmaf((float)h[0], w.x, dot); dot = fmaf((float)h[1], w.y, dot); dot = fmaf((float)h[2], w.z, dot); dot = fmaf((float)h[3], w.w, dot);
                }
            }
        }
    }
    for (int j = lane; j < t.d_tau; j += 64) {
        const float v = (j == 0) ? fmaf(t.w0[0], tv, t.b0[0]) : sinf(fmaf(t.w[j - 1], tv, t.b[j - 1]));
        if (t.dst) t.dst[(size_t)r * t.ld_dst + j] = v;
        if (t.dst_h) t.dst_h[(size_t)r * t.ld_dst + j] = (bf16_t)v;
        if (a.score_u) dot = fmaf((float)(bf16_t)v, a.score_u[a.width + j], dot);
    }
    if (a.score_u) {
        dot = wave_sum(dot);
        if (lane == 0) a.score_out[r] = dot;
    }
}
__global__ __launch_bounds__(256) void notes_stage_kernel(GatherJob a, int na, T2VJob t, int nb, MatvecJob m, int nc, MatvecJob m2) {
    int bid = blockIdx.x;
    if (bid < na) {
        if (a.vec == 2) notes_row_wave_body(bid * 4 + (threadIdx.x >> 6), a, t);
        else gather_rows_body(bid, a.src, a.ld_src, a.rowmap, a.total, a.width, a.dst, a.ld_dst, a.dst_h, a.vec);
        return;
    }
    bid -= na;
    if (bid < nb) { time2vec_fwd_body(bid, t.tau_pad, t.rowmap, t.total, t.d_tau, t.w0, t.b0, t.w, t.b, t.dst, t.ld_dst, t.max_rows, t.dst_h); return; }
    bid -= nb;
    if (bid < nc) { matvec_body(bid, m.W, m.ldw, m.x, m.b, m.rows, m.cols, m.y, m.ys, m.scale, nullptr); return; }
    matvec_body(bid - nc, m2.W, m2.ldw, m2.x, m2.b, m2.rows, m2.cols, m2.y, m2.ys, m2.scale, nullptr);       // (a second, optional mat-vec)
}

// y[j] = sum_i W[i,j] x[i]: 64 columns x 16 row lanes per workgroup, rows strided (coalesced over j)
// gridDim.y row slabs; with more than one slab the result is ADDED to y with atomics (y must hold the running value)
__global__ __launch_bounds__(1024) void matvec_t_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ x,
                                                         int rows, int cols, float* __restrict__ y, int accumulate) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, j = blockIdx.x * 64 + tx;
    const int rps = (rows + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * rps, r1 = min(rows, r0 + rps);
    float a = 0.f;
    if (j < cols)
        for (int i = r0 + ty; i < r1; i += 16) a = fmaf(W[(size_t)i * ldw + j], x[i], a);
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && j < cols) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += red[k][tx];
        if (gridDim.y > 1) atomicAdd(y + j, t);
        else y[j] = accumulate ? y[j] + t : t;
    }
}

__global__ __launch_bounds__(256) void outer_kernel(const float* __restrict__ a, const float* __restrict__ b, int rows,
                                                     int cols, float* __restrict__ out, int ld) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)rows * cols) return;
    const int i = (int)(idx / cols), j = (int)(idx % cols);
    out[(size_t)i * ld + j] = a[i] * b[j];
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ src, float alpha, float* __restrict__ dst, int n,
                                                    int accumulate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = alpha * src[i];
    dst[i] = accumulate ? dst[i] + v : v;
}

// Backward of a learned query q = W_q Q + b_q, qs = scale * q (TTF_T2V_XAttn's single query) in ONE launch, from the
// per-window partial gradients dqs_part (B, d):
//   dq = scale * sum_b dqs_part[b] ;  dW_q = dq Q^T ;  db_q = dq ;  dQ += W_q^T dq
// (colsum partial + final, axpy, outer, axpy, transposed mat-vec: six launches of ~5 us each otherwise.)
// grid ceil(d / QB_ROWS): a workgroup owns QB_ROWS rows i: dq[i] (16 lanes per row over the B windows), those rows of dW_q and
// db_q, and its rows' share of W_q^T dq, added into dQ with one atomic per column.
constexpr int QB_ROWS = 16;
__device__ __forceinline__ void query_bwd_body(int bid, float* dq, const float* __restrict__ dqs_part, int B, int d, float scale,
                                               const float* __restrict__ Wq, int ldw, const float* __restrict__ Q,
                                               float* __restrict__ dWq, int ldg, float* __restrict__ dbq,
                                               float* __restrict__ dQ) {
    // dq[QB_ROWS] (LDS): only this workgroup's rows of dq enter its share of all three results
    const int tid = threadIdx.x, i0 = bid * QB_ROWS, rows = min(QB_ROWS, d - i0);
    {
        const int r = tid >> 4, part = tid & 15;      // 16 lanes per row split the B windows
        float a = 0.f;
        if (r < rows)
            for (int b = part; b < B; b += 16) a += dqs_part[(size_t)b * d + i0 + r];
        a = row16_sum(a);
        if (part == 0 && r < rows) dq[r] = a * scale;
    }
    __syncthreads();
    if (tid < rows) dbq[i0 + tid] = dq[tid];
    for (int j = tid; j < d; j += 256) {
        const float qj = Q[j];
        float a = 0.f;
        for (int r = 0; r < rows; ++r) {
            const float g = dq[r];
            dWq[(size_t)(i0 + r) * ldg + j] = g * qj;
            a = fmaf(Wq[(size_t)(i0 + r) * ldw + j], g, a);
        }
        atomicAdd(dQ + j, a);
    }
}
__global__ __launch_bounds__(256) void query_bwd_kernel(const float* __restrict__ dqs_part, int B, int d, float scale,
                                                         const float* __restrict__ Wq, int ldw, const float* __restrict__ Q,
                                                         float* __restrict__ dWq, int ldg, float* __restrict__ dbq,
                                                         float* __restrict__ dQ) {
    __shared__ float dq[QB_ROWS];
    query_bwd_body(blockIdx.x, dq, dqs_part, B, d, scale, Wq, ldw, Q, dWq, ldg, dbq, dQ);
}
// the learned query's backward and the first stage of Time2Vec's parameter gradients in one launch (both end in parameter
// gradients only; the query's sat in the middle of the text side's backward chain): workgroups [0, nq) the query, the rest
// Time2Vec's (column block, slab) grid
struct QueryBwdJob { const float* dqs_part; int B, d; float scale; const float* Wq; int ldw; const float* Q; float* dWq; int ldg; float *dbq, *dQ; };
struct T2VBwdJob { const float* tau_pad; const int *rowmap, *total; int d_tau; const float *w, *b, *dfeat; int ld; float* partial; int max_rows, gx, nsl; };
__global__ __launch_bounds__(256) void query_t2v_bwd_kernel(QueryBwdJob q, int nq, T2VBwdJob t) {
    __shared__ float red[2][4][64];
    int bid = blockIdx.x;
    if (bid < nq) {
        query_bwd_body(bid, &red[0][0][0], q.dqs_part, q.B, q.d, q.scale, q.Wq, q.ldw, q.Q, q.dWq, q.ldg, q.dbq, q.dQ);
        return;
    }
    bid -= nq;
    time2vec_bwd_body(bid % t.gx, bid / t.gx, t.nsl, red, t.tau_pad, t.rowmap, t.total, t.d_tau, t.w, t.b, t.dfeat, t.ld, t.partial, t.max_rows);
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ dst, float v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint64_t seed, uint64_t site, size_t n, float p,
                                                            unsigned char* __restrict__ out) {
    const float inv = 1.f / (1.f - p);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = dropout_scale(seed, site, i, p, inv) != 0.f ? 1 : 0;
}

}  // namespace

int launch_note_mask(const float* V, int rows, int d_m, unsigned char* mask, int* nan_flag, hipStream_t s) {
    if (rows <= 0) return IMMTSF_OK;
    const int vec = ((d_m % 4) == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(note_mask_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, V, rows, d_m, mask, nan_flag, vec);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_ragged_index(const unsigned char* mask, int B, int N, int* lengths, int* offsets, int* rowmap, int* seg,
                        unsigned char* mtxt, hipStream_t s, unsigned char* mtxt2) {
    if (B > 256) {
        hipLaunchKernelGGL(ragged_count_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, mask, B, N, lengths, mtxt, mtxt2);
        IMMTSF_LAUNCH_CHECK();
        hipLaunchKernelGGL(ragged_scan_kernel, dim3(1), dim3(1024), 0, s, lengths, B, offsets);
        IMMTSF_LAUNCH_CHECK();
        hipLaunchKernelGGL(ragged_fill_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, mask, B, N, offsets, rowmap, seg);
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    hipLaunchKernelGGL(ragged_index_kernel, dim3(1), dim3(1024), 0, s, mask, B, N, lengths, offsets, rowmap, seg, mtxt, mtxt2);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_mask_from_lengths(const int* lengths, int B, int N, unsigned char* mask, hipStream_t s) {
    if (B * N <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(mask_from_lengths_kernel, dim3(cdiv(B * N, 256)), dim3(256), 0, s, lengths, B, N, mask);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_gather_rows(const float* src, int ld_src, const int* rowmap, const int* total, int max_rows, int width,
                       float* dst, int ld_dst, hipStream_t s, void* dst_h) {
    if (max_rows <= 0) return IMMTSF_OK;
    const uintptr_t al = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(dst_h);
    const int vec = ((width | ld_src | ld_dst) & 3) == 0 && (al & 15) == 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(max_rows), dim3(256), 0, s, src, ld_src, rowmap, total, width, dst, ld_dst,
                       static_cast<bf16_t*>(dst_h), vec);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_time2vec_fwd(const float* tau_pad, const int* rowmap, const int* total, int max_rows, int d_tau,
                        const float* w0, const float* b0, const float* w, const float* b, float* dst, int ld_dst,
                        hipStream_t s, void* dst_h) {
    if (max_rows <= 0) return IMMTSF_OK;
    const long n = (long)max_rows * d_tau;
    hipLaunchKernelGGL(time2vec_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tau_pad, rowmap, total, d_tau,
                       w0, b0, w, b, dst, ld_dst, max_rows, static_cast<bf16_t*>(dst_h));
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_notes_stage(const float* src, int ld_src, const int* gmap, const int* total, int max_rows, int width, void* dst_h, int ld_dst,
                       const float* tau_pad, const int* rowmap, int d_tau, const float* w0, const float* b0, const float* w, const float* b,
                       float* t_dst, int t_ld, void* t_dst_h, const float* W, int ldw, const float* x, const float* bias, int rows, int cols,
                       float* y, float* ys, float scale, hipStream_t s, const float* W2, int ldw2, const float* x2, const float* bias2, int rows2,
                       int cols2, float* y2, const float* score_u, float* score_out) {
    if (max_rows <= 0) return IMMTSF_OK;
    const uintptr_t al = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst_h);
    int vec = ((width | ld_src | ld_dst) & 3) == 0 && (al & 15) == 0;
    if (vec && dst_h && gmap && total) vec = 2;          // a packed row per wave, gather + Time2Vec together
    // the fused score rides on the row-per-wave form only, with the Time2Vec columns right behind the row (one vector u for both)
    if (score_u && (vec != 2 || !score_out || (reinterpret_cast<uintptr_t>(score_u) & 15) || t_dst_h != static_cast<bf16_t*>(dst_h) + width || t_ld != ld_dst))
        return IMMTSF_EUNSUPPORTED;
    const GatherJob a{src, ld_src, gmap, total, width, nullptr, ld_dst, static_cast<bf16_t*>(dst_h), vec, score_u, score_out};
    const T2VJob t{tau_pad, rowmap, total, d_tau, w0, b0, w, b, t_dst, t_ld, max_rows, static_cast<bf16_t*>(t_dst_h)};
    const MatvecJob m{W, ldw, x, bias, rows, cols, y, ys, scale};
    const MatvecJob m2{W2, ldw2, x2, bias2, W2 ? rows2 : 0, cols2, y2, nullptr, 1.f};
    const int na = vec == 2 ? cdiv(max_rows, 4) : max_rows, nb = vec == 2 ? 0 : (int)(((long)max_rows * d_tau + 255) / 256);
    const int nc = cdiv(rows, 4), nd = W2 ? cdiv(rows2, 4) : 0;
    hipLaunchKernelGGL(notes_stage_kernel, dim3(na + nb + nc + nd), dim3(256), 0, s, a, na, t, nb, m, nc, m2);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_time2vec_bwd(const float* tau_pad, const int* rowmap, const int* total, int max_rows, int d_tau,
                        const float* w, const float* b, const float* dfeat, int ld, float* dw0, float* db0, float* dw,
                        float* db, float* scratch, int nslabs, hipStream_t s, int accumulate) {
    if (!rowmap && !total && d_tau <= 32 && max_rows <= (1 << 15) && max_rows > 0) {
        int CT = 1;
        while (CT < d_tau) CT <<= 1;
        hipLaunchKernelGGL(time2vec_bwd_small_kernel, dim3(1), dim3(1024), 0, s, tau_pad, max_rows, d_tau, CT, w, b, dfeat, ld, dw0, db0, dw, db,
                           accumulate);
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    if (nslabs < 1) nslabs = kSlabs;
    hipLaunchKernelGGL(time2vec_bwd_kernel, dim3(cdiv(d_tau, 64), nslabs), dim3(256), 0, s, tau_pad, rowmap, total, d_tau, w, b,
                       dfeat, ld, scratch, max_rows);
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL(time2vec_bwd_final_kernel, dim3(cdiv(d_tau, 4)), dim3(256), 0, s, scratch, d_tau, dw0, db0, dw, db, nslabs, accumulate);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_colsum(const float* X, const float* Y, int M, const int* dyn, int N, int ld, float* out, int accumulate,
                  float* scratch, hipStream_t s, bool big_scratch) {
    if (N <= 0) return IMMTSF_OK;
    if (!dyn && colsum_vec_ok(X, Y, nullptr, nullptr, M, N, ld, big_scratch))
        return launch_colsum_vec<1>(X, Y, nullptr, M, N, ld, out, nullptr, nullptr, accumulate, scratch, nullptr, 1, nullptr, s);
    if (N <= 32 && !dyn && (long)M * N <= (1L << 17)) {
        int CT = 1;
        while (CT < N) CT <<= 1;
        hipLaunchKernelGGL(colsum_small_kernel, dim3(1), dim3(1024), 0, s, X, Y, M, N, ld, CT, out, accumulate);
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(N, 64), kSlabs), dim3(256), 0, s, X, Y, M, dyn, N, ld, scratch);
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, scratch, N, out, accumulate);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

bool colsum_small_pair_ok(int M, int N) { return N >= 1 && N <= 32 && M >= 1 && (long)M * N <= (1L << 17); }
int launch_colsum_small_pair(const float* X, const float* Y, const float* Z, int M, int N, int ld, float* out_xy, float* out_x, float* out_z,
                             hipStream_t s, int outputs_zero) {
    int CT = 1;
    while (CT < N) CT <<= 1;
    if (outputs_zero && M >= 128) {       // pre-zeroed outputs: ~32 rows per workgroup, atomics
        const int RT = 256 / CT, per = ((32 + RT - 1) / RT) * RT;
        hipLaunchKernelGGL(colsum_small_atomic_kernel, dim3(cdiv(M, per)), dim3(256), 0, s, X, Y, Z, M, N, ld, CT, per, out_xy, out_x, out_z);
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    hipLaunchKernelGGL(colsum_small_pair_kernel, dim3(2), dim3(1024), 0, s, X, Y, Z, M, N, ld, CT, out_xy, out_x, out_z);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_colsum2(const float* X, const float* Y, int M, int N, int ld, float* out_xy, float* out_x, float* scratch,
                   hipStream_t s, bool big_scratch) {
    if (N <= 0) return IMMTSF_OK;
    if (colsum_vec_ok(X, Y, nullptr, nullptr, M, N, ld, big_scratch))
        return launch_colsum_vec<2>(X, Y, nullptr, M, N, ld, out_xy, out_x, nullptr, 0, scratch, nullptr, 1, nullptr, s);
    if (N <= 32 && (long)M * N <= (1L << 17)) {
        int CT = 1;
        while (CT < N) CT <<= 1;
        const uintptr_t al = reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(out_xy) |
                             reinterpret_cast<uintptr_t>(out_x);
        if ((N & 3) == 0 && (ld & 3) == 0 && (al & 15) == 0 && M >= 256) {
            int CV = 1;
            while (CV < N / 4) CV <<= 1;
            hipLaunchKernelGGL(colsum2_small_vec_kernel, dim3(1), dim3(1024), 0, s, X, Y, M, N, ld, CV, out_xy, out_x);
        } else {
            hipLaunchKernelGGL(colsum2_small_kernel, dim3(1), dim3(1024), 0, s, X, Y, M, N, ld, CT, out_xy, out_x);
        }
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    hipLaunchKernelGGL(colsum2_partial_kernel, dim3(cdiv(N, 64), kSlabs), dim3(256), 0, s, X, Y, M, N, ld, scratch);
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum2_final_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, scratch, N, out_xy, out_x);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_colsum3(const float* X, const float* Y, float* Z, int M, int N, int ld, float* out_xy, float* out_x, float* out_z,
                   float* scratch, const unsigned char* row_flag, int flag_div, void* Zh, hipStream_t s, bool big_scratch) {
    if (N <= 0) return IMMTSF_OK;
    if (colsum_vec_ok(X, Y, Z, Zh, M, N, ld, big_scratch))
        return launch_colsum_vec<3>(X, Y, Z, M, N, ld, out_xy, out_x, out_z, 0, scratch, row_flag, flag_div, Zh, s);
    const int nsl = M >= 1024 ? 64 : kSlabs;
    hipLaunchKernelGGL(colsum3_partial_kernel, dim3(cdiv(N, 64), nsl), dim3(256), 0, s, X, Y, Z, M, N, ld, scratch, row_flag,
                       flag_div > 0 ? flag_div : 1, static_cast<bf16_t*>(Zh));
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum3_final_kernel, dim3(cdiv(N, 64)), dim3(256), 0, s, scratch, N, out_xy, out_x, out_z, nsl);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_layernorm_fwd(const float* x, int rows, int d, const float* gamma, const float* beta, float eps, float* xhat,
                         float* rstd, float* z, DropCfg drop, uint64_t site, hipStream_t s, void* zh, const float* res, DropCfg pre,
                         uint64_t presite, void* xhat_h) {
    if (rows <= 0) return IMMTSF_OK;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(xhat) |
                         reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(res);
    if ((d & 3) == 0 && d <= 256 * LN_DV && (al & 15) == 0 && (reinterpret_cast<uintptr_t>(xhat_h) & 7) == 0)
        hipLaunchKernelGGL(layernorm_fwd_vec_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, d, gamma, beta, eps, xhat, rstd, z,
                           drop, site, static_cast<bf16_t*>(zh), res, pre, presite, static_cast<bf16_t*>(xhat_h));
    else if (res || xhat_h)
        return IMMTSF_EUNSUPPORTED;      // the residual form and the bf16 x_hat image need d % 4 == 0, d <= 1024 and 16-byte aligned rows
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, d, gamma, beta, eps, xhat, rstd, z,
                           drop, site, static_cast<bf16_t*>(zh));
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_layernorm_bwd(float* dz_dy, int rows, int d, const float* gamma, const float* xhat, const float* rstd,
                         float* dx, DropCfg drop, uint64_t site, hipStream_t s, float* dbranch, DropCfg pre, uint64_t presite) {
    if (rows <= 0) return IMMTSF_OK;
    const uintptr_t al = reinterpret_cast<uintptr_t>(dz_dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(xhat) |
                         reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(dbranch);
    if ((d & 3) == 0 && d <= 256 * LN_DV && (al & 15) == 0)
        hipLaunchKernelGGL(layernorm_bwd_vec_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, dz_dy, rows, d, gamma, xhat, rstd, dx, drop,
                           site, dbranch, pre, presite);
    else if (dbranch)
        return IMMTSF_EUNSUPPORTED;
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, dz_dy, rows, d, gamma, xhat, rstd, dx, drop,
                           site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// LayerNorm backward + its parameter-gradient sums in one pass (+ final reduce): out_gw = sum_rows dy * xhat, out_gb = sum_rows dy,
// out_q (optional) = sum_rows dx; row_flag / dxh: see the kernel.  scratch: ln_sums_scratch_floats(d, 3 or 2) floats.
// IMMTSF_EUNSUPPORTED when the vector path does not apply (the caller then runs launch_layernorm_bwd + launch_colsum2/3).
bool ln_sums_compact_ok(int rows, int d) {        // the K = 3 kernel's own limits, for callers that store x_hat / dx as bf16 only
    return (d & 3) == 0 && d <= 256 * LN_DV && rows >= 512 && (size_t)3 * 3 * d * sizeof(float) <= 64 * 1024;
}
int launch_layernorm_bwd_sums(float* dz_dy, int rows, int d, const float* gamma, const float* xhat, const float* rstd, float* dx,
                              DropCfg drop, uint64_t site, float* out_gw, float* out_gb, float* out_q, float* scratch,
                              const unsigned char* row_flag, int flag_div, void* dxh, hipStream_t s, const void* xhat_h) {
    if (rows <= 0) return IMMTSF_OK;
    constexpr bool on = true;
    const uintptr_t al = reinterpret_cast<uintptr_t>(dz_dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(xhat) |
                         reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(scratch);
    if (!on || (d & 3) || d > 256 * LN_DV || (al & 15) || ((reinterpret_cast<uintptr_t>(dxh) | reinterpret_cast<uintptr_t>(xhat_h)) & 7) ||
        rows < 512)
        return IMMTSF_EUNSUPPORTED;
    if ((!xhat && !xhat_h) || (!dx && !dxh)) return IMMTSF_EINVAL;
    // one row per wave up to 2048 workgroups (a wave's rows are a dependent chain: few rows per wave, many waves), then more rows
    // per wave; the partial sums are nsl x K x d floats (ln_sums_scratch_floats)
    int nsl = (rows + 3) / 4;
    nsl = nsl > kLnSlabsMax ? kLnSlabsMax : nsl;
    const size_t lds = (size_t)3 * (out_q ? 3 : 2) * d * sizeof(float);
    if (out_q) {
        if (lds > 64 * 1024) return IMMTSF_EUNSUPPORTED;
        if (xhat_h)
            hipLaunchKernelGGL((layernorm_bwd_sums_kernel<3, true>), dim3(nsl), dim3(256), lds, s, dz_dy, rows, d, gamma, xhat_h, rstd, dx, drop, site,
                               scratch, row_flag, flag_div > 0 ? flag_div : 1, static_cast<bf16_t*>(dxh));
        else
            hipLaunchKernelGGL((layernorm_bwd_sums_kernel<3, false>), dim3(nsl), dim3(256), lds, s, dz_dy, rows, d, gamma, xhat, rstd, dx, drop, site,
                               scratch, row_flag, flag_div > 0 ? flag_div : 1, static_cast<bf16_t*>(dxh));
        IMMTSF_LAUNCH_CHECK();
        hipLaunchKernelGGL((colsum_vec_final_kernel<3>), dim3(cdiv(d, 32), 3), dim3(256), 0, s, scratch, d, nsl, out_gw, out_gb, out_q, 0);
    } else {
        if (lds > 64 * 1024 || row_flag || dxh || xhat_h || !dx) return IMMTSF_EUNSUPPORTED;
        hipLaunchKernelGGL((layernorm_bwd_sums_kernel<2, false>), dim3(nsl), dim3(256), lds, s, dz_dy, rows, d, gamma, xhat, rstd, dx, drop, site, scratch,
                           nullptr, 1, nullptr);
        IMMTSF_LAUNCH_CHECK();
        hipLaunchKernelGGL((colsum_vec_final_kernel<2>), dim3(cdiv(d, 32), 2), dim3(256), 0, s, scratch, d, nsl, out_gw, out_gb, nullptr, 0);
    }
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_instance_norm(const float* x, int B, int L, int C, float* xn, float* means, float* stdev, hipStream_t s) {
    if (B <= 0 || L <= 0 || C <= 0) return IMMTSF_OK;
    const size_t lds = ((size_t)L * C + 2 * (size_t)C) * sizeof(float);
    if (lds > 64 * 1024) return IMMTSF_EUNSUPPORTED;
    hipLaunchKernelGGL(instance_norm_kernel, dim3(B), dim3(256), lds, s, x, B, L, C, xn, means, stdev);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// the low-rank form: dy = G Wb (G: rows x PW, pitch ldg; Wb: PW x d) is never written -- see layernorm_bwd_lr_kernel.  x_hat as the
// bf16 image, dx fp32 and / or bf16; sums and row flags as launch_layernorm_bwd_sums with out_q.  IMMTSF_EUNSUPPORTED outside its limits.
bool ln_lr_ok(int rows, int d, int PW) {
    return (d & 3) == 0 && d <= 1024 && d >= 256 && PW >= 1 && PW <= 64 && rows >= 512 && (size_t)(PW + 3) * d * sizeof(float) <= 150 * 1024;
}
int launch_layernorm_bwd_lr(const float* G, int ldg, int PW, const float* Wb, int rows, int d, const float* gamma, const float* xhat,
                            const void* xhat_h, const float* rstd, float* dx, void* dxh, DropCfg drop, uint64_t site, float* out_gw, float* out_gb,
                            float* out_q, float* scratch, const unsigned char* row_flag, int flag_div, hipStream_t s, const void* keep, int keep_T) {
    if (!ln_lr_ok(rows, d, PW)) return IMMTSF_EUNSUPPORTED;
    if (keep && (keep_T <= 0 || keep_T > 32)) return IMMTSF_EINVAL;
    if (!G || !Wb || (!xhat && !xhat_h) || (!dx && !dxh) || !out_gw || !out_gb || !out_q || !scratch) return IMMTSF_EINVAL;
    const uintptr_t al = reinterpret_cast<uintptr_t>(Wb) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(gamma) |
                         reinterpret_cast<uintptr_t>(scratch) | reinterpret_cast<uintptr_t>(xhat);
    if ((al & 15) || ((reinterpret_cast<uintptr_t>(dxh) | reinterpret_cast<uintptr_t>(xhat_h)) & 7)) return IMMTSF_EUNSUPPORTED;
    const void* xa = xhat_h ? xhat_h : static_cast<const void*>(xhat);
    const size_t lds = (size_t)(PW + 3) * d * sizeof(float);
    const bool many = rows >= 8192;
    const int per_wg = 8 * (many ? (d <= 768 ? 4 : 2) : 1);
    int nsl = cdiv(rows, per_wg);
    nsl = nsl > 256 ? 256 : nsl;          // (one workgroup per CU: the LDS copy of Wb)
    const int fd = flag_div > 0 ? flag_div : 1;
#define LNLR(RR, DVV, XHH)                                                                                                                       \
    do {                                                                                                                                         \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(layernorm_bwd_lr_kernel<RR, DVV, XHH>),                 \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);                              \
        if (attr != hipSuccess) return (int)attr;                                                                                                \
        hipLaunchKernelGGL((layernorm_bwd_lr_kernel<RR, DVV, XHH>), dim3(nsl), dim3(512), lds, s, G, ldg, PW, Wb, rows, d, gamma, xa, rstd, dx,   \
                           static_cast<bf16_t*>(dxh), drop, site, scratch, row_flag, fd, static_cast<const unsigned long long*>(keep), keep_T); \
    } while (0)
#define LNLR2(RR, DVV) do { if (xhat_h) LNLR(RR, DVV, true); else LNLR(RR, DVV, false); } while (0)
    if (d <= 768) { if (many) LNLR2(4, 3); else LNLR2(1, 3); }
    else { if (many) LNLR2(2, 4); else LNLR2(1, 4); }       // (four chunks per lane: two rows at a time fit the registers)
#undef LNLR2
#undef LNLR
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL((colsum_vec_final_kernel<3>), dim3(cdiv(d, 32), 3), dim3(256), 0, s, scratch, d, nsl, out_gw, out_gb, out_q, 0);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_matvec(const float* W, int ldw, const float* x, const float* b, int rows, int cols, float* y, float* ys,
                  float scale, hipStream_t s, float* y_nobias) {
    hipLaunchKernelGGL(matvec_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, W, ldw, x, b, rows, cols, y, ys, scale, y_nobias);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_matvec_t(const float* W, int ldw, const float* x, int rows, int cols, float* y, int accumulate,
                    hipStream_t s) {
    // 12 workgroups cannot stream a 768x768 matrix quickly: when the result accumulates anyway, spread the rows over
    // slabs and add with atomics
    const int slabs = (accumulate && rows >= 256) ? 16 : 1;
    hipLaunchKernelGGL(matvec_t_kernel, dim3(cdiv(cols, 64), slabs), dim3(1024), 0, s, W, ldw, x, rows, cols, y, accumulate);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_outer(const float* a, const float* b, int rows, int cols, float* out, int ld, hipStream_t s) {
    const long n = (long)rows * cols;
    hipLaunchKernelGGL(outer_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, rows, cols, out, ld);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_query_bwd(const float* dqs_part, int B, int d, float scale, const float* Wq, int ldw, const float* Q, float* dWq, int ldg,
                     float* dbq, float* dQ, hipStream_t s) {
    if (d <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(query_bwd_kernel, dim3(cdiv(d, QB_ROWS)), dim3(256), 0, s, dqs_part, B, d, scale, Wq, ldw, Q,
                       dWq, ldg, dbq, dQ);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// launch_query_bwd + launch_time2vec_bwd (general two-stage form) with their first kernels merged
int launch_query_t2v_bwd(const float* dqs_part, int B, int d, float scale, const float* Wq, int ldw, const float* Q, float* dWq, int ldg,
                         float* dbq, float* dQ, const float* tau_pad, const int* rowmap, const int* total, int max_rows, int d_tau,
                         const float* w, const float* b, const float* dfeat, int ld, float* dw0, float* db0, float* dw, float* db,
                         float* scratch, int nslabs, hipStream_t s, int accumulate) {
    if (d <= 0 || max_rows <= 0) return IMMTSF_EINVAL;
    if (nslabs < 1) nslabs = kSlabs;
    const int nq = cdiv(d, QB_ROWS), gx = cdiv(d_tau, 64);
    const QueryBwdJob q{dqs_part, B, d, scale, Wq, ldw, Q, dWq, ldg, dbq, dQ};
    const T2VBwdJob t{tau_pad, rowmap, total, d_tau, w, b, dfeat, ld, scratch, max_rows, gx, nslabs};
    hipLaunchKernelGGL(query_t2v_bwd_kernel, dim3(nq + gx * nslabs), dim3(256), 0, s, q, nq, t);
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL(time2vec_bwd_final_kernel, dim3(cdiv(d_tau, 4)), dim3(256), 0, s, scratch, d_tau, dw0, db0, dw, db, nslabs, accumulate);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// ---- the small reductions of MMF_XAttn_Add's parameter gradients (C <= 32 series variables), two launches instead of six:
// head_sums:  workgroup 0: LayerNorm(C)'s (sum dn*xhat, sum dn) over the rows; workgroup 1: the column sums of ddelta over the rows
//             of windows WITH text (s_live) -- the row mask applied while reading, ddelta stays as it is
// head_outer: d b_out[j] = sum_c W_res[c][j] s_live[c] and dW_res[c][j] = s_live[c] b_out[j], one thread per column j
// (before: colsum2, mask_rows, colsum, matvec_t, outer as five launches on the chain in front of the backbone's backward)
__global__ __launch_bounds__(1024) void head_sums_kernel(const float* __restrict__ dn, const float* __restrict__ xhat,
                                                          const float* __restrict__ ddelta, const unsigned char* __restrict__ flag,
                                                          int flag_div, int M, int C, int CT, float* __restrict__ out_xy,
                                                          float* __restrict__ out_x, float* __restrict__ slive) {
    __shared__ float ra[16 * 32], rb[16 * 32];
    const int RT = 1024 / CT, tx = threadIdx.x % CT, ty = threadIdx.x / CT, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a = 0.f, b = 0.f;
    if (blockIdx.x == 0) {
        if (tx < C) {
#pragma unroll 8
            for (int r = ty; r < M; r += RT) {
                const float x = dn[(size_t)r * C + tx];
                a = fmaf(x, xhat[(size_t)r * C + tx], a);
                b += x;
            }
        }
    } else if (tx < C) {
#pragma unroll 8
        for (int r = ty; r < M; r += RT) {
            const float x = ddelta[(size_t)r * C + tx];
            if (flag[r / flag_div]) a += x;
        }
    }
    a = coset_sum(a, CT);
    b = coset_sum(b, CT);
    if (lane < CT) { ra[wave * CT + lane] = a; rb[wave * CT + lane] = b; }
    __syncthreads();
    if (threadIdx.x < C) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { ta += ra[w * CT + threadIdx.x]; tb += rb[w * CT + threadIdx.x]; }
        if (blockIdx.x == 0) { out_xy[threadIdx.x] = ta; out_x[threadIdx.x] = tb; }
        else slive[threadIdx.x] = ta;
    }
}
__global__ __launch_bounds__(256) void head_outer_kernel(const float* __restrict__ Wres, int d, const float* __restrict__ slive,
                                                          const float* __restrict__ bout, int C, float* __restrict__ dbout,
                                                          float* __restrict__ dWres) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    const float bj = bout[j];
    float a = 0.f;
    for (int c = 0; c < C; ++c) {
        const float sl = slive[c];
        a = fmaf(Wres[(size_t)c * d + j], sl, a);
        dWres[(size_t)c * d + j] = sl * bj;
    }
    dbout[j] = a;
}
bool head_sums_supported(int M, int C) { return C >= 1 && C <= 32 && (long)M * C <= (1L << 17); }
int launch_head_sums(const float* dn, const float* xhat, const float* ddelta, const unsigned char* flag, int flag_div, int M, int C,
                     float* out_xy, float* out_x, float* slive, hipStream_t s) {
    if (!head_sums_supported(M, C) || flag_div <= 0) return IMMTSF_EUNSUPPORTED;
    int CT = 1;
    while (CT < C) CT <<= 1;
    hipLaunchKernelGGL(head_sums_kernel, dim3(2), dim3(1024), 0, s, dn, xhat, ddelta, flag, flag_div, M, C, CT, out_xy, out_x, slive);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int launch_head_outer(const float* Wres, int d, const float* slive, const float* bout, int C, float* dbout, float* dWres, hipStream_t s) {
    hipLaunchKernelGGL(head_outer_kernel, dim3(cdiv(d, 256)), dim3(256), 0, s, Wres, d, slive, bout, C, dbout, dWres);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_axpy(const float* src, float alpha, float* dst, int n, int accumulate, hipStream_t s) {
    if (n <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, src, alpha, dst, n, accumulate);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_fill(float* dst, float v, size_t n, hipStream_t s) {
    if (n == 0) return IMMTSF_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, dst, v, n);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_dropout_mask(uint64_t seed, uint64_t site, size_t n, float p, unsigned char* out, hipStream_t s) {
    if (n == 0) return IMMTSF_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, s, seed, site, n, p, out);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
