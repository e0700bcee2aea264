// Backward of a SMALL linear layer y = x W^T + b as one launch: dx (M, K) = dy W (x the ReLU mask of the layer's input), and
// dW (N, K) += dy^T x, db (N) += dy^T 1 by atomics into gradient buffers that read zero (FlatTrainer's sinks; LinearFn's own
// zero-filled allocation).  tPatchGNN's temporal aggregation (models/tPatchGNN.py:236-241: Linear(M * hid, hid) on B * N rows:
// 512 x 64 -> 32 at the benchmark shape) was two GEMM launches of 8 and 4 workgroups, 9.5 + 12.4 us on the backbone's dependent
// backward chain for 4 MFLOP.  Exact fp32 (plain v_fma) in both precision modes.
//
// A workgroup takes ROWS rows (64, or 16 when 64-row groups would leave most of the chip idle: 512 rows were 8 workgroups and 23 us
// on the backbone's dependent backward chain): dy (ROWS x N), x (ROWS x K) and W (N x K) sit in LDS; a thread owns ROWS / 16 rows of
// a column quarter of dx (N FMAs per element) and NK / 256 elements of dW (ROWS FMAs each), which it adds to global memory with one
// atomic per element.
#include "gemm.hpp"
#include "../../include/immtsf.h"
#include "common.hpp"

namespace {

constexpr int LS_NMAX = 32, LS_KMAX = 64;

template <int LS_ROWS>
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ dy, int M, int N, int K, float* __restrict__ dx,
                                                                const float* __restrict__ relu_x, float* __restrict__ dW,
                                                                float* __restrict__ db) {
    __shared__ float dys[LS_ROWS][LS_NMAX + 1], xs[LS_ROWS][LS_KMAX + 1], Ws[LS_NMAX][LS_KMAX + 1];
    const int tid = threadIdx.x, r0 = blockIdx.x * LS_ROWS, rows = min(LS_ROWS, M - r0);
    for (int i = tid; i < LS_ROWS * N; i += 256) {
        const int r = i / N, n = i - r * N;
        dys[r][n] = r < rows ? dy[(size_t)(r0 + r) * N + n] : 0.f;
    }
    if (dW)
        for (int i = tid; i < LS_ROWS * K; i += 256) {
            const int r = i / K, k = i - r * K;
            xs[r][k] = r < rows ? x[(size_t)(r0 + r) * K + k] : 0.f;
        }
    if (dx)
        for (int i = tid; i < N * K; i += 256) Ws[i / K][i % K] = W[i];
    __syncthreads();
    if (dx) {       // 64 x K outputs: thread -> rows rq .. rq + 3 (stride 16), columns kq, kq + 16, ..
        constexpr int RJ = LS_ROWS / 16;
        const int kq = tid & 15, rq = tid >> 4;
        for (int k = kq; k < K; k += 16) {
            float a[RJ];
#pragma unroll
            for (int j = 0; j < RJ; ++j) a[j] = 0.f;
            for (int n = 0; n < N; ++n) {
                const float w = Ws[n][k];
#pragma unroll
                for (int j = 0; j < RJ; ++j) a[j] = fmaf(dys[rq + 16 * j][n], w, a[j]);
            }
#pragma unroll
            for (int j = 0; j < RJ; ++j) {
                const int r = rq + 16 * j;
                if (r < rows) {
                    const size_t o = (size_t)(r0 + r) * K + k;
                    dx[o] = (relu_x && relu_x[o] <= 0.f) ? 0.f : a[j];
                }
            }
        }
    }
    if (dW) {
        for (int i = tid; i < N * K; i += 256) {
            const int n = i / K, k = i - n * K;
            float a = 0.f;
#pragma unroll 8
            for (int r = 0; r < LS_ROWS; ++r) a = fmaf(dys[r][n], xs[r][k], a);
            atomicAdd(dW + i, a);
        }
        if (db && tid < N) {
            float a = 0.f;
            for (int r = 0; r < LS_ROWS; ++r) a += dys[r][tid];
            atomicAdd(db + tid, a);
        }
    }
}

}  // namespace

bool linear_small_ok(int M, int N, int K) { return N >= 1 && N <= LS_NMAX && K >= 1 && K <= LS_KMAX && M >= 1 && M <= 4096; }

// dW / db (may be null together) must read zero (or hold a running sum); dx (may be null) is overwritten
int launch_linear_small_bwd(const float* x, const float* W, const float* dy, int M, int N, int K, float* dx, const float* relu_x, float* dW,
                            float* db, hipStream_t s) {
    if (M <= 2048)       // few rows: 16-row groups (32 workgroups at the benchmark's 512 rows instead of 8)
        hipLaunchKernelGGL(linear_small_bwd_kernel<16>, dim3(cdiv(M, 16)), dim3(256), 0, s, x, W, dy, M, N, K, dx, relu_x, dW, db);
    else
        hipLaunchKernelGGL(linear_small_bwd_kernel<64>, dim3(cdiv(M, 64)), dim3(256), 0, s, x, W, dy, M, N, K, dx, relu_x, dW, db);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
