// TimesNet's Inception block (reference layers/Conv_Blocks.py:5-31: the MEAN of num_kernels same-padded 2-D convolutions
// with kernel sizes 1, 3, ..., 2 n - 1) as ONE convolution on the MFMA GEMM.
//
// Convolution is linear in the kernel, so mean_i conv(x, W_i) = conv(x, W_eff) with W_eff = (1/n) sum_i W_i zero-padded
// to the largest size KS = 2 n - 1 (immtsf_inception_merge; its backward, immtsf_inception_unmerge, hands every W_i the
// centre crop of dW_eff / n).  The merged convolution runs channels-last -- TimesBlock's (B, length, d_model) activations
// ARE channels-last images (B, length / period, period, d_model), so the reference's permutes disappear -- as
//   im2col:  col[(b,h,w), (dy,dx,ci)] = x[b, h+dy-r, w+dx-r, ci]   (zero outside; r = KS / 2)
//   GEMM  :  y[(b,h,w), co] = act(col W_eff^T + b_eff)              (bias / GELU in the epilogue, pre-activation kept)
// and backward as dW_eff = dz^T col + db_eff (GEMM with the bias-gradient reduction) and dx = the SAME im2col + GEMM
// convolution applied to dz with the kernel flipped and its channel roles swapped.  Per period that is 2 launches forward
// and 4-5 backward per convolution instead of 6 MIOpen convolutions + stack + mean (and their 12+ backward kernels).
#include "../../include/immtsf.h"
#include "block_util.hpp"
#include "rowops.hpp"

namespace {

struct ConvDims { int B, H, W, C, KS; };

// grid rows = B*H*W; 256 threads walk the K = KS*KS*C columns of one row (C contiguous floats per tap)
__global__ __launch_bounds__(256) void im2col_cl_kernel(ConvDims d, const float* __restrict__ x, float* __restrict__ col) {
    const int row = blockIdx.x, w = row % d.W, h = (row / d.W) % d.H, b = row / (d.W * d.H), r = d.KS >> 1;
    const int K = d.KS * d.KS * d.C;
    float* out = col + (size_t)row * K;
    for (int k = threadIdx.x; k < K; k += 256) {
        const int ci = k % d.C, tap = k / d.C, dx = tap % d.KS, dy = tap / d.KS;
        const int hh = h + dy - r, ww = w + dx - r;
        out[k] = (hh >= 0 && hh < d.H && ww >= 0 && ww < d.W) ? x[(((size_t)b * d.H + hh) * d.W + ww) * d.C + ci] : 0.f;
    }
}

// ---- the same convolution on TimesNet's period images with the period ON THE DEVICE (reference models/TimesNet.py:9-18, 44-62: the top-k
// periods are read on the host and decide the image shapes -- two host syncs per step).  Images live "position-major": row l * B + b of
// a (Lmax * B, C) matrix is position l of window b, so the rows of an image of any length are a PREFIX of the buffer, and the period
// only enters as numbers the kernels read from device memory: rows = length * B valid rows (length = the series length rounded up to a
// multiple of the period), image height = length / period, width = period.  Rows beyond `rows` are never read or written; the GEMMs
// take their row count (NT / NN) or reduction length (TN) from the same device word.
__global__ void period_rows_kernel(const long long* __restrict__ top, int k, int total, int B, int* __restrict__ period, int* __restrict__ rows) {
    const int j = threadIdx.x;
    if (j >= k) return;
    const int f = (int)top[j];
    const int p = f > 0 ? total / f : total;
    const int length = (total % p == 0) ? total : (total / p + 1) * p;
    period[j] = p;
    rows[j] = length * B;
}
constexpr int IMP_ROWS = 4;       // rows per workgroup (one row per workgroup was dispatch-bound: 8192 workgroups of 4 KB each, 20 us)
__device__ __forceinline__ void imp_store(float* o, const float (&v)[8], int n) {
    if (n == 8) { reinterpret_cast<float4*>(o)[0] = make_float4(v[0], v[1], v[2], v[3]); reinterpret_cast<float4*>(o)[1] = make_float4(v[4], v[5], v[6], v[7]); }
    else o[0] = v[0];
}
__device__ __forceinline__ void imp_store(bf16_t* o, const float (&v)[8], int n) {
    if (n == 8) {
        typedef bf16_t bf16x8_t __attribute__((ext_vector_type(8)));
        bf16x8_t h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (bf16_t)v[j];
        *reinterpret_cast<bf16x8_t*>(o) = h;
    } else o[0] = (bf16_t)v[0];
}
// an item = CW channels of one tap of one row (CW = 8 when the channel count is a multiple of 8: two 16-byte loads, one 16- or 32-byte
// store; else 1); a workgroup's IMP_ROWS rows are one flat item range
// (grid.y: the period images of a batched call -- image z reads x + z xs, writes col + z cs, takes period[z] / rows[z])
template <typename OT, int CW>
__global__ __launch_bounds__(256) void im2col_period_kernel(int B, int C, int KS, const int* __restrict__ period, const int* __restrict__ rows,
                                                             const float* __restrict__ x, OT* __restrict__ col, long xs, long cs) {
    const int z = blockIdx.y;
    x += (size_t)z * xs;
    col += (size_t)z * cs;
    const int nrows = rows[z], p = period[z], H = (nrows / B) / p, r = KS >> 1, K = KS * KS * C;
    const int cpt = C / CW, per_row = KS * KS * cpt, row0 = blockIdx.x * IMP_ROWS;
    const int nr = min(IMP_ROWS, nrows - row0);
    for (int it = threadIdx.x; it < nr * per_row; it += 256) {
        const int rr = it / per_row, q = it - rr * per_row, tap = q / cpt, cc = (q - tap * cpt) * CW;
        const int row = row0 + rr, l = row / B, b = row - l * B, h = l / p, w = l - h * p;
        const int dy = tap / KS, dx = tap - dy * KS, hh = h + dy - r, ww = w + dx - r;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (hh >= 0 && hh < H && ww >= 0 && ww < p) {
            const float* src = x + ((size_t)(hh * p + ww) * B + b) * C + cc;
            if (CW == 8) {
                const float4 a = reinterpret_cast<const float4*>(src)[0], c = reinterpret_cast<const float4*>(src)[1];
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
            } else v[0] = src[0];
        }
        imp_store(col + (size_t)row * K + tap * C + cc, v, CW);
    }
}
template <typename OT>
static void launch_im2col_period(int B, int C, int KS, const int* period, const int* rows, const float* x, OT* col, int max_rows, hipStream_t s,
                                 int k = 1, long xs = 0, long cs = 0) {
    const dim3 grid(cdiv(max_rows, IMP_ROWS), k);
    if ((C & 7) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(col) & 15) == 0 && (xs & 3) == 0 && (cs & 7) == 0)
        hipLaunchKernelGGL((im2col_period_kernel<OT, 8>), grid, dim3(256), 0, s, B, C, KS, period, rows, x, col, xs, cs);
    else
        hipLaunchKernelGGL((im2col_period_kernel<OT, 1>), grid, dim3(256), 0, s, B, C, KS, period, rows, x, col, xs, cs);
}
__device__ __forceinline__ float pg_gelu_grad(float z) { return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z); }
// the gradient entering a batched period convolution's backward: g0 = dy (act == 0) or dy gelu'(z_pre), as fp32 (dz, act != 0 only: what
// the im2col of the data gradient reads) and as the bf16 image the weight-gradient product reads; rows beyond an image's are skipped
__global__ __launch_bounds__(256) void period_g0_kernel(const float* __restrict__ dy, const float* __restrict__ z_pre, float* __restrict__ dz,
                                                         bf16_t* __restrict__ g16, int C, long per, const int* __restrict__ rows) {
    const int z = blockIdx.y;
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (long)rows[z] * C) return;
    const size_t at = (size_t)z * per + i;
    float4 v = *reinterpret_cast<const float4*>(dy + at);
    if (z_pre) {
        const float4 q = *reinterpret_cast<const float4*>(z_pre + at);
        v = make_float4(v.x * pg_gelu_grad(q.x), v.y * pg_gelu_grad(q.y), v.z * pg_gelu_grad(q.z), v.w * pg_gelu_grad(q.w));
        *reinterpret_cast<float4*>(dz + at) = v;
    }
    const bf16x4 h = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
    *reinterpret_cast<bf16x4*>(g16 + at) = h;
}
// dx[row, :] = sum over the k images that hold the row (row < rows[z]) of dxk[z, row, :]   (one input shared by every period image)
__global__ __launch_bounds__(256) void period_sum_kernel(const float* __restrict__ dxk, int k, long per, const int* __restrict__ rows, int C,
                                                          float* __restrict__ dx) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= per) return;
    const int row = (int)(i / C);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < k; ++z) {
        if (row < rows[z]) {
            const float4 v = *reinterpret_cast<const float4*>(dxk + (size_t)z * per + i);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    *reinterpret_cast<float4*>(dx + i) = a;
}
__global__ __launch_bounds__(256) void gelu_rows_kernel(const float* __restrict__ z, float* __restrict__ y, int Cout, const int* __restrict__ rows) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x, n = (long)*rows * Cout;
    if (i < n) { const float v = z[i]; y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
}
__global__ __launch_bounds__(256) void gelu_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ z, float* __restrict__ dz, int Cout,
                                                             const int* __restrict__ rows);

__device__ __forceinline__ float gelu_grad(float z) {
    return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z, float* __restrict__ dz, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dz[i] = dy[i] * gelu_grad(z[i]);
}

__global__ __launch_bounds__(256) void gelu_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ z, float* __restrict__ dz, int Cout,
                                                             const int* __restrict__ rows) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x, n = (long)*rows * Cout;
    if (i < n) dz[i] = dy[i] * gelu_grad(z[i]);
}

// ---- TimesBlock's adaptive aggregation over the k period images (reference models/TimesNet.py:80-86: stack, softmax weights, sum, residual)
// on the position-major images: out[b, t, :] = x[b, t, :] + sum_j w[b, j] Y[j, t B + b, :],  t < total
__global__ __launch_bounds__(256) void period_agg_fwd_kernel(const float* __restrict__ Y, const float* __restrict__ w, const float* __restrict__ x,
                                                              int B, int total, long per, int N, int k, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * total * N) return;
    const int n = (int)(i % N);
    const long bt = i / N;
    const int t = (int)(bt % total), b = (int)(bt / total);
    float a = x[i];
    for (int j = 0; j < k; ++j) a = fmaf(w[b * k + j], Y[(size_t)j * per + ((size_t)t * B + b) * N + n], a);
    out[i] = a;
}
// dY[j, t B + b, :] = w[b, j] dout[b, t, :] for t < total, 0 for the rows the crop dropped (every row up to Lmax: the images' own rows end
// somewhere in between)
__global__ __launch_bounds__(256) void period_agg_bwd_y_kernel(const float* __restrict__ w, const float* __restrict__ dout, int B, int total, int Lmax,
                                                                int N, int k, float* __restrict__ dY) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x, per = (long)Lmax * B * N;
    if (i >= per * k) return;
    const int j = (int)(i / per);
    const long r = i - (long)j * per;
    const int n = (int)(r % N);
    const long lb = r / N;
    const int b = (int)(lb % B), t = (int)(lb / B);
    dY[i] = t < total ? w[b * k + j] * dout[((size_t)b * total + t) * N + n] : 0.f;
}
// dw[b, j] = sum_{t, n} dout[b, t, n] Y[j, t B + b, n]: one workgroup per window
__global__ __launch_bounds__(256) void period_agg_bwd_w_kernel(const float* __restrict__ Y, const float* __restrict__ dout, int B, int total, long per,
                                                                int N, int k, float* __restrict__ dw) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    for (int j = 0; j < k; ++j) {
        float a = 0.f;
        for (int x = threadIdx.x; x < total * N; x += 256) {
            const int t = x / N, n = x - t * N;
            a = fmaf(dout[((size_t)b * total + t) * N + n], Y[(size_t)j * per + ((size_t)t * B + b) * N + n], a);
        }
        a = block_sum(a, red);
        if (threadIdx.x == 0) dw[b * k + j] = a;
    }
}

struct KernelPtrs { const float* w[IMMTSF_INCEPTION_MAX]; const float* b[IMMTSF_INCEPTION_MAX]; };
struct KernelGrads { float* w[IMMTSF_INCEPTION_MAX]; float* b[IMMTSF_INCEPTION_MAX]; };

// W_eff[co][(dy,dx,ci)] = (1/n) sum_i W_i[co][ci][dy-r+i][dx-r+i] over the kernels that reach (dy,dx); b_eff = mean b_i
__global__ __launch_bounds__(256) void inception_merge_kernel(int n, int Cin, int Cout, KernelPtrs p, float* __restrict__ Weff,
                                                               float* __restrict__ beff) {
    const int KS = 2 * n - 1, r = n - 1, K = KS * KS * Cin;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long)Cout * K) {
        const int co = (int)(i / K), k = (int)(i % K), ci = k % Cin, tap = k / Cin, dx = tap % KS, dy = tap / KS;
        float a = 0.f;
        for (int j = 0; j < n; ++j) {
            const int s = 2 * j + 1, yy = dy - r + j, xx = dx - r + j;
            if (yy >= 0 && yy < s && xx >= 0 && xx < s) a += p.w[j][(((size_t)co * Cin + ci) * s + yy) * s + xx];
        }
        Weff[i] = a / (float)n;
    }
    if (i < Cout) {
        float a = 0.f;
        for (int j = 0; j < n; ++j) a += p.b[j][i];
        beff[i] = a / (float)n;
    }
}

// dW_j[co][ci][yy][xx] = dW_eff[co][((yy-j+r)*KS + (xx-j+r))*Cin + ci] / n ; db_j = db_eff / n   (written)
__global__ __launch_bounds__(256) void inception_unmerge_kernel(int n, int Cin, int Cout, const float* __restrict__ dWeff,
                                                                 const float* __restrict__ dbeff, KernelGrads g) {
    const int KS = 2 * n - 1, r = n - 1, K = KS * KS * Cin, j = blockIdx.y, s = 2 * j + 1;
    const long i = (long)blockIdx.x * 256 + threadIdx.x, cnt = (long)Cout * Cin * s * s;
    const float inv = 1.f / (float)n;
    if (i < cnt) {
        const int xx = (int)(i % s), yy = (int)((i / s) % s), ci = (int)((i / ((long)s * s)) % Cin), co = (int)(i / ((long)s * s * Cin));
        g.w[j][i] = dWeff[(size_t)co * K + ((yy - j + r) * KS + (xx - j + r)) * Cin + ci] * inv;
    }
    if (i < Cout) g.b[j][i] = dbeff[i] * inv;
}

// the data gradient of a same-padded convolution is the convolution of dz with the kernel flipped in both directions and
// its channel roles swapped: Wf[ci][(dy,dx,co)] = W_eff[co][(KS-1-dy, KS-1-dx, ci)]
template <typename OT>
__global__ __launch_bounds__(256) void flip_weight_kernel(int Cin, int Cout, int KS, const float* __restrict__ Weff, OT* __restrict__ Wf) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x, K2 = (long)KS * KS * Cout;
    if (i >= (long)Cin * K2) return;
    const int ci = (int)(i / K2), k = (int)(i % K2), co = k % Cout, tap = k / Cout, dx = tap % KS, dy = tap / KS;
    Wf[i] = (OT)Weff[(size_t)co * KS * KS * Cin + ((KS - 1 - dy) * KS + (KS - 1 - dx)) * Cin + ci];
}

// ---- a period image's convolution WITHOUT its im2col image (bf16 mode).  A period image is small -- at most Lmax positions x C channels:
// 8 KB as bf16 at TimesNet's cfg4 shape -- so a workgroup keeps the whole image of one (window, period) in LDS and forms the rows of the
// im2col matrix as MFMA operands straight from it: lane (row r, k-group q) reads the 8 channels of ONE tap of position r, 16 contiguous
// bytes of LDS, or zeros outside the image (same padding).  The kernel matrix (Cout x KS KS C, bf16, 124 KB: L2-resident) is read as B
// fragments, one k-step ahead.  Per convolution that replaces a 161 MB bf16 image written by im2col and read back by the product (both
// HBM-bound: 39 + 44 us) with ~2 000 MFMAs per workgroup.  grid (B, k images), 256 threads: wave w owns row tiles w, w + 4 (Lmax <= 128)
// and every column tile (Cout <= 64).  Output y[z][(l B + b) Cout + n] (+ z_pre, GELU) for the image's rows only.
constexpr int CPM_CT = 4;      // column tiles (of 16) at most
constexpr int CPM_ITEMS = 2048;  // 16-byte pieces of the kernel matrix a chunk holds: eight per thread
// NCT column tiles and RT row tiles (the whole image: 4 / 5 / 6 / 8 for <= 64 / 80 / 96 / 128 positions) are compile-time numbers; the FOUR
// WAVES SPLIT THE REDUCTION -- wave w takes the k-steps ks = w (mod 4) for every row tile and the partial accumulators meet in LDS at the
// end -- so that a k-step's tap arithmetic and B fragment serve RT A fragments instead of one or two (with a wave per row tile the loop
// was ~100 instructions a k-step on one wave per SIMD, mostly address arithmetic: 33 - 40 us a launch; this form 18 - 22).  The kernel
// matrix comes through LDS in chunks of CPM_ITEMS 16-byte pieces, each fetched into registers while the previous one is multiplied (the
// first before the period and the image are known); operand rows outside the image read a 16-byte block of zeros instead of branching.
// Measured inside one workgroup (device clock, C = 32 -> 16, 11 x 11 taps): image + first chunk 4 us, k loop 7 - 9 us, reduction and
// epilogue 1.5 us -- 8 MFMAs and ~80 other instructions a k-step per wave, on one wave per SIMD (320 workgroups: nothing to overlap with).
template <int NCT, int RT>
__device__ __forceinline__ void cpm_body(int B, int Lmax, int C, int KS, int p, int len, int b, const float* __restrict__ x,
                                         const bf16_t* __restrict__ W16, const float* __restrict__ bias, int Cout, int act,
                                         float* __restrict__ zpre, float* __restrict__ y, int ks_inv, int c8_shift, unsigned char* lds,
                                         bf16x8 (&wv)[8]) {
    constexpr int ncol = NCT * 16, KC = CPM_ITEMS / (ncol * 4), KC4 = KC * 4, pitch = KC * 32 + 32;
    bf16_t* img = reinterpret_cast<bf16_t*>(lds);                            // [Lmax][C], then a 16-byte block of zeros
    bf16_t* wl = img + (size_t)Lmax * C + 8;                                 // one chunk of the kernel matrix / at the end the partial tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int rad = KS >> 1, taps = KS * KS, c8 = C >> 3, nch = taps * c8, K = taps * C, zoff = Lmax * C;
    const bf16x8 zero8 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (tid < 8) img[zoff + tid] = (bf16_t)0.f;
    {   // the image, cast once: every load of a pair of passes in flight before the first store
        const int items = len * c8;
        for (int base = tid; base < items; base += 512) {
            float4 lo[2], hi[2];
            int at[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = base + u * 256, ii = i < items ? i : 0, l = ii / c8, cc = (ii - l * c8) * 8;
                const float* src = x + ((size_t)l * B + b) * C + cc;
                lo[u] = reinterpret_cast<const float4*>(src)[0];
                hi[u] = reinterpret_cast<const float4*>(src)[1];
                at[u] = i < items ? l * C + cc : -1;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bf16x8 h = {(bf16_t)lo[u].x, (bf16_t)lo[u].y, (bf16_t)lo[u].z, (bf16_t)lo[u].w,
                                  (bf16_t)hi[u].x, (bf16_t)hi[u].y, (bf16_t)hi[u].z, (bf16_t)hi[u].w};
                if (at[u] >= 0) *reinterpret_cast<bf16x8*>(img + at[u]) = h;
            }
        }
    }
    // a tap (dy, dx) of output position r reads position r + dy * p + dx when its column ww + dx stays inside the period and that
    // position is inside the image (the column test makes "row inside" and "position inside" the same statement); anything else reads
    // the zero block.  Rows r >= len of the last tile read whatever is valid for them and are never stored.
    int ww0[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int r = i * 16 + fr;
        ww0[i] = r - (r / p) * p;
    }
    f32x4 acc[RT][NCT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nks = (nch + 3) >> 2;
    for (int k0 = 0; k0 < nks; k0 += KC) {
        const int kc = min(KC, nks - k0);
        // this chunk waits in registers (the kernel fetched the first before it knew its image; the next is fetched below, before this
        // one is multiplied): thread t holds the 16-byte pieces t, t + 256, ... of the chunk's ncol x KC4 grid
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256, n = i / KC4, qq = i - n * KC4;
            *reinterpret_cast<bf16x8*>(wl + (size_t)n * pitch + qq * 8) = (n < Cout && k0 * 4 + qq < nch) ? wv[u] : zero8;
        }
        __syncthreads();                                  // the image and the chunk are in place
        if (k0 + KC < nks) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = tid + u * 256, n = i / KC4, q = (k0 + KC) * 4 + (i - n * KC4);
                wv[u] = *reinterpret_cast<const bf16x8*>(W16 + (size_t)min(n, Cout - 1) * K + (size_t)min(q, nch - 1) * 8);   // (masked at the store)
            }
        }
        for (int ks = wave; ks < kc; ks += 4) {           // this wave's share of the reduction
            // the taps of a k-step are wave-uniform numbers: with C >= 32 its four 8-channel chunks belong to ONE tap, with C = 16 to
            // two, with C = 8 to four -- a lane picks its tap's (dy, dx, source offset) by its chunk index
            const int q0 = (k0 + ks) * 4;
            int dxl, posl, chl;                            // tap column offset, position offset dy * p + dx, channel offset (elements)
            if (c8_shift >= 2) {
                const int tap = q0 >> c8_shift, dyq = (tap * ks_inv) >> 16;
                dxl = tap - dyq * KS - rad; posl = (dyq - rad) * p + dxl; chl = ((q0 & (c8 - 1)) + fq) * 8;
            } else if (c8_shift >= 0) {
                const int per = 4 >> c8_shift, sel = fq >> c8_shift;
                dxl = posl = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u < per) {
                        const int tap = (q0 >> c8_shift) + u, dyq = (tap * ks_inv) >> 16, dx = tap - dyq * KS - rad;
                        if (sel == u) { dxl = dx; posl = (dyq - rad) * p + dx; }
                    }
                }
                chl = (fq & (c8 - 1)) * 8;
            } else {
                const int q = q0 + fq, tap = q / c8, dyq = (tap * ks_inv) >> 16;
                dxl = tap - dyq * KS - rad; posl = (dyq - rad) * p + dxl; chl = (q - tap * c8) * 8;
            }
            posl += fr;
            bf16x8 bc[NCT], av[RT];
#pragma unroll
            for (int j = 0; j < NCT; ++j) bc[j] = *reinterpret_cast<const bf16x8*>(wl + (size_t)(j * 16 + fr) * pitch + (ks * 4 + fq) * 8);
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int src = i * 16 + posl;
                const bool ok = (unsigned)(ww0[i] + dxl) < (unsigned)p && (unsigned)src < (unsigned)len;
                av[i] = *reinterpret_cast<const bf16x8*>(img + (ok ? __mul24(src, C) + chl : zoff));
            }
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < NCT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bc[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                  // the chunk has been read
    }
    // the four waves' partial tiles meet in LDS; wave w finishes the row tiles i = w (mod 4)
    f32x4* red = reinterpret_cast<f32x4*>(wl);            // [wave][RT][NCT][64 lanes]
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) red[((wave * RT + i) * NCT + j) * 64 + lane] = acc[i][j];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        if ((i & 3) != wave || i * 16 >= len) continue;    // (wave-uniform)
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            f32x4 t = red[((0 * RT + i) * NCT + j) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) { const f32x4 u = red[((w * RT + i) * NCT + j) * 64 + lane]; t += u; }
            const int n = j * 16 + fr;
            if (n >= Cout) continue;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {                  // C / D map: col = fr, row = fq * 4 + e
                const int r = i * 16 + fq * 4 + e;
                if (r >= len) continue;
                float v = t[e] + bv;
                const size_t at = ((size_t)r * B + b) * Cout + n;
                if (zpre) zpre[at] = v;
                if (act == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
                y[at] = v;
            }
        }
    }
}
template <int NCT>
__global__ __launch_bounds__(256) void conv_period_mfma_kernel(int B, int Lmax, int C, int KS, const int* __restrict__ period,
                                                                const int* __restrict__ rows, const float* __restrict__ x, long xs,
                                                                const bf16_t* __restrict__ W16, const float* __restrict__ bias, int Cout, int act,
                                                                float* __restrict__ zpre, float* __restrict__ y, long ys, int ks_inv, int c8_shift) {
    // ks_inv: tap / KS == (tap * ks_inv) >> 16 for every tap (checked by the launcher); c8_shift: log2(C / 8) or -1
    extern __shared__ __attribute__((aligned(16))) unsigned char cpm_lds[];
    const int b = blockIdx.x, z = blockIdx.y;
    bf16x8 wv[8];
    {   // the first chunk of the kernel matrix is on its way before the period and the image are known
        constexpr int KC4 = CPM_ITEMS / (NCT * 16 * 4) * 4;
        const int nch = KS * KS * (C >> 3), K = KS * KS * C;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = threadIdx.x + u * 256, n = i / KC4, q = i - n * KC4;
            wv[u] = *reinterpret_cast<const bf16x8*>(W16 + (size_t)min(n, Cout - 1) * K + (size_t)min(q, nch - 1) * 8);
        }
    }
    const int p = period[z], len = rows[z] / B;
    x += (size_t)z * xs;
    y += (size_t)z * ys;
    if (zpre) zpre += (size_t)z * ys;
#define CPM_BODY(RT) cpm_body<NCT, RT>(B, Lmax, C, KS, p, len, b, x, W16, bias, Cout, act, zpre, y, ks_inv, c8_shift, cpm_lds, wv)
    if (len <= 64) CPM_BODY(4);
    else if (len <= 80) CPM_BODY(5);
    else if (len <= 96) CPM_BODY(6);
    else CPM_BODY(8);
#undef CPM_BODY
}

static bool conv_period_mfma_ok(int Lmax, int C, int Cout, int KS) {
    return (C % 8) == 0 && C >= 8 && Cout >= 1 && Cout <= 16 * CPM_CT && Lmax <= 128 && KS * KS <= 1024 && (size_t)Lmax * C * 2 <= 16 * 1024;
}
static int launch_conv_period_mfma(int B, int Lmax, int k, int C, int KS, const int* period, const int* rows, const float* x, long xs,
                                   const bf16_t* W16, const float* bias, int Cout, int act, float* zpre, float* y, long ys, hipStream_t s) {
    const int taps = KS * KS;
    const int ncol = Cout <= 16 ? 16 : Cout <= 32 ? 32 : 64;        // (the kernel's column tiles: 1, 2 or 4)
    // image | 16 bytes of zeros | max(one chunk of the kernel matrix: ncol rows x (KC x 32 + 32) bf16, the four waves' partial tiles)
    const int KC = CPM_ITEMS / (ncol * 4), rt = Lmax <= 64 ? 4 : Lmax <= 80 ? 5 : Lmax <= 96 ? 6 : 8;
    const size_t chunk = (size_t)ncol * (KC * 32 + 32) * 2, red = (size_t)4 * rt * (ncol / 16) * 64 * 16;
    const size_t lds = (size_t)Lmax * C * 2 + 16 + (chunk > red ? chunk : red) + 64;
    static const hipError_t attr[3] = {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_period_mfma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_period_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_period_mfma_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)};
    for (int i = 0; i < 3; ++i)
        if (attr[i] != hipSuccess) return (int)attr[i];
    const int ks_inv = (65536 + KS - 1) / KS;
    for (int t = 0; t < taps; ++t)
        if (((t * ks_inv) >> 16) != t / KS) return IMMTSF_EUNSUPPORTED;       // (never for KS <= 31)
    const int c8 = C / 8;
    int c8_shift = -1;
    for (int sh = 0; sh < 8; ++sh)
        if ((1 << sh) == c8) c8_shift = sh;
#define CPM(NCT) hipLaunchKernelGGL(conv_period_mfma_kernel<NCT>, dim3(B, k), dim3(256), lds, s, B, Lmax, C, KS, period, rows, x, xs, W16, bias, Cout, act, \
                                    zpre, y, ys, ks_inv, c8_shift)
    if (ncol <= 16) CPM(1); else if (ncol <= 32) CPM(2); else CPM(4);
#undef CPM
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

inline bool bad_conv(int B, int H, int W, int Cin, int Cout, int KS) {
    return B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KS < 1 || !(KS & 1) || (long)B * H * W > (1L << 30);
}

}  // namespace

extern "C" {

int immtsf_inception_merge(int32_t n, int32_t Cin, int32_t Cout, const float* const* W, const float* const* b, float* W_eff, float* b_eff,
                           immtsf_stream_t stream) {
    if (n < 1 || n > IMMTSF_INCEPTION_MAX || Cin <= 0 || Cout <= 0 || !W || !b || !W_eff || !b_eff) return IMMTSF_EINVAL;
    KernelPtrs p;
    for (int j = 0; j < n; ++j) {
        if (!W[j] || !b[j]) return IMMTSF_EINVAL;
        p.w[j] = W[j];
        p.b[j] = b[j];
    }
    const long total = (long)Cout * (2 * n - 1) * (2 * n - 1) * Cin;
    hipLaunchKernelGGL(inception_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n, Cin,
                       Cout, p, W_eff, b_eff);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int immtsf_inception_unmerge(int32_t n, int32_t Cin, int32_t Cout, const float* dW_eff, const float* db_eff, float* const* dW, float* const* db,
                             immtsf_stream_t stream) {
    if (n < 1 || n > IMMTSF_INCEPTION_MAX || Cin <= 0 || Cout <= 0 || !dW_eff || !db_eff || !dW || !db) return IMMTSF_EINVAL;
    KernelGrads g;
    for (int j = 0; j < n; ++j) {
        if (!dW[j] || !db[j]) return IMMTSF_EINVAL;
        g.w[j] = dW[j];
        g.b[j] = db[j];
    }
    const long biggest = (long)Cout * Cin * (2 * n - 1) * (2 * n - 1);
    hipLaunchKernelGGL(inception_unmerge_kernel, dim3((unsigned)((biggest + 255) / 256), n), dim3(256), 0, static_cast<hipStream_t>(stream), n,
                       Cin, Cout, dW_eff, db_eff, g);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

/* x (B, H, W, Cin) channels-last; W_eff (Cout, KS*KS*Cin) tap-major / channel-minor; y (B, H, W, Cout).  col: workspace of
 * B*H*W * KS*KS*Cin floats, kept for backward; z_pre (may be NULL unless act == 2): the pre-activation, kept for backward */
int immtsf_conv2d_same_cl_forward(int32_t precision, const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t KS,
                                  const float* W_eff, const float* b_eff, int32_t Cout, int32_t act, float* col, float* z_pre, float* y,
                                  immtsf_stream_t stream) {
    if (!x || !W_eff || !col || !y || bad_conv(B, H, W, Cin, Cout, KS) || (act != 0 && act != 2) || (act == 2 && !z_pre)) return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows = B * H * W, K = KS * KS * Cin;
    hipLaunchKernelGGL(im2col_cl_kernel, dim3(rows), dim3(256), 0, s, ConvDims{B, H, W, Cin, KS}, x, col);
    IMMTSF_LAUNCH_CHECK();
    GemmArgs g = gemm_args(rows, Cout, K, K, K, Cout);
    set_problem(g, 0, col, W_eff, y, b_eff);
    g.p[0].Cpre = act == 2 ? z_pre : nullptr;
    g.act = act;
    return immtsf_launch_gemm(GEMM_NT, precision, g, s);
}

/* dy (B, H, W, Cout) -> dx (B, H, W, Cin; may be NULL), dW_eff (Cout, K), db_eff (Cout) (written).  scratch:
 * immtsf_conv2d_same_cl_scratch_floats(...) floats (the im2col image of dz, dz itself when act != 0, the flipped kernel) */
size_t immtsf_conv2d_same_cl_scratch_floats(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t KS, int32_t Cout) {
    const size_t rows = (size_t)B * H * W, K = (size_t)KS * KS * Cin, K2 = (size_t)KS * KS * Cout;
    return rows * (K > K2 ? K : K2) + rows * Cout + (size_t)Cin * K2 + 64;
}

int immtsf_conv2d_same_cl_backward(int32_t precision, const float* col, const float* z_pre, const float* y, const float* dy, int32_t B,
                                   int32_t H, int32_t W, int32_t Cin, int32_t KS, const float* W_eff, int32_t Cout, int32_t act, float* dx,
                                   float* dW_eff, float* db_eff, float* scratch, immtsf_stream_t stream) {
    (void)y;
    if (!col || !dy || !W_eff || !dW_eff || !db_eff || !scratch || bad_conv(B, H, W, Cin, Cout, KS) || (act != 0 && act != 2) ||
        (act == 2 && !z_pre))
        return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows = B * H * W, K = KS * KS * Cin;
    const int K2s = KS * KS * Cout;
    float* dz = scratch + (size_t)rows * (K > K2s ? K : K2s);
    const float* g0 = dy;
    if (act == 2) {
        const long n = (long)rows * Cout;
        hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy, z_pre, dz, n);
        IMMTSF_LAUNCH_CHECK();
        g0 = dz;
    }
    {   // dW_eff = g0^T col ; db_eff = column sums of g0
        GemmArgs h = gemm_args(Cout, K, rows, Cout, K, K);
        set_problem(h, 0, g0, col, dW_eff, nullptr, db_eff);
        CHECK(immtsf_launch_gemm(GEMM_TN, precision, h, s));
    }
    if (dx) {
        // dx = conv(g0, flipped kernel): the same im2col + GEMM as the forward (K = KS*KS*Cout) instead of materialising
        // dcol = g0 W_eff (rows x KS*KS*Cin floats: 64 MB at TimesNet's cfg4 shape, written by a K = Cout = 16 GEMM at 0.7 TB/s)
        // and gathering it back
        const int K2 = KS * KS * Cout;
        float* colz = scratch;                                   // rows x K2
        float* Wf = scratch + (size_t)rows * (K > K2 ? K : K2) + (size_t)rows * Cout;     // Cin x K2, behind dz
        const long nw = (long)Cin * K2;
        hipLaunchKernelGGL(flip_weight_kernel<float>, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, Cin, Cout, KS, W_eff, Wf);
        IMMTSF_LAUNCH_CHECK();
        hipLaunchKernelGGL(im2col_cl_kernel, dim3(rows), dim3(256), 0, s, ConvDims{B, H, W, Cout, KS}, g0, colz);
        IMMTSF_LAUNCH_CHECK();
        GemmArgs g = gemm_args(rows, Cin, K2, K2, K2, Cin);
        set_problem(g, 0, colz, Wf, dx, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NT, precision, g, s));
    }
    return IMMTSF_OK;
}

/* ---- period images with the period on the device (see the kernels above) ---- */
int immtsf_period_rows(const int64_t* top, int32_t k, int32_t total, int32_t B, int32_t* period, int32_t* rows, immtsf_stream_t stream) {
    if (!top || !period || !rows || k < 1 || k > 64 || total <= 0 || B <= 0) return IMMTSF_EINVAL;
    hipLaunchKernelGGL(period_rows_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), reinterpret_cast<const long long*>(top), k, total, B,
                       period, rows);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// precision 1: `col` holds the im2col image as bf16 (R * K * 2 bytes suffice), the products run on the bf16-in-HBM kernels with the
// device row count, GELU is its own row kernel behind the product (z_pre = the product)
static bool period_hf(int precision, int Cin, int Cout, int KS) { return precision == 1 && ((KS * KS * Cin) % 8) == 0 && ((KS * KS * Cout) % 8) == 0 && (Cin % 8) == 0 && (Cout % 8) == 0; }

int immtsf_conv2d_period_forward(int32_t precision, const float* x, int32_t B, int32_t Lmax, const int32_t* period, const int32_t* rows, int32_t Cin,
                                 int32_t KS, const float* W_eff, const float* b_eff, int32_t Cout, int32_t act, float* col, float* z_pre, float* y,
                                 void* w16, int32_t w16_ready, immtsf_stream_t stream) {
    if (!x || !period || !rows || !W_eff || !col || !y || bad_conv(B, Lmax, 1, Cin, Cout, KS) || (act != 0 && act != 2) || (act == 2 && !z_pre))
        return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int max_rows = B * Lmax, K = KS * KS * Cin;
    if (period_hf(precision, Cin, Cout, KS) && w16) {
        bf16_t* col16 = reinterpret_cast<bf16_t*>(col);
        launch_im2col_period<bf16_t>(B, Cin, KS, period, rows, x, col16, max_rows, s);
        IMMTSF_LAUNCH_CHECK();
        if (!w16_ready) CHECK(launch_f32_to_bf16(W_eff, w16, (size_t)Cout * K, s));      // (the caller casts once per block and step otherwise)
        GemmArgs g = gemm_args(max_rows, Cout, K, K, K, Cout);
        set_problem2(g, 0, mat(nullptr, col16), cmat(W_eff, w16), mat(act == 2 ? z_pre : y), b_eff);
        g.dyn = rows; g.dyn_which = 0;
        CHECK(immtsf_launch_gemm(GEMM_NT, 1, g, s));
        if (act == 2) {
            const long n = (long)max_rows * Cout;
            hipLaunchKernelGGL(gelu_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z_pre, y, Cout, rows);
            IMMTSF_LAUNCH_CHECK();
        }
        return IMMTSF_OK;
    }
    launch_im2col_period<float>(B, Cin, KS, period, rows, x, col, max_rows, s);
    IMMTSF_LAUNCH_CHECK();
    GemmArgs g = gemm_args(max_rows, Cout, K, K, K, Cout);
    set_problem(g, 0, col, W_eff, y, b_eff);
    g.p[0].Cpre = act == 2 ? z_pre : nullptr;
    g.act = act;
    g.dyn = rows; g.dyn_which = 0;
    return immtsf_launch_gemm(GEMM_NT, precision, g, s);
}

size_t immtsf_conv2d_period_scratch_floats(int32_t B, int32_t Lmax, int32_t Cin, int32_t KS, int32_t Cout) {
    return immtsf_conv2d_same_cl_scratch_floats(B, Lmax, 1, Cin, KS, Cout) + (size_t)B * Lmax * Cout;      // (+ the bf16 image of dz)
}

int immtsf_conv2d_period_backward(int32_t precision, const float* col, const float* z_pre, const float* dy, int32_t B, int32_t Lmax,
                                  const int32_t* period, const int32_t* rows, int32_t Cin, int32_t KS, const float* W_eff, int32_t Cout, int32_t act,
                                  float* dx, float* dW_eff, float* db_eff, float* scratch, void* w16, immtsf_stream_t stream) {
    if (!col || !dy || !period || !rows || !W_eff || !dW_eff || !db_eff || !scratch || bad_conv(B, Lmax, 1, Cin, Cout, KS) || (act != 0 && act != 2) ||
        (act == 2 && !z_pre))
        return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int max_rows = B * Lmax, K = KS * KS * Cin, K2 = KS * KS * Cout;
    const bool hf = period_hf(precision, Cin, Cout, KS) && w16;
    float* dz = scratch + (size_t)max_rows * (K > K2 ? K : K2);
    const float* g0 = dy;
    if (act == 2) {
        const long n = (long)max_rows * Cout;
        hipLaunchKernelGGL(gelu_bwd_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dy, z_pre, dz, Cout, rows);
        IMMTSF_LAUNCH_CHECK();
        g0 = dz;
    }
    float* Wf = scratch + (size_t)max_rows * (K > K2 ? K : K2) + (size_t)max_rows * Cout;      // Cin x K2 (fp32, or its bf16 image), behind dz
    if (hf) {
        // bf16 images: g0 (behind the flipped kernel), the forward's col, the flipped kernel, the im2col image of g0
        bf16_t* g016 = reinterpret_cast<bf16_t*>(Wf + (size_t)Cin * K2);
        CHECK(launch_f32_to_bf16(g0, g016, (size_t)max_rows * Cout, s));       // (rows beyond the valid ones: never read -- dynamic M / K)
        {   // dW_eff = g0^T col ; db_eff = column sums of g0 -- over the valid rows
            GemmArgs h = gemm_args(Cout, K, max_rows, Cout, K, K);
            set_problem2(h, 0, cmat(g0, g016), cmat(nullptr, col), mat(dW_eff), nullptr, db_eff);
            h.dyn = rows; h.dyn_which = 1;
            CHECK(immtsf_launch_gemm(GEMM_TN, 1, h, s));
        }
        if (dx) {
            bf16_t* colz16 = reinterpret_cast<bf16_t*>(scratch);
            bf16_t* Wf16 = reinterpret_cast<bf16_t*>(Wf);
            const long nw = (long)Cin * K2;
            hipLaunchKernelGGL(flip_weight_kernel<bf16_t>, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, Cin, Cout, KS, W_eff, Wf16);
            IMMTSF_LAUNCH_CHECK();
            launch_im2col_period<bf16_t>(B, Cout, KS, period, rows, g0, colz16, max_rows, s);
            IMMTSF_LAUNCH_CHECK();
            GemmArgs g = gemm_args(max_rows, Cin, K2, K2, K2, Cin);
            set_problem2(g, 0, mat(nullptr, colz16), mat(nullptr, Wf16), mat(dx), nullptr);
            g.dyn = rows; g.dyn_which = 0;
            CHECK(immtsf_launch_gemm(GEMM_NT, 1, g, s));
        }
        return IMMTSF_OK;
    }
    {   // dW_eff = g0^T col ; db_eff = column sums of g0 -- over the valid rows
        GemmArgs h = gemm_args(Cout, K, max_rows, Cout, K, K);
        set_problem(h, 0, g0, col, dW_eff, nullptr, db_eff);
        h.dyn = rows; h.dyn_which = 1;
        CHECK(immtsf_launch_gemm(GEMM_TN, precision, h, s));
    }
    if (dx) {
        float* colz = scratch;
        const long nw = (long)Cin * K2;
        hipLaunchKernelGGL(flip_weight_kernel<float>, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, Cin, Cout, KS, W_eff, Wf);
        IMMTSF_LAUNCH_CHECK();
        launch_im2col_period<float>(B, Cout, KS, period, rows, g0, colz, max_rows, s);
        IMMTSF_LAUNCH_CHECK();
        GemmArgs g = gemm_args(max_rows, Cin, K2, K2, K2, Cin);
        set_problem(g, 0, colz, Wf, dx, nullptr);
        g.dyn = rows; g.dyn_which = 0;
        CHECK(immtsf_launch_gemm(GEMM_NT, precision, g, s));
    }
    return IMMTSF_OK;
}

/* ---- k period images in ONE call (TimesNet's TimesBlock: top-k periods, the same merged kernel for each; reference models/TimesNet.py:62-79
 * loops over the periods).  x: one input shared by every image (x_stride == 0) or k inputs x + j x_stride; period / rows: k device
 * numbers each (immtsf_period_rows); y, z_pre: (k, B*Lmax, Cout); col: k im2col images of B*Lmax x KS*KS*Cin (bf16 in bf16 mode, else
 * fp32).  bf16 mode (the shapes period_hf takes): one im2col launch, one product launch with the k images along grid.z (bias + GELU in
 * its epilogue) -- 2 launches instead of 3 k.  Otherwise: the k single-image calls.  k <= 16. */
int immtsf_conv2d_periods_forward(int32_t precision, const float* x, int64_t x_stride, int32_t B, int32_t Lmax, int32_t k, const int32_t* period,
                                  const int32_t* rows, int32_t Cin, int32_t KS, const float* W_eff, const float* b_eff, int32_t Cout, int32_t act,
                                  float* col, float* z_pre, float* y, void* w16, int32_t w16_ready, immtsf_stream_t stream) {
    if (!x || !period || !rows || !W_eff || !y || k < 1 || k > 16 || x_stride < 0 || bad_conv(B, Lmax, 1, Cin, Cout, KS) ||
        (act != 0 && act != 2) || (act == 2 && !z_pre))
        return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int R = B * Lmax, K = KS * KS * Cin;
    const size_t per_y = (size_t)R * Cout, per_col = (size_t)R * K;
    if (!col) {     // no im2col image: the workgroup of a (window, period) forms the operand rows from the image it holds in LDS
        if (!(period_hf(precision, Cin, Cout, KS) && w16 && conv_period_mfma_ok(Lmax, Cin, Cout, KS))) return IMMTSF_EINVAL;
        if (!w16_ready) CHECK(launch_f32_to_bf16(W_eff, w16, (size_t)Cout * K, s));
        return launch_conv_period_mfma(B, Lmax, k, Cin, KS, period, rows, x, (long)x_stride, static_cast<const bf16_t*>(w16), b_eff, Cout, act,
                                       act == 2 ? z_pre : nullptr, y, (long)per_y, s);
    }
    if (!(period_hf(precision, Cin, Cout, KS) && w16)) {
        for (int j = 0; j < k; ++j)
            CHECK(immtsf_conv2d_period_forward(precision, x + (size_t)j * x_stride, B, Lmax, period + j, rows + j, Cin, KS, W_eff, b_eff, Cout, act,
                                               col + (size_t)j * per_col, act == 2 ? z_pre + (size_t)j * per_y : nullptr, y + (size_t)j * per_y, nullptr, 0,
                                               stream));
        return IMMTSF_OK;
    }
    bf16_t* col16 = reinterpret_cast<bf16_t*>(col);
    launch_im2col_period<bf16_t>(B, Cin, KS, period, rows, x, col16, R, s, k, (long)x_stride, (long)per_col);
    IMMTSF_LAUNCH_CHECK();
    if (!w16_ready) CHECK(launch_f32_to_bf16(W_eff, w16, (size_t)Cout * K, s));
    GemmArgs g = gemm_args(R, Cout, K, K, K, Cout);
    set_problem2(g, 0, mat(nullptr, col16), cmat(W_eff, w16), mat(y), b_eff);
    g.p[0].Cpre = act == 2 ? z_pre : nullptr;
    g.act = act;
    g.dyn = rows; g.dyn_which = 0; g.dyn_stride = 1;
    g.zbatch = k; g.zsA = (long)per_col; g.zsB = 0; g.zsC = (long)per_y;
    return immtsf_launch_gemm2(GEMM_NT, g, s);
}

size_t immtsf_conv2d_periods_scratch_floats(int32_t B, int32_t Lmax, int32_t k, int32_t Cin, int32_t KS, int32_t Cout) {
    // k x [im2col image of g0 | dz | bf16 image of g0], the flipped kernel, k x (B Lmax, Cin) data gradients of a shared input, one
    // kernel-gradient slot (the single-image fallback sums through it)
    const size_t R = (size_t)B * Lmax, K = (size_t)KS * KS * Cin, K2 = (size_t)KS * KS * Cout;
    return (size_t)k * (R * (K > K2 ? K : K2) + 2 * R * Cout + R * Cin) + (size_t)Cin * K2 + 64 + (size_t)Cout * K + Cout + 256;
}

/* dy (k, B*Lmax, Cout) -> dx ((B*Lmax, Cin) summed over the images when dx_shared, else (k, B*Lmax, Cin); may be NULL), dW_eff / db_eff: the
 * SUM over the k images, ACCUMULATED by atomics -- the caller hands them in zeroed (one fill for both when they are one buffer).  Rows of
 * dx beyond an image's rows are not written (a shared dx: zero where no image holds the row). */
int immtsf_conv2d_periods_backward(int32_t precision, const float* col, const float* z_pre, const float* dy, int32_t B, int32_t Lmax, int32_t k,
                                   const int32_t* period, const int32_t* rows, int32_t Cin, int32_t KS, const float* W_eff, int32_t Cout,
                                   int32_t act, float* dx, int32_t dx_shared, float* dW_eff, float* db_eff, float* scratch, void* w16,
                                   immtsf_stream_t stream) {
    if (!col || !dy || !period || !rows || !W_eff || !dW_eff || !db_eff || !scratch || k < 1 || k > 16 || bad_conv(B, Lmax, 1, Cin, Cout, KS) ||
        (act != 0 && act != 2) || (act == 2 && !z_pre))
        return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int R = B * Lmax, K = KS * KS * Cin, K2 = KS * KS * Cout;
    const size_t per_y = (size_t)R * Cout, per_col = (size_t)R * K, per_colz = (size_t)R * K2, per_x = (size_t)R * Cin;
    const size_t colz_floats = (size_t)k * R * (K > K2 ? K : K2);
    float* dz = scratch + colz_floats;                       // k x R x Cout
    float* g16f = dz + (size_t)k * per_y;                    // k x R x Cout bf16 (in float slots)
    float* Wf = g16f + (size_t)k * per_y;                    // Cin x K2 (+ 64)
    float* dxk = Wf + (size_t)Cin * K2 + 64;                 // k x R x Cin
    float* tW = dxk + (size_t)k * per_x;                     // Cout x K, then Cout
    const bool hf = period_hf(precision, Cin, Cout, KS) && w16;
    if (!hf) {
        // the k single-image calls (their scratch: the head of this one, up to and including the flipped kernel's slot); their kernel
        // gradients are summed through a slot of the scratch
        float* tb = tW + (size_t)Cout * K;
        for (int j = 0; j < k; ++j) {
            float* dxj = dx ? (dx_shared ? dxk + (size_t)j * per_x : dx + (size_t)j * per_x) : nullptr;
            CHECK(immtsf_conv2d_period_backward(precision, col + (size_t)j * per_col, act == 2 ? z_pre + (size_t)j * per_y : nullptr,
                                                dy + (size_t)j * per_y, B, Lmax, period + j, rows + j, Cin, KS, W_eff, Cout, act, dxj, tW, tb, scratch,
                                                nullptr, stream));
            CHECK(launch_axpy(tW, 1.f, dW_eff, Cout * K, 1, s));
            CHECK(launch_axpy(tb, 1.f, db_eff, Cout, 1, s));
        }
    } else {
        bf16_t* g16 = reinterpret_cast<bf16_t*>(g16f);
        const long n4 = ((long)per_y + 3) / 4;
        hipLaunchKernelGGL(period_g0_kernel, dim3((unsigned)((n4 + 255) / 256), k), dim3(256), 0, s, dy, act == 2 ? z_pre : nullptr, dz, g16, Cout,
                           (long)per_y, rows);
        IMMTSF_LAUNCH_CHECK();
        const float* g0 = act == 2 ? dz : dy;
        {   // dW_eff += sum_z g0_z^T col_z ; db_eff += column sums -- the k images along grid.z, fp32 atomics into the zeroed buffers
            GemmArgs h = gemm_args(Cout, K, R, Cout, K, K);
            set_problem2(h, 0, cmat(nullptr, g16), cmat(nullptr, col), mat(dW_eff), nullptr, db_eff);
            h.dyn = rows; h.dyn_which = 1; h.dyn_stride = 1;
            h.zbatch = k; h.zsA = (long)per_y; h.zsB = (long)per_col; h.zsC = 0; h.atomic_c = 1; h.c_prezeroed = 1;
            CHECK(immtsf_launch_gemm2(GEMM_TN, h, s));
        }
        if (dx) {
            bf16_t* colz16 = reinterpret_cast<bf16_t*>(scratch);
            bf16_t* Wf16 = reinterpret_cast<bf16_t*>(Wf);
            const long nw = (long)Cin * K2;
            hipLaunchKernelGGL(flip_weight_kernel<bf16_t>, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, Cin, Cout, KS, W_eff, Wf16);
            IMMTSF_LAUNCH_CHECK();
            launch_im2col_period<bf16_t>(B, Cout, KS, period, rows, g0, colz16, R, s, k, (long)per_y, (long)per_colz);
            IMMTSF_LAUNCH_CHECK();
            GemmArgs g = gemm_args(R, Cin, K2, K2, K2, Cin);
            set_problem2(g, 0, mat(nullptr, colz16), mat(nullptr, Wf16), mat(dx_shared ? dxk : dx), nullptr);
            g.dyn = rows; g.dyn_which = 0; g.dyn_stride = 1;
            g.zbatch = k; g.zsA = (long)per_colz; g.zsB = 0; g.zsC = (long)per_x;
            CHECK(immtsf_launch_gemm2(GEMM_NT, g, s));
        }
    }
    if (dx && dx_shared) {
        const long n4 = ((long)per_x + 3) / 4;
        hipLaunchKernelGGL(period_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, dxk, k, (long)per_x, rows, Cin, dx);
        IMMTSF_LAUNCH_CHECK();
    }
    return IMMTSF_OK;
}

int immtsf_period_aggregate_forward(const float* Y, const float* w, const float* x, int32_t B, int32_t total, int32_t Lmax, int32_t N, int32_t k,
                                    float* out, immtsf_stream_t stream) {
    if (!Y || !w || !x || !out || B <= 0 || total <= 0 || Lmax < total || N <= 0 || k < 1 || k > 16) return IMMTSF_EINVAL;
    const long n = (long)B * total * N;
    hipLaunchKernelGGL(period_agg_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), Y, w, x, B, total,
                       (long)Lmax * B * N, N, k, out);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int immtsf_period_aggregate_backward(const float* Y, const float* w, const float* dout, int32_t B, int32_t total, int32_t Lmax, int32_t N, int32_t k,
                                     float* dY, float* dw, immtsf_stream_t stream) {
    if (!Y || !w || !dout || !dY || !dw || B <= 0 || total <= 0 || Lmax < total || N <= 0 || k < 1 || k > 16) return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long n = (long)Lmax * B * N * k;
    hipLaunchKernelGGL(period_agg_bwd_y_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, dout, B, total, Lmax, N, k, dY);
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL(period_agg_bwd_w_kernel, dim3(B), dim3(256), 0, s, Y, dout, B, total, (long)Lmax * B * N, N, k, dw);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

/* 1 when immtsf_conv2d_periods_forward takes col == NULL for these dimensions (the implicit form: bf16 mode, channel counts multiples of 8 up
 * to 64, Lmax <= 128) and immtsf_conv2d_periods_backward_x exists for them, 0 otherwise */
int immtsf_conv2d_periods_implicit_ok(int32_t precision, int32_t Lmax, int32_t Cin, int32_t KS, int32_t Cout) {
    return (period_hf(precision, Cin, Cout, KS) && conv_period_mfma_ok(Lmax, Cin, Cout, KS) && conv_period_mfma_ok(Lmax, Cout, Cin, KS)) ? 1 : 0;
}

size_t immtsf_conv2d_periods_backward_x_scratch_floats(int32_t B, int32_t Lmax, int32_t k, int32_t Cin, int32_t KS, int32_t Cout) {
    // [im2col image of x, bf16 | dz | bf16 image of g0 | flipped kernel, bf16 | per-image data gradients of a shared input]
    const size_t R = (size_t)B * Lmax, K = (size_t)KS * KS * Cin, K2 = (size_t)KS * KS * Cout;
    return (size_t)k * (R * K / 2 + 2 * R * Cout + R * Cin) + (size_t)Cin * K2 + 256;
}

/* the backward of the implicit form, from the INPUT x instead of a saved im2col image.  phase bit 0 = the data path (g0 = dy gelu'(z_pre),
 * its bf16 image, dx through the same LDS-image kernel with the flipped kernel), bit 1 = the kernel's gradient (an im2col image of x in
 * scratch -- nothing but this product reads it -- and the summed weight-gradient product; dW_eff / db_eff handed in ZEROED) from the images
 * the bit-0 call left in `scratch`, on any stream ordered behind it; 3 = both. */
int immtsf_conv2d_periods_backward_x(int32_t precision, const float* x, int64_t x_stride, const float* z_pre, const float* dy, int32_t B, int32_t Lmax,
                                     int32_t k, const int32_t* period, const int32_t* rows, int32_t Cin, int32_t KS, const float* W_eff,
                                     int32_t Cout, int32_t act, float* dx, int32_t dx_shared, float* dW_eff, float* db_eff, float* scratch,
                                     int32_t phase, immtsf_stream_t stream) {
    if (!x || !dy || !period || !rows || !W_eff || !scratch || k < 1 || k > 16 || x_stride < 0 || bad_conv(B, Lmax, 1, Cin, Cout, KS) ||
        (act != 0 && act != 2) || (act == 2 && !z_pre) || phase < 1 || phase > 3 || ((phase & 2) && (!dW_eff || !db_eff)))
        return IMMTSF_EINVAL;
    if (!immtsf_conv2d_periods_implicit_ok(precision, Lmax, Cin, KS, Cout)) return IMMTSF_EUNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int R = B * Lmax, K = KS * KS * Cin, K2 = KS * KS * Cout;
    const size_t per_y = (size_t)R * Cout, per_col = (size_t)R * K, per_x = (size_t)R * Cin;
    bf16_t* col16 = reinterpret_cast<bf16_t*>(scratch);                      // k x R x K bf16
    float* dz = scratch + ((size_t)k * per_col + 1) / 2;                     // k x R x Cout
    dz = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(dz) + 15) & ~(uintptr_t)15);
    float* g16f = dz + (size_t)k * per_y;                                    // k x R x Cout bf16 (in float slots)
    float* Wf = g16f + (size_t)k * per_y;                                    // Cin x K2 bf16 (in float slots)
    float* dxk = Wf + (size_t)Cin * K2;                                      // k x R x Cin
    bf16_t* g16 = reinterpret_cast<bf16_t*>(g16f);
    const float* g0 = act == 2 ? dz : dy;
    if (phase & 1) {
        const long n4 = ((long)per_y + 3) / 4;
        hipLaunchKernelGGL(period_g0_kernel, dim3((unsigned)((n4 + 255) / 256), k), dim3(256), 0, s, dy, act == 2 ? z_pre : nullptr, dz, g16, Cout,
                           (long)per_y, rows);
        IMMTSF_LAUNCH_CHECK();
        if (dx) {
            bf16_t* Wf16 = reinterpret_cast<bf16_t*>(Wf);
            const long nw = (long)Cin * K2;
            hipLaunchKernelGGL(flip_weight_kernel<bf16_t>, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, Cin, Cout, KS, W_eff, Wf16);
            IMMTSF_LAUNCH_CHECK();
            CHECK(launch_conv_period_mfma(B, Lmax, k, Cout, KS, period, rows, g0, (long)per_y, Wf16, nullptr, Cin, 0, nullptr, dx_shared ? dxk : dx,
                                          (long)per_x, s));
            if (dx_shared) {
                const long m4 = ((long)per_x + 3) / 4;
                hipLaunchKernelGGL(period_sum_kernel, dim3((unsigned)((m4 + 255) / 256)), dim3(256), 0, s, dxk, k, (long)per_x, rows, Cin, dx);
                IMMTSF_LAUNCH_CHECK();
            }
        }
    }
    if (phase & 2) {
        launch_im2col_period<bf16_t>(B, Cin, KS, period, rows, x, col16, R, s, k, (long)x_stride, (long)per_col);
        IMMTSF_LAUNCH_CHECK();
        GemmArgs h = gemm_args(Cout, K, R, Cout, K, K);
        set_problem2(h, 0, cmat(nullptr, g16), cmat(nullptr, col16), mat(dW_eff), nullptr, db_eff);
        h.dyn = rows; h.dyn_which = 1; h.dyn_stride = 1;
        h.zbatch = k; h.zsA = (long)per_y; h.zsB = (long)per_col; h.zsC = 0; h.atomic_c = 1; h.c_prezeroed = 1;
        CHECK(immtsf_launch_gemm2(GEMM_TN, h, s));
    }
    return IMMTSF_OK;
}

}  // extern "C"
