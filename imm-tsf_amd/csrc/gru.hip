// MMF_GR_Add (fusions/MMF_GR_Add.py:31-61).  The input-side GRU product W_ih [Y;E] for all (b,t) is hoisted into
// one MFMA GEMM by the caller; what is sequential -- the Hd-wide hidden-state recurrence -- runs here with one
// workgroup per window, the hidden state in LDS and W_hh read through L1 in a transposed (coalesced) order.
#include "gru.hpp"

namespace {

__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, int wa, const float* __restrict__ b, int wb,
                                                       int rows, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* o = out + (size_t)row * (wa + wb);
    for (int i = lane; i < wa; i += 64) o[i] = a[(size_t)row * wa + i];
    for (int i = lane; i < wb; i += 64) o[wa + i] = b[(size_t)row * wb + i];
}

__global__ __launch_bounds__(256) void split2_kernel(const float* __restrict__ x, int wa, int wb, int rows, float* __restrict__ da,
                                                      int acc_a, float* __restrict__ db) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + (size_t)row * (wa + wb);
    if (da) for (int i = lane; i < wa; i += 64) {
        float* q = da + (size_t)row * wa + i;
        *q = acc_a ? *q + p[i] : p[i];
    }
    if (db) for (int i = lane; i < wb; i += 64) db[(size_t)row * wb + i] = p[wa + i];
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// one block per window, blockDim = Hd rounded up to 64.  LDS: hs[Hd]
__global__ void gru_fwd_kernel(int T, int Hd, const float* __restrict__ gi, const float* __restrict__ w_hh,
                               const float* __restrict__ b_hh, float* __restrict__ r_o, float* __restrict__ z_o,
                               float* __restrict__ n_o, float* __restrict__ hn_o, float* __restrict__ h_o,
                               float* __restrict__ hp_o) {
    extern __shared__ float hs[];
    const int b = blockIdx.x, j = threadIdx.x;
    const bool on = j < Hd;
    if (on) hs[j] = 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)b * T + t;
        float hr = 0.f, hz = 0.f, hn = 0.f, hprev = 0.f;
        if (on) {
            hr = b_hh[j]; hz = b_hh[Hd + j]; hn = b_hh[2 * Hd + j];
            const float* wr = w_hh + (size_t)j * Hd;
            const float* wz = w_hh + (size_t)(Hd + j) * Hd;
            const float* wn = w_hh + (size_t)(2 * Hd + j) * Hd;
            for (int k = 0; k < Hd; ++k) {
                const float hk = hs[k];
                hr = fmaf(wr[k], hk, hr);
                hz = fmaf(wz[k], hk, hz);
                hn = fmaf(wn[k], hk, hn);
            }
            hprev = hs[j];
        }
        __syncthreads();
        if (on) {
            const float* g = gi + row * 3 * Hd;
            const float r = sigmoidf_(g[j] + hr);
            const float z = sigmoidf_(g[Hd + j] + hz);
            const float n = tanhf(g[2 * Hd + j] + r * hn);
            const float h = (1.f - z) * n + z * hprev;
            r_o[row * Hd + j] = r; z_o[row * Hd + j] = z; n_o[row * Hd + j] = n; hn_o[row * Hd + j] = hn;
            h_o[row * Hd + j] = h; hp_o[row * Hd + j] = hprev;
            hs[j] = h;
        }
        __syncthreads();
    }
}

// reverse-time recurrence.  LDS: dh[Hd] | gh[3*Hd]
__global__ void gru_bwd_kernel(int T, int Hd, const float* __restrict__ dh_in, const float* __restrict__ w_hh,
                               const float* __restrict__ r_i, const float* __restrict__ z_i, const float* __restrict__ n_i,
                               const float* __restrict__ hn_i, const float* __restrict__ hp_i, float* __restrict__ dgi,
                               float* __restrict__ dgh) {
    extern __shared__ float sm[];
    float* gh = sm;           // [3*Hd] dgh of the current step
    const int b = blockIdx.x, j = threadIdx.x;
    const bool on = j < Hd;
    float dh_carry = 0.f;     // gradient flowing into h_t from step t+1 (component j)
    for (int t = T - 1; t >= 0; --t) {
        const size_t row = (size_t)b * T + t;
        if (on) {
            const float dh = dh_in[row * Hd + j] + dh_carry;
            const float r = r_i[row * Hd + j], z = z_i[row * Hd + j], n = n_i[row * Hd + j], hn = hn_i[row * Hd + j],
                        hp = hp_i[row * Hd + j];
            const float dn = dh * (1.f - z), dz = dh * (hp - n);
            const float dan = dn * (1.f - n * n);
            const float daz = dz * z * (1.f - z);
            const float dar = dan * hn * r * (1.f - r);
            dgi[row * 3 * Hd + j] = dar; dgi[row * 3 * Hd + Hd + j] = daz; dgi[row * 3 * Hd + 2 * Hd + j] = dan;
            const float dhn = dan * r;
            dgh[row * 3 * Hd + j] = dar; dgh[row * 3 * Hd + Hd + j] = daz; dgh[row * 3 * Hd + 2 * Hd + j] = dhn;
            gh[j] = dar; gh[Hd + j] = daz; gh[2 * Hd + j] = dhn;
            dh_carry = dh * z;
        }
        __syncthreads();
        if (on) {   // dh_prev[j] += sum_g W_hh[g, j] * dgh[g]   (column j: stride Hd, coalesced over j)
            float a = 0.f;
            for (int g = 0; g < 3 * Hd; ++g) a = fmaf(w_hh[(size_t)g * Hd + j], gh[g], a);
            dh_carry += a;
        }
        __syncthreads();
    }
}

// thread per (b,t) row.  reference: fusions/MMF_GR_Add.py:47-60
__global__ __launch_bounds__(256) void gr_tail_fwd_kernel(int BT, int T, int C, int Hd, const float* __restrict__ h,
                                                           const float* __restrict__ res_w, const float* __restrict__ res_b,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ gl, const float* __restrict__ Y,
                                                           const unsigned char* __restrict__ mtxt, float* __restrict__ xhat,
                                                           float* __restrict__ rstd, float* __restrict__ g_out,
                                                           float* __restrict__ dd_out, float* __restrict__ Yout, DropCfg drop,
                                                           uint64_t site) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= BT) return;
    const bool live = mtxt[row / T] != 0;
    const float* hr = h + (size_t)row * Hd;
    float* xh = xhat + (size_t)row * C;   // used as scratch for delta first
    float mu = 0.f;
    for (int c = 0; c < C; ++c) {
        float a = res_b[c];
        for (int k = 0; k < Hd; ++k) a = fmaf(res_w[(size_t)c * Hd + k], hr[k], a);
        xh[c] = a;
        mu += a;
    }
    mu /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; ++c) { const float t = xh[c] - mu; var = fmaf(t, t, var); }
    const float rs = 1.0f / sqrtf(var / (float)C + 1e-5f);
    rstd[row] = rs;
    for (int c = 0; c < C; ++c) {
        const size_t i = (size_t)row * C + c;
        const float hh = (xh[c] - mu) * rs;
        xh[c] = hh;
        const float dd = fmaf(hh, gamma[c], beta[c]) * dropout_scale(drop, site, i);
        const float g = live ? sigmoidf_(gl[i]) : 1.f;
        g_out[i] = g;
        dd_out[i] = dd;
        const float y = Y[i];
        Yout[i] = g * y + (1.f - g) * (y + dd);
    }
}

__global__ __launch_bounds__(256) void gr_tail_bwd_kernel(int BT, int T, int C, int Hd, const float* __restrict__ dYout,
                                                           const float* __restrict__ res_w, const float* __restrict__ gamma,
                                                           const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                           const float* __restrict__ g_i, const float* __restrict__ dd_i,
                                                           const unsigned char* __restrict__ mtxt, float* __restrict__ dn,
                                                           float* __restrict__ ddelta, float* __restrict__ dgl,
                                                           float* __restrict__ dh_in, DropCfg drop, uint64_t site) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= BT) return;
    const bool live = mtxt[row / T] != 0;
    float c1 = 0.f, c2 = 0.f;
    for (int c = 0; c < C; ++c) {
        const size_t i = (size_t)row * C + c;
        const float go = dYout[i], g = g_i[i];
        // out = y + (1-g)*dd
        dgl[i] = live ? (-dd_i[i] * go) * g * (1.f - g) : 0.f;
        const float gn = (1.f - g) * go * dropout_scale(drop, site, i);
        dn[i] = gn;
        const float t = gn * gamma[c];
        c1 += t;
        c2 = fmaf(t, xhat[i], c2);
    }
    c1 /= (float)C;
    c2 /= (float)C;
    const float rs = rstd[row];
    for (int c = 0; c < C; ++c) {
        const size_t i = (size_t)row * C + c;
        ddelta[i] = rs * (dn[i] * gamma[c] - c1 - xhat[i] * c2);
    }
    for (int k = 0; k < Hd; ++k) {
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(res_w[(size_t)c * Hd + k], ddelta[(size_t)row * C + c], a);
        dh_in[(size_t)row * Hd + k] = a;
    }
}

}  // namespace

int launch_concat2(const float* a, int wa, const float* b, int wb, int rows, float* out, hipStream_t s) {
    if (rows <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(concat2_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, a, wa, b, wb, rows, out);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_split2(const float* x, int wa, int wb, int rows, float* da, int accumulate_a, float* db, hipStream_t s) {
    if (rows <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(split2_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, wa, wb, rows, da, accumulate_a, db);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_gru_fwd(int B, int T, int Hd, const float* gi, const float* w_hh, const float* b_hh, float* r, float* z,
                   float* n, float* hn, float* h, float* hprev, hipStream_t s) {
    if (B <= 0) return IMMTSF_OK;
    if (Hd > 1024) return IMMTSF_EUNSUPPORTED;
    const int threads = cdiv(Hd, 64) * 64;
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(B), dim3(threads), Hd * sizeof(float), s, T, Hd, gi, w_hh, b_hh, r, z, n, hn, h, hprev);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_gru_bwd(int B, int T, int Hd, const float* dh_in, const float* w_hh, const float* r, const float* z,
                   const float* n, const float* hn, const float* hprev, float* dgi, float* dgh, hipStream_t s) {
    if (B <= 0) return IMMTSF_OK;
    if (Hd > 1024) return IMMTSF_EUNSUPPORTED;
    const int threads = cdiv(Hd, 64) * 64;
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(B), dim3(threads), 3 * Hd * sizeof(float), s, T, Hd, dh_in, w_hh, r, z, n, hn, hprev,
                       dgi, dgh);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_gr_tail_fwd(int BT, int T, int C, int Hd, const float* h, const float* res_w, const float* res_b,
                       const float* gamma, const float* beta, const float* gl, const float* Y, const unsigned char* mtxt,
                       float* xhat, float* rstd, float* g_out, float* dd_out, float* Yout, DropCfg drop, uint64_t site,
                       hipStream_t s) {
    if (BT <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(gr_tail_fwd_kernel, dim3(cdiv(BT, 256)), dim3(256), 0, s, BT, T, C, Hd, h, res_w, res_b, gamma, beta, gl, Y,
                       mtxt, xhat, rstd, g_out, dd_out, Yout, drop, site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_gr_tail_bwd(int BT, int T, int C, int Hd, const float* dYout, const float* res_w, const float* gamma,
                       const float* xhat, const float* rstd, const float* g, const float* dd, const unsigned char* mtxt,
                       float* dn, float* ddelta, float* dgl, float* dh_in, DropCfg drop, uint64_t site, hipStream_t s) {
    if (BT <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(gr_tail_bwd_kernel, dim3(cdiv(BT, 256)), dim3(256), 0, s, BT, T, C, Hd, dYout, res_w, gamma, xhat, rstd, g, dd,
                       mtxt, dn, ddelta, dgl, dh_in, drop, site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
