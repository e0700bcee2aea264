// MMF_GR_Add in SPLIT form (reference: fusions/MMF_GR_Add.py:31-61).
//
// The block's two maps of x = [Y_ts ; E_txt] -- the GRU's input side W_ih x + b_ih and the gate net W_g x + b_g -- are linear, so the
// text columns can be applied where the text is, ahead of the backbone:
//
//   P[b t, :] = E_txt[b t, :] [W_ih[:, C:] ; W_g[:, C:]]^T + [b_ih ; b_g]          (B T) x (3 Hd + C): "the P half", text only
//   gi = P[:, :3 Hd] + Y_ts W_ih[:, :C]^T        gl = P[:, 3 Hd:] + Y_ts W_g[:, :C]^T          C-column products: a few FMAs per row
//
// What is left between the backbone's forward and its backward -- the Hd-wide recurrence over the T steps of a window, the residual
// head, LayerNorm(C), dropout, the sigmoid gate and the blend, and for a training step the masked MSE and the backward of all of it
// (through time) -- is a few thousand FMAs per window: ONE launch, one wave per window, every intermediate in LDS
// (gr_train_kernel).  The as-written form (fusion_blocks_rec_gr.hip: concat, two 774-column GEMMs, recurrence, tail; then the loss;
// then ten launches back) put ~290 us of dependent launches between cfg3's backbone forward and backward.
#include "../../include/immtsf.h"
#include "gemm.hpp"
#include "rowops.hpp"
#include "block_util.hpp"
#include <math.h>

int launch_rank_expand(const float* A, int lda, const float* Bm, int ldb, float* Cm, void* Ch, int M, int N, int K, hipStream_t s);
bool rank_expand_ok(int M, int N, int K, const void* Bm, int ldb, const void* Cm, const void* Ch);

namespace {

struct GRTDims { int B, T, C, Hd, PW, ld; };       // PW: pitch of P (3 Hd + C rounded up to 8); ld: pitch of W_ih / W_g (C + d)
struct GRTP { const float *w_ih, *w_hh, *b_hh, *res_w, *res_b, *gate_w, *ln_w, *ln_b; };
struct GRTG { float *w_ih, *w_hh, *b_hh, *res_w, *res_b, *gate_w, *ln_w, *ln_b; };

// the hardware's exp2 / reciprocal (1 ulp each): the recurrence is a chain of dependent instructions on ONE wave per CU, where expf's
// range reduction and an IEEE division cost several hundred cycles a step (measured: 0.6 us per step with them, 19 of the kernel's 63 us)
__device__ __forceinline__ float sigm(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_e(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }

// LDS layout (floats), see gr_lds_floats
struct GRL {
    float *Y, *P, *Wy, *Gy, *Whh, *bhh, *Rw, *rb, *gam, *bet, *r, *z, *n, *hn, *hp, *h, *xhat, *g, *dd, *dout, *dhin, *dgh, *hs, *gh, *tr, *mk, *cn;
};
__host__ __device__ inline size_t gr_lds_floats(int T, int C, int Hd, int PW) {
    return (size_t)T * C + (size_t)T * PW + 3 * Hd * C + C * C + 3 * Hd * Hd + 3 * Hd + C * Hd + 3 * C + 6 * (size_t)T * Hd + 4 * (size_t)T * C +
           (size_t)T * Hd + 3 * (size_t)T * Hd + Hd + 3 * Hd + 2 * (size_t)T * C + C + 64;
}
__device__ inline GRL gr_carve(float* p, int T, int C, int Hd, int PW) {
    GRL l;
    l.Y = p; p += T * C;
    l.P = p; p += T * PW;
    l.Wy = p; p += 3 * Hd * C;
    l.Gy = p; p += C * C;
    l.Whh = p; p += 3 * Hd * Hd;
    l.bhh = p; p += 3 * Hd;
    l.Rw = p; p += C * Hd;
    l.rb = p; p += C;
    l.gam = p; p += C;
    l.bet = p; p += C;
    l.r = p; p += T * Hd;
    l.z = p; p += T * Hd;
    l.n = p; p += T * Hd;
    l.hn = p; p += T * Hd;
    l.hp = p; p += T * Hd;
    l.h = p; p += T * Hd;
    l.xhat = p; p += T * C;
    l.g = p; p += T * C;
    l.dd = p; p += T * C;
    l.dout = p; p += T * C;
    l.dhin = p; p += T * Hd;
    l.dgh = p; p += 3 * T * Hd;
    l.hs = p; p += Hd;
    l.gh = p; p += 3 * Hd;
    l.tr = p; p += T * C;
    l.mk = p; p += T * C;
    l.cn = p; p += C;
    return l;
}

// grid B, 64 threads.  TRAIN: loss + backward; else only Y_out.
template <bool TRAIN>
__global__ __launch_bounds__(64) void gr_train_kernel(GRTDims dm, GRTP p, const float* __restrict__ Y, const float* __restrict__ Pin,
                                                       const unsigned char* __restrict__ mtxt, const float* __restrict__ truth,
                                                       const float* __restrict__ mask, const float* __restrict__ cnt, float grad_scale,
                                                       float* __restrict__ Yout, float* __restrict__ loss, float* __restrict__ partial,
                                                       unsigned int* __restrict__ ticket, float* __restrict__ dY, float* __restrict__ dP,
                                                       GRTG gr, DropCfg drop, uint64_t site, int* __restrict__ done_flag) {
    extern __shared__ __attribute__((aligned(16))) float gr_lds[];
    const int T = dm.T, C = dm.C, Hd = dm.Hd, PW = dm.PW, G3 = 3 * Hd;
    const GRL l = gr_carve(gr_lds, T, C, Hd, PW);
    const int b = blockIdx.x, lane = threadIdx.x;
    const size_t row0 = (size_t)b * T;
    const bool live = mtxt[b] != 0;
    // ---- A: the window's rows and the block's small weights
    for (int i = lane; i < T * C; i += 64) l.Y[i] = Y[row0 * C + i];
    for (int i = lane; i < T * PW; i += 64) l.P[i] = Pin[row0 * PW + i];
    for (int i = lane; i < G3 * C; i += 64) l.Wy[i] = p.w_ih[(size_t)(i / C) * dm.ld + i % C];
    for (int i = lane; i < C * C; i += 64) l.Gy[i] = p.gate_w[(size_t)(i / C) * dm.ld + i % C];
    for (int i = lane; i < G3 * Hd; i += 64) l.Whh[i] = p.w_hh[i];
    for (int i = lane; i < G3; i += 64) l.bhh[i] = p.b_hh[i];
    for (int i = lane; i < C * Hd; i += 64) l.Rw[i] = p.res_w[i];
    for (int i = lane; i < C; i += 64) { l.rb[i] = p.res_b[i]; l.gam[i] = p.ln_w[i]; l.bet[i] = p.ln_b[i]; }
    if (lane < Hd) l.hs[lane] = 0.f;
    if (TRAIN) {        // (the loss's operands too: a global load inside the per-row loop below is a full round trip with one wave per CU)
        for (int i = lane; i < T * C; i += 64) { l.tr[i] = truth[row0 * C + i]; l.mk[i] = mask[row0 * C + i]; }
        for (int i = lane; i < C; i += 64) l.cn[i] = cnt[i];
    }
    __syncthreads();
    // ---- B: the Y_ts columns of the two maps on top of the text half
    for (int i = lane; i < T * (G3 + C); i += 64) {
        const int t = i / (G3 + C), j = i - t * (G3 + C);
        const float* w = j < G3 ? l.Wy + j * C : l.Gy + (j - G3) * C;
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(w[c], l.Y[t * C + c], a);
        l.P[t * PW + j] += a;
    }
    __syncthreads();
    // ---- C: the recurrence (nn.GRU, gate order r, z, n), hidden unit j on lane j.  One wave: the hidden state lives in the lanes' registers
    // (lane k's value reaches the others by v_readlane), this lane's three rows of W_hh too -- no LDS round trip and no barrier on the
    // step-to-step dependence (through LDS + two barriers a step took ~1.2 us: 38 us of an 82 us kernel, as much again backwards)
    constexpr int HM = 16;
    {
        float wr[HM], wz[HM], wn[HM];
#pragma unroll
        for (int k = 0; k < HM; ++k) {
            const bool on = lane < Hd && k < Hd;
            wr[k] = on ? l.Whh[lane * Hd + k] : 0.f;
            wz[k] = on ? l.Whh[(Hd + lane) * Hd + k] : 0.f;
            wn[k] = on ? l.Whh[(2 * Hd + lane) * Hd + k] : 0.f;
        }
        const float br = lane < Hd ? l.bhh[lane] : 0.f, bz = lane < Hd ? l.bhh[Hd + lane] : 0.f, bn = lane < Hd ? l.bhh[2 * Hd + lane] : 0.f;
        float hcur = 0.f;
        const int jl = lane < Hd ? lane : 0;
        float g_r = l.P[jl], g_z = l.P[Hd + jl], g_n = l.P[2 * Hd + jl];       // (the step's input-side terms, one step ahead of their use)
        for (int t = 0; t < T; ++t) {
            const float gr_ = g_r, gz_ = g_z, gn_ = g_n;
            if (t + 1 < T) { const float* gi = l.P + (t + 1) * PW; g_r = gi[jl]; g_z = gi[Hd + jl]; g_n = gi[2 * Hd + jl]; }
            float hr = br, hz = bz, hn = bn;
#pragma unroll
            for (int k = 0; k < HM; ++k) {
                if (k < Hd) {                    // (wave-uniform)
                    const float hk = lane_bcast(hcur, k);
                    hr = fmaf(wr[k], hk, hr); hz = fmaf(wz[k], hk, hz); hn = fmaf(wn[k], hk, hn);
                }
            }
            if (lane < Hd) {
                const float r = sigm(gr_ + hr), z = sigm(gz_ + hz), n = tanh_e(gn_ + r * hn);
                const float h = (1.f - z) * n + z * hcur;
                const int o = t * Hd + lane;
                l.r[o] = r; l.z[o] = z; l.n[o] = n; l.hn[o] = hn; l.hp[o] = hcur; l.h[o] = h;
                hcur = h;
            }
        }
    }
    __syncthreads();
    // ---- D: per row: residual head, LayerNorm(C), dropout, gate, blend; TRAIN: the loss term and the tail's backward in the same lane
    float e = 0.f, navail = 0.f;
    if (TRAIN) {
        for (int c = 0; c < C; ++c) navail += l.cn[c] != 0.f ? 1.f : 0.f;
    }
    for (int t = lane; t < T; t += 64) {
        const float* hr = l.h + t * Hd;
        float* xh = l.xhat + t * C;
        float mu = 0.f;
        for (int c = 0; c < C; ++c) {
            float a = l.rb[c];
            for (int k = 0; k < Hd; ++k) a = fmaf(l.Rw[c * Hd + k], hr[k], a);
            xh[c] = a;
            mu += a;
        }
        mu /= (float)C;
        float var = 0.f;
        for (int c = 0; c < C; ++c) { const float q = xh[c] - mu; var = fmaf(q, q, var); }
        const float rs = 1.0f / sqrtf(var / (float)C + 1e-5f);
        float c1 = 0.f, c2 = 0.f;
        for (int c = 0; c < C; ++c) {
            const size_t gi_ = (row0 + t) * C + c;
            const float hh = (xh[c] - mu) * rs;
            xh[c] = hh;
            const float sc = dropout_scale(drop, site, gi_);
            const float dd = fmaf(hh, l.gam[c], l.bet[c]) * sc;
            const float g = live ? sigm(l.P[t * PW + G3 + c]) : 1.f;
            const float y = l.Y[t * C + c];
            const float out = g * y + (1.f - g) * (y + dd);
            if (Yout) Yout[gi_] = out;
            if (TRAIN) {
                const float dlt = l.tr[t * C + c] - out, m = l.mk[t * C + c], den = l.cn[c] + 1e-8f;
                e += dlt * dlt * m / den;
                const float go = -dlt * m * (grad_scale * 2.f / (den * navail));       // d loss / d out
                l.dout[t * C + c] = go;
                // out = y + (1 - g) dd
                l.P[t * PW + G3 + c] = live ? (-dd * go) * g * (1.f - g) : 0.f;         // d gate logit (in place of the logit)
                const float gn = (1.f - g) * go * sc;                                    // d LayerNorm output
                l.dd[t * C + c] = gn;
                const float q = gn * l.gam[c];
                c1 += q;
                c2 = fmaf(q, hh, c2);
            }
        }
        if (TRAIN) {
            c1 /= (float)C;
            c2 /= (float)C;
            for (int c = 0; c < C; ++c) l.g[t * C + c] = rs * (l.dd[t * C + c] * l.gam[c] - c1 - xh[c] * c2);      // d delta
            for (int k = 0; k < Hd; ++k) {
                float a = 0.f;
                for (int c = 0; c < C; ++c) a = fmaf(l.Rw[c * Hd + k], l.g[t * C + c], a);
                l.dhin[t * Hd + k] = a;
            }
        }
    }
    if (!TRAIN) return;
    __syncthreads();
    // ---- E: back through time (hidden unit j on lane j; this lane's COLUMN of W_hh in registers, the step's d gh read from the lanes
    // that formed it); d gi goes where gi was, d gh to its own tile
    {
        float wc[3 * HM];
#pragma unroll
        for (int g = 0; g < HM; ++g) {
            const bool on = lane < Hd && g < Hd;
            wc[g] = on ? l.Whh[g * Hd + lane] : 0.f;
            wc[HM + g] = on ? l.Whh[(Hd + g) * Hd + lane] : 0.f;
            wc[2 * HM + g] = on ? l.Whh[(2 * Hd + g) * Hd + lane] : 0.f;
        }
        float dh_carry = 0.f;
        const int jl = lane < Hd ? lane : 0;
        int o1 = (T - 1) * Hd + jl;
        float n_dh = l.dhin[o1], n_r = l.r[o1], n_z = l.z[o1], n_n = l.n[o1], n_hn = l.hn[o1], n_hp = l.hp[o1];      // (one step ahead)
        for (int t = T - 1; t >= 0; --t) {
            float dar = 0.f, daz = 0.f, dhn = 0.f, dhz = 0.f;
            const float c_dh = n_dh, r = n_r, z = n_z, n = n_n, hn = n_hn, hp = n_hp;
            if (t > 0) { o1 = (t - 1) * Hd + jl; n_dh = l.dhin[o1]; n_r = l.r[o1]; n_z = l.z[o1]; n_n = l.n[o1]; n_hn = l.hn[o1]; n_hp = l.hp[o1]; }
            if (lane < Hd) {
                const float dh = c_dh + dh_carry;
                const float dn = dh * (1.f - z), dz = dh * (hp - n);
                const float dan = dn * (1.f - n * n);
                daz = dz * z * (1.f - z); dar = dan * hn * r * (1.f - r); dhn = dan * r;
                float* dgi = l.P + t * PW;
                dgi[lane] = dar; dgi[Hd + lane] = daz; dgi[2 * Hd + lane] = dan;
                float* dgh = l.dgh + t * G3;
                dgh[lane] = dar; dgh[Hd + lane] = daz; dgh[2 * Hd + lane] = dhn;
                dhz = dh * z;
            }
            float a = 0.f;
#pragma unroll
            for (int g = 0; g < HM; ++g) {
                if (g < Hd) {                    // (wave-uniform)
                    a = fmaf(wc[g], lane_bcast(dar, g), a);
                    a = fmaf(wc[HM + g], lane_bcast(daz, g), a);
                    a = fmaf(wc[2 * HM + g], lane_bcast(dhn, g), a);
                }
            }
            dh_carry = dhz + a;
        }
    }
    __syncthreads();
    // ---- F: what leaves.  dY first (the backbone's backward waits for it), then dP, then the parameter gradients (atomics into zeroed
    // buffers: one add per element and window)
    for (int i = lane; i < T * C; i += 64) {
        const int t = i / C, c = i - t * C;
        float a = l.dout[i];
        const float* dgi = l.P + t * PW;
        for (int j = 0; j < G3; ++j) a = fmaf(dgi[j], l.Wy[j * C + c], a);
        for (int c2 = 0; c2 < C; ++c2) a = fmaf(dgi[G3 + c2], l.Gy[c2 * C + c], a);
        dY[row0 * C + i] = a;
    }
    if (done_flag) {
        __threadfence();
        __syncthreads();
        if (lane == 0) {
            const unsigned int old = atomicAdd(ticket + 1, 1u);
            if (old == (unsigned int)dm.B - 1) {
                ticket[1] = 0u;
                __threadfence();
                __hip_atomic_store(done_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    for (int i = lane; i < T * PW; i += 64) {
        const int j = i % PW;
        dP[row0 * PW + i] = j < G3 + C ? l.P[i] : 0.f;
    }
    for (int i = lane; i < G3 * C; i += 64) {           // d W_ih[:, :C] = sum_t d gi[t, j] Y[t, c]
        const int j = i / C, c = i - j * C;
        float a = 0.f;
        for (int t = 0; t < T; ++t) a = fmaf(l.P[t * PW + j], l.Y[t * C + c], a);
        atomicAdd(gr.w_ih + (size_t)j * dm.ld + c, a);
    }
    for (int i = lane; i < C * C; i += 64) {            // d W_g[:, :C]
        const int j = i / C, c = i - j * C;
        float a = 0.f;
        for (int t = 0; t < T; ++t) a = fmaf(l.P[t * PW + G3 + j], l.Y[t * C + c], a);
        atomicAdd(gr.gate_w + (size_t)j * dm.ld + c, a);
    }
    for (int i = lane; i < G3 * Hd; i += 64) {          // d W_hh = sum_t d gh[t, g] h_prev[t, k]
        const int g = i / Hd, k = i - g * Hd;
        float a = 0.f;
        for (int t = 0; t < T; ++t) a = fmaf(l.dgh[t * G3 + g], l.hp[t * Hd + k], a);
        atomicAdd(gr.w_hh + i, a);
    }
    for (int i = lane; i < G3; i += 64) {
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += l.dgh[t * G3 + i];
        atomicAdd(gr.b_hh + i, a);
    }
    for (int i = lane; i < C * Hd; i += 64) {           // d W_r = sum_t d delta[t, c] h[t, k]
        const int c = i / Hd, k = i - c * Hd;
        float a = 0.f;
        for (int t = 0; t < T; ++t) a = fmaf(l.g[t * C + c], l.h[t * Hd + k], a);
        atomicAdd(gr.res_w + i, a);
    }
    for (int c = lane; c < C; c += 64) {
        float a = 0.f, gw = 0.f, gb = 0.f;
        for (int t = 0; t < T; ++t) {
            a += l.g[t * C + c];
            gw = fmaf(l.dd[t * C + c], l.xhat[t * C + c], gw);
            gb += l.dd[t * C + c];
        }
        atomicAdd(gr.res_b + c, a);
        atomicAdd(gr.ln_w + c, gw);
        atomicAdd(gr.ln_b + c, gb);
    }
    // ---- G: the loss: per-window partials, summed in window order by the last workgroup to arrive
    e = wave_sum(e);
    if (lane == 0) {
        partial[b] = e;
        __threadfence();
        const unsigned int old = atomicAdd(ticket, 1u);
        if (old == (unsigned int)dm.B - 1) {
            __threadfence();
            float tot = 0.f;
            for (int k = 0; k < dm.B; ++k) tot += __builtin_nontemporal_load(partial + k);
            loss[0] = tot / navail;
            *ticket = 0u;
        }
    }
}

// W_cat (PW x d) = [W_ih[:, C:] ; W_g[:, C:] ; 0], b_cat (PW) = [b_ih ; b_g ; 0]  (fp32 + bf16 image)
__global__ __launch_bounds__(256) void gr_pack_kernel(int C, int Hd, int d, int PW, const float* __restrict__ w_ih, const float* __restrict__ b_ih,
                                                       const float* __restrict__ gate_w, const float* __restrict__ gate_b, float* __restrict__ Wc,
                                                       bf16_t* __restrict__ Wc16, float* __restrict__ bc) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int G3 = 3 * Hd, ld = C + d;
    if (i < (long)PW * d) {
        const int j = (int)(i / d), k = (int)(i - (long)j * d);
        const float v = j < G3 ? w_ih[(size_t)j * ld + C + k] : j < G3 + C ? gate_w[(size_t)(j - G3) * ld + C + k] : 0.f;
        Wc[i] = v;
        if (Wc16) Wc16[i] = (bf16_t)v;
    }
    if (i < PW) bc[i] = i < G3 ? b_ih[i] : i < G3 + C ? gate_b[i - G3] : 0.f;
}
// the text columns of d W_ih / d W_g and the two bias gradients out of d W_cat (PW x d) / d b_cat
__global__ __launch_bounds__(256) void gr_unpack_kernel(int C, int Hd, int d, int PW, const float* __restrict__ dWc, const float* __restrict__ dbc,
                                                         float* __restrict__ g_w_ih, float* __restrict__ g_b_ih, float* __restrict__ g_gate_w,
                                                         float* __restrict__ g_gate_b) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int G3 = 3 * Hd, ld = C + d;
    if (i < (long)(G3 + C) * d) {
        const int j = (int)(i / d), k = (int)(i - (long)j * d);
        if (j < G3) g_w_ih[(size_t)j * ld + C + k] = dWc[i];
        else g_gate_w[(size_t)(j - G3) * ld + C + k] = dWc[i];
    }
    if (i < G3) g_b_ih[i] = dbc[i];
    else if (i < G3 + C) g_gate_b[i - G3] = dbc[i];
}

inline int gr_pw(int C, int Hd) { return (3 * Hd + C + 7) & ~7; }
inline bool gr_split_ok(const immtsf_fusion_cfg* c, int Hd) {
    return c && c->B > 0 && c->T >= 1 && c->T <= 64 && c->C >= 1 && c->C <= 16 && Hd >= 1 && Hd <= 16 && c->d >= 8 && (c->d % 8) == 0 &&
           gr_lds_floats(c->T, c->C, Hd, gr_pw(c->C, Hd)) * sizeof(float) <= 150 * 1024;
}

struct GRPWs { float *Wc, *bc; void *Wc16, *E16; size_t bytes; };
GRPWs carve_grp(const immtsf_fusion_cfg* c, int Hd, void* base) {
    const size_t PW = gr_pw(c->C, Hd), d = c->d, BT = (size_t)c->B * c->T;
    const bool hf = c->precision == 1;
    Carver k(base);
    GRPWs w;
    w.Wc = k.take<float>(PW * d);
    w.bc = k.take<float>(PW);
    w.Wc16 = hf ? k.take<unsigned short>(PW * d) : nullptr;
    w.E16 = hf ? k.take<unsigned short>(BT * d) : nullptr;
    w.bytes = k.bytes();
    return w;
}
struct GRPScratch { float *dWc, *dbc; void *dP16; size_t bytes; };
GRPScratch carve_grp_scratch(const immtsf_fusion_cfg* c, int Hd, void* base) {
    const size_t PW = gr_pw(c->C, Hd), d = c->d, BT = (size_t)c->B * c->T;
    Carver k(base);
    GRPScratch s;
    s.dWc = k.take<float>(PW * d);
    s.dbc = k.take<float>(PW);
    s.dP16 = c->precision == 1 ? k.take<unsigned short>(BT * PW) : nullptr;
    s.bytes = k.bytes();
    return s;
}

}  // namespace

extern "C" {

int32_t immtsf_mmf_gr_pw(const immtsf_fusion_cfg* cfg, int32_t hidden) { return gr_split_ok(cfg, hidden) ? gr_pw(cfg->C, hidden) : 0; }

size_t immtsf_mmf_gr_p_workspace_bytes(const immtsf_fusion_cfg* cfg, int32_t hidden) {
    return gr_split_ok(cfg, hidden) ? carve_grp(cfg, hidden, nullptr).bytes : 0;
}
size_t immtsf_mmf_gr_p_scratch_bytes(const immtsf_fusion_cfg* cfg, int32_t hidden) {
    return gr_split_ok(cfg, hidden) ? carve_grp_scratch(cfg, hidden, nullptr).bytes : 0;
}

int immtsf_mmf_gr_p_forward(const immtsf_fusion_cfg* cfg, int32_t Hd, const immtsf_gr_params* p, const float* E_txt, float* P, void* workspace,
                            size_t workspace_bytes, immtsf_stream_t stream) {
    if (!gr_split_ok(cfg, Hd) || !p || !E_txt || !P || !workspace) return IMMTSF_EINVAL;
    GRPWs w = carve_grp(cfg, Hd, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int C = cfg->C, d = cfg->d, BT = cfg->B * cfg->T, PW = gr_pw(C, Hd);
    const bool hf = cfg->precision == 1;
    const long n = (long)PW * d;
    hipLaunchKernelGGL(gr_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, C, Hd, d, PW, p->w_ih, p->b_ih, p->gate_w, p->gate_b, w.Wc,
                       static_cast<bf16_t*>(w.Wc16), w.bc);
    IMMTSF_LAUNCH_CHECK();
    Mat Em = cmat(E_txt);
    if (hf && cfg->in_h) Em.h = const_cast<void*>(cfg->in_h);
    else if (hf) { CHECK(launch_f32_to_bf16(E_txt, w.E16, (size_t)BT * d, s)); Em.h = w.E16; }
    GemmArgs g = gemm_args(BT, PW, d, d, d, PW);
    set_problem2(g, 0, Em, mat(w.Wc, w.Wc16), mat(P), w.bc);
    return immtsf_launch_gemm(GEMM_NT, cfg->precision, g, s);
}

/* dP (B T, PW) -> dE_txt (B T, d) and the text columns of d W_ih / d W_g, d b_ih, d b_g (written; the Y columns of the two weight
 * gradients are immtsf_mmf_gr_q_train's) */
int immtsf_mmf_gr_p_backward(const immtsf_fusion_cfg* cfg, int32_t Hd, const immtsf_gr_params* p, const float* E_txt, const float* dP, float* dE_txt,
                             void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes, const immtsf_gr_params* gr,
                             immtsf_stream_t stream) {
    if (!gr_split_ok(cfg, Hd) || !p || !gr || !E_txt || !dP || !dE_txt || !workspace || !scratch || !gr->w_ih || !gr->b_ih || !gr->gate_w || !gr->gate_b)
        return IMMTSF_EINVAL;
    GRPWs w = carve_grp(cfg, Hd, workspace);
    GRPScratch sc = carve_grp_scratch(cfg, Hd, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int C = cfg->C, d = cfg->d, BT = cfg->B * cfg->T, PW = gr_pw(C, Hd), prec = cfg->precision;
    const bool hf = prec == 1;
    Mat dPm = cmat(dP), Em = cmat(E_txt, hf ? (cfg->aux_h ? cfg->aux_h : static_cast<const void*>(w.E16)) : nullptr);
    if (hf) { CHECK(launch_f32_to_bf16(dP, sc.dP16, (size_t)BT * PW, s)); dPm.h = sc.dP16; }
    // dE = dP W_cat: many rows of rank PW
    if (rank_expand_ok(BT, d, PW, w.Wc, d, dE_txt, hf ? cfg->out_h : nullptr)) {
        CHECK(launch_rank_expand(dP, PW, w.Wc, d, dE_txt, hf ? cfg->out_h : nullptr, BT, d, PW, s));
    } else {
        GemmArgs g = gemm_args(BT, d, PW, PW, d, d);
        set_problem2(g, 0, dPm, mat(w.Wc, w.Wc16), mat(dE_txt, hf ? cfg->out_h : nullptr), nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    {   // d W_cat = dP^T E, d b_cat = column sums of dP
        GemmArgs h = gemm_args(PW, d, BT, PW, d, d);
        set_problem2(h, 0, dPm, Em, mat(sc.dWc), nullptr, sc.dbc);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, s));
    }
    const long n = (long)(3 * Hd + C) * d;
    hipLaunchKernelGGL(gr_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, C, Hd, d, PW, sc.dWc, sc.dbc, gr->w_ih, gr->b_ih, gr->gate_w,
                       gr->gate_b);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

/* the Y half.  TRAIN (truth != NULL): forward + the masked-MSE loss of immtsf_masked_mse_counted + the backward of both in ONE launch:
 * loss (1 float), dY_ts, dP, and -- ADDED by atomics into buffers the caller hands in zeroed -- the Y columns of d W_ih / d W_g, d W_hh,
 * d b_hh, d res_w, d res_b, d ln_w, d ln_b.  Else (truth == NULL): Y_out only.  scratch: B floats; ticket: two zero-initialised words
 * the call leaves zero; done_flag (optional): set to 1 as soon as dY_ts is complete. */
int immtsf_mmf_gr_q_train(const immtsf_fusion_cfg* cfg, int32_t Hd, const immtsf_gr_params* p, const float* Y_ts, const float* P,
                          const uint8_t* M_txt, const float* truth, const float* mask, const float* cnt, float grad_scale, float* Y_out,
                          float* loss, float* dY_ts, float* dP, const immtsf_gr_params* gr, float* scratch, uint32_t* ticket, int32_t* done_flag,
                          immtsf_stream_t stream) {
    if (!gr_split_ok(cfg, Hd) || !p || !Y_ts || !P || !M_txt || !p->w_ih || !p->w_hh || !p->b_hh || !p->res_w || !p->res_b || !p->gate_w || !p->ln_w ||
        !p->ln_b)
        return IMMTSF_EINVAL;
    const bool train = truth != nullptr;
    if (train && (!mask || !cnt || !loss || !dY_ts || !dP || !gr || !scratch || !ticket || !gr->w_ih || !gr->w_hh || !gr->b_hh || !gr->res_w ||
                  !gr->res_b || !gr->gate_w || !gr->ln_w || !gr->ln_b))
        return IMMTSF_EINVAL;
    if (!train && !Y_out) return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const GRTDims dm{cfg->B, cfg->T, cfg->C, Hd, gr_pw(cfg->C, Hd), cfg->C + cfg->d};
    const GRTP q{p->w_ih, p->w_hh, p->b_hh, p->res_w, p->res_b, p->gate_w, p->ln_w, p->ln_b};
    const DropCfg drop = drop_of(cfg);
    const size_t lds = gr_lds_floats(dm.T, dm.C, dm.Hd, dm.PW) * sizeof(float);
    if (train) {
        const GRTG g{gr->w_ih, gr->w_hh, gr->b_hh, gr->res_w, gr->res_b, gr->gate_w, gr->ln_w, gr->ln_b};
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gr_train_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(gr_train_kernel<true>, dim3(dm.B), dim3(64), lds, s, dm, q, Y_ts, P, M_txt, truth, mask, cnt, grad_scale, Y_out, loss, scratch,
                           ticket, dY_ts, dP, g, drop, SITE_GR_OUT, done_flag);
    } else {
        const GRTG g{};
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gr_train_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(gr_train_kernel<false>, dim3(dm.B), dim3(64), lds, s, dm, q, Y_ts, P, M_txt, nullptr, nullptr, nullptr, 0.f, Y_out, nullptr,
                           nullptr, nullptr, nullptr, nullptr, g, drop, SITE_GR_OUT, nullptr);
    }
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

}  // extern "C"
