// Fused tails of the MMF blocks + loss + optimizer.  All HBM/latency-bound: one thread or one wave per row.
#include "tail.hpp"
#include "gemm.hpp"

namespace {

__global__ __launch_bounds__(256) void mask_rows_kernel(float* __restrict__ x, int rows, int d,
                                                         const unsigned char* __restrict__ flag, int div, bf16_t* __restrict__ xh) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bool live = flag[row / div] != 0;
    if (live && !xh) return;
    for (int i = lane; i < d; i += 64) {
        const size_t o = (size_t)row * d + i;
        if (!live) x[o] = 0.f;
        if (xh) xh[o] = live ? (bf16_t)x[o] : (bf16_t)0.f;
    }
}

// reference: fusions/MMF_XAttn_Add.py:93-102
__global__ __launch_bounds__(256) void ln_blend_fwd_kernel(const float* __restrict__ delta, const float* __restrict__ Y,
                                                            const unsigned char* __restrict__ mtxt, int BT, int T, int C,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float kappa, float* __restrict__ xhat, float* __restrict__ rstd,
                                                            float* __restrict__ Yout, DropCfg drop, uint64_t site) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= BT) return;
    const bool live = mtxt[row / T] != 0;
    const float* x = delta + (size_t)row * C;
    float mu = 0.f;
    for (int c = 0; c < C; ++c) mu += x[c];
    mu /= (float)C;
    float var = 0.f;
    for (int c = 0; c < C; ++c) { const float t = x[c] - mu; var = fmaf(t, t, var); }
    const float rs = 1.0f / sqrtf(var / (float)C + 1e-5f);
    rstd[row] = rs;
    const float inv = 1.f / (1.f + kappa);
    for (int c = 0; c < C; ++c) {
        const float h = (x[c] - mu) * rs;
        xhat[(size_t)row * C + c] = h;
        float y = fmaf(h, gamma[c], beta[c]) * dropout_scale(drop, site, (uint64_t)row * C + c);
        if (!live) y = 0.f;
        Yout[(size_t)row * C + c] = (Y[(size_t)row * C + c] + kappa * y) * inv;
    }
}

__global__ __launch_bounds__(256) void ln_blend_bwd_kernel(const float* __restrict__ dYout, const unsigned char* __restrict__ mtxt,
                                                            int BT, int T, int C, const float* __restrict__ gamma,
                                                            const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                            float kappa, float* __restrict__ dY, float* __restrict__ dn,
                                                            float* __restrict__ ddelta, DropCfg drop, uint64_t site) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= BT) return;
    const float live = mtxt[row / T] ? 1.f : 0.f;
    const float inv = 1.f / (1.f + kappa), kk = kappa * inv * live;
    const float* g = dYout + (size_t)row * C;
    const float* h = xhat + (size_t)row * C;
    float c1 = 0.f, c2 = 0.f;
    for (int c = 0; c < C; ++c) {
        const float gn = kk * g[c] * dropout_scale(drop, site, (uint64_t)row * C + c);
        dn[(size_t)row * C + c] = gn;
        dY[(size_t)row * C + c] = g[c] * inv;
        const float t = gn * gamma[c];
        c1 += t;
        c2 = fmaf(t, h[c], c2);
    }
    c1 /= (float)C;
    c2 /= (float)C;
    const float rs = rstd[row];
    for (int c = 0; c < C; ++c)
        ddelta[(size_t)row * C + c] = rs * (dn[(size_t)row * C + c] * gamma[c] - c1 - h[c] * c2);
}

// C <= 64: one lane per element, CT = pow2 >= C lanes per row, the two row sums by xor-shuffles (the kernel above walks a
// row serially in one thread: 8 workgroups, 8 us at 2048 x 8)
__global__ __launch_bounds__(256) void ln_blend_bwd_lanes_kernel(const float* __restrict__ dYout, const unsigned char* __restrict__ mtxt,
                                                                  int BT, int T, int C, int CT, const float* __restrict__ gamma,
                                                                  const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                                  float kappa, float* __restrict__ dY, float* __restrict__ dn,
                                                                  float* __restrict__ ddelta, DropCfg drop, uint64_t site) {
    const int t = blockIdx.x * 256 + threadIdx.x, row = t / CT, c = t - row * CT;
    const bool on = row < BT && c < C;
    float gn = 0.f, tt = 0.f, h = 0.f, gm = 0.f;
    if (on) {
        const size_t o = (size_t)row * C + c;
        const float inv = 1.f / (1.f + kappa), kk = kappa * inv * (mtxt[row / T] ? 1.f : 0.f), g = dYout[o];
        gn = kk * g * dropout_scale(drop, site, (uint64_t)o);
        dn[o] = gn;
        dY[o] = g * inv;
        gm = gamma[c];
        h = xhat[o];
        tt = gn * gm;
    }
    float c1 = tt, c2 = tt * h;
    c1 = group_sum(c1, CT);
    c2 = group_sum(c2, CT);
    if (on) ddelta[(size_t)row * C + c] = rstd[row] * (tt - c1 / (float)C - h * (c2 / (float)C));
}

// ---- masked MSE (lib/evaluation.py:17-62, func="MSE", reduce="mean") -------------------------------------
// stage 1: grid = 64 row slabs; block = CT column lanes x (256/CT) row lanes (CT = pow2 >= min(C,64)), so one
// wave-instruction reads whole contiguous rows.  partial[slab][0][c] = sum err, partial[slab][1][c] = sum mask.
constexpr int kMseSlabs = 64;
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ truth, const float* __restrict__ pred,
                                                           const float* __restrict__ mask, int rows, int C, int CT,
                                                           float* __restrict__ partial) {
    __shared__ float re[256], rc[256];
    const int RT = 256 / CT, tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    const int rps = (rows + kMseSlabs - 1) / kMseSlabs;
    const int r0 = blockIdx.x * rps, r1 = min(rows, r0 + rps);
    for (int c0 = 0; c0 < C; c0 += CT) {
        const int c = c0 + tx;
        float e = 0.f, n = 0.f;
        if (c < C)
            for (int r = r0 + ty; r < r1; r += RT) {
                const size_t i = (size_t)r * C + c;
                const float dlt = truth[i] - pred[i], m = mask[i];
                e = fmaf(dlt * dlt, m, e);
                n += m;
            }
        re[threadIdx.x] = e;
        rc[threadIdx.x] = n;
        __syncthreads();
        if (ty == 0 && c < C) {
            float se = 0.f, sn = 0.f;
            for (int k = 0; k < RT; ++k) { se += re[k * CT + tx]; sn += rc[k * CT + tx]; }
            partial[((size_t)blockIdx.x * 2 + 0) * C + c] = se;
            partial[((size_t)blockIdx.x * 2 + 1) * C + c] = sn;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void mse_sums_final_kernel(const float* __restrict__ partial, int C, float* __restrict__ err_sum,
                                                              float* __restrict__ cnt) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float se = 0.f, sn = 0.f;
    for (int s = 0; s < kMseSlabs; ++s) {
        se += partial[((size_t)s * 2 + 0) * C + c];
        sn += partial[((size_t)s * 2 + 1) * C + c];
    }
    err_sum[c] = se;
    cnt[c] = sn;
}

__global__ __launch_bounds__(256) void mse_finish_kernel(const float* __restrict__ truth, const float* __restrict__ pred,
                                                          const float* __restrict__ mask, int rows, int C,
                                                          const float* __restrict__ err_sum, const float* __restrict__ cnt,
                                                          float* __restrict__ loss, float* __restrict__ dpred, float grad_scale) {
    // every block recomputes the (tiny) per-variable reduction; block 0 writes the loss
    float tot = 0.f, navail = 0.f;
    for (int c = 0; c < C; ++c) {
        tot += err_sum[c] / (cnt[c] + 1e-8f);
        navail += (cnt[c] != 0.f) ? 1.f : 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && loss) loss[0] = tot / navail;
    if (!dpred) return;
    const size_t n = (size_t)rows * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        dpred[i] = grad_scale * 2.f * (pred[i] - truth[i]) * mask[i] / ((cnt[c] + 1e-8f) * navail);
    }
}

// the three stages above in ONE workgroup, for the usual case of a prediction tensor of a few thousand elements and no
// cross-rank reduction between the sums and the divide (the three launches are ~16 us of latency on the step's
// critical path between the forward and the backward; this is one).  cnt_in != null: per-variable counts of the
// global batch (data parallel without a collective in the step); err_sum / cnt are also written for the caller.
__global__ __launch_bounds__(1024) void mse_small_kernel(const float* __restrict__ truth, const float* __restrict__ pred,
                                                          const float* __restrict__ mask, int rows, int C, int CT,
                                                          const float* __restrict__ cnt_in, float* __restrict__ err_sum,
                                                          float* __restrict__ cnt, float* __restrict__ loss,
                                                          float* __restrict__ dpred, float grad_scale) {
    extern __shared__ float sm[];              // se[C] | sn[C] | re[1024] | rc[1024]
    float *se = sm, *sn = sm + C, *re = sm + 2 * C, *rc = re + 1024;
    __shared__ float s_nav;
    const int RT = 1024 / CT, tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    // CT == C (C a power of two) and <= 16 K elements: thread t touches exactly the elements t + 1024 k in both passes, so the
    // masked differences stay in registers for the gradient pass (no second read of the three tensors)
    constexpr int KEEP = 16;
    const bool keep = CT == C && (size_t)rows * C <= (size_t)KEEP * 1024;
    float dm[KEEP];
#pragma unroll
    for (int k = 0; k < KEEP; ++k) dm[k] = 0.f;
    for (int c0 = 0; c0 < C; c0 += CT) {
        const int c = c0 + tx;
        float e = 0.f, n = 0.f;
        if (c < C && keep) {
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                const int r = ty + k * RT;
                if (r < rows) {
                    const size_t i = (size_t)r * C + c;
                    const float dlt = truth[i] - pred[i], m = mask[i];
                    e = fmaf(dlt * dlt, m, e);
                    n += m;
                    dm[k] = -dlt * m;
                }
            }
        } else if (c < C)
#pragma unroll 8
            for (int r = ty; r < rows; r += RT) {       // independent loads: keep 8 rows in flight (one CU does all the work)
                const size_t i = (size_t)r * C + c;
                const float dlt = truth[i] - pred[i], m = mask[i];
                e = fmaf(dlt * dlt, m, e);
                n += m;
            }
        if (CT <= 32) {
            // the row lanes of one column sit CT lanes apart inside a wave: xor-shuffle over the offsets >= CT first (no barrier),
            // then one LDS round over the 16 waves -- the 7-step tree with a 1024-thread barrier per step was half of this kernel
            e = coset_sum(e, CT);
            n = coset_sum(n, CT);
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            if (lane < CT) { re[wave * CT + lane] = e; rc[wave * CT + lane] = n; }
            __syncthreads();
            if (threadIdx.x < CT) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w = 0; w < 16; ++w) { a += re[w * CT + threadIdx.x]; b += rc[w * CT + threadIdx.x]; }
                re[threadIdx.x] = a;
                rc[threadIdx.x] = b;
            }
            __syncthreads();
        } else {
            re[threadIdx.x] = e;
            rc[threadIdx.x] = n;
            __syncthreads();
            for (int st = RT >> 1; st > 0; st >>= 1) {      // tree over the row lanes (RT is a power of two): log2(RT) steps, not RT
                if (ty < st) { re[threadIdx.x] += re[threadIdx.x + st * CT]; rc[threadIdx.x] += rc[threadIdx.x + st * CT]; }
                __syncthreads();
            }
        }
        if (ty == 0 && c < C) {
            const float a = re[tx], b = rc[tx];
            se[c] = a;
            sn[c] = cnt_in ? cnt_in[c] : b;
            if (err_sum) err_sum[c] = a;
            if (cnt) cnt[c] = b;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float tot = 0.f, navail = 0.f;
        for (int c = 0; c < C; ++c) {
            tot += se[c] / (sn[c] + 1e-8f);
            navail += (sn[c] != 0.f) ? 1.f : 0.f;
        }
        if (loss) loss[0] = tot / navail;
        s_nav = navail;
    }
    __syncthreads();
    if (!dpred) return;
    const float navail = s_nav;
    const size_t n = (size_t)rows * C;
    if (keep) {
        const float f = grad_scale * 2.f / ((sn[tx] + 1e-8f) * navail);
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const size_t i = (size_t)threadIdx.x + (size_t)k * 1024;
            if (i < n) dpred[i] = dm[k] * f;
        }
        return;
    }
#pragma unroll 8
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        const int c = (int)(i % C);
        dpred[i] = grad_scale * 2.f * (pred[i] - truth[i]) * mask[i] / ((sn[c] + 1e-8f) * navail);
    }
}

// ---- clip_grad_norm_ + Adam (main.py:1024,1098-1101) on one flat buffer -------------------------------------
// 16-byte loads (the flat buffers are 16-byte aligned; a tail of < 4 elements is handled by the last threads).  The
// first thread also advances the device-side step / dropout counters when given (graph replay: one launch less).
// skip (optional device word): non-zero = this step is DROPPED -- a spin of the flag engine timed out, i.e. some gradient may be
// incomplete (csrc/sync.hip): the step number is not advanced here and adam_kernel leaves parameters and moments alone
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, size_t n, int vec, float* __restrict__ part,
                                                              long long* step_dev, unsigned long long* drop_dev, const int* skip) {
    __shared__ float red[16];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (step_dev && !(skip && *skip)) step_dev[0] += 1;
        if (drop_dev) drop_dev[0] += 1;
    }
    float a = 0.f;
    const size_t n4 = vec ? n >> 2 : 0, stride = (size_t)gridDim.x * 256, t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = t0; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4*>(g)[i];
        a = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, a))));
    }
    for (size_t i = (n4 << 2) + t0; i < n; i += stride) a = fmaf(g[i], g[i], a);
    a = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}

// sqnorm_partial_kernel for a step whose optimizer pass sits at the head of the NEXT replay (immtsf.train.FlagStep): the gradient may
// come as its bf16 wire image (gh), and the first thread takes the step decision -- drop when no gradient is pending, when a flag wait
// of this rank timed out, or when the guard slot the last collective summed over the ranks is non-zero -- for the range updates that
// follow on the branches (adam_kernel reads *skip_out)
__global__ __launch_bounds__(256) void adam_prepare_kernel(const float* __restrict__ g, const bf16_t* __restrict__ gh, size_t n, int vec,
                                                            float* __restrict__ part, long long* step_dev, unsigned long long* drop_dev,
                                                            int* pending, const int* err, const bf16_t* guard_h, const float* guard_f,
                                                            int* skip_out) {
    __shared__ float red[16];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int skip = 0;
        if (pending) { if (*pending == 0) skip = 1; *pending = 0; }
        if (err && *err) skip = 1;
        if (guard_h && (float)guard_h[0] != 0.f) skip = 1;
        if (guard_f && guard_f[0] != 0.f) skip = 1;
        if (skip_out) *skip_out = skip;
        if (step_dev && !skip) step_dev[0] += 1;
        if (drop_dev) drop_dev[0] += 1;
    }
    float a = 0.f;
    const size_t stride = (size_t)gridDim.x * 256, t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gh) {
        const size_t n8 = vec ? n >> 3 : 0;
        for (size_t i = t0; i < n8; i += stride) {
            const bf16x8 x = reinterpret_cast<const bf16x8*>(gh)[i];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float f = (float)x[k]; a = fmaf(f, f, a); }
        }
        for (size_t i = (n8 << 3) + t0; i < n; i += stride) { const float f = (float)gh[i]; a = fmaf(f, f, a); }
    } else {
        const size_t n4 = vec ? n >> 2 : 0;
        for (size_t i = t0; i < n4; i += stride) {
            const float4 x = reinterpret_cast<const float4*>(g)[i];
            a = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, a))));
        }
        for (size_t i = (n4 << 2) + t0; i < n; i += stride) a = fmaf(g[i], g[i], a);
    }
    a = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}
// this rank's guard word in the gradient wire's element type, for the slot the step's last all-reduce sums over the ranks
__global__ void guard_pack_kernel(const int* err, void* slot, int is_bf16) {
    const float v = (*err != 0) ? 1.f : 0.f;
    if (is_bf16) *static_cast<bf16_t*>(slot) = (bf16_t)v;
    else *static_cast<float*>(slot) = v;
}

// (g and gz may be the same buffer -- read, then zeroed: neither is declared __restrict__)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2s, float max_norm,
                                                    const float* __restrict__ part, int nparts,
                                                    const long long* __restrict__ step_dev, bf16_t* __restrict__ twin, int vec,
                                                    float* gz, const int* skip, const bf16_t* __restrict__ gh = nullptr) {
    // gz != null (== g): the gradient is left zero behind the update -- the next step's zero-fill rides on this pass
    // gh != null: the gradient is read from its bf16 wire image (what a bf16 all-reduce left behind) instead of g
    __shared__ float red[16];
    if (skip && *skip) {          // dropped step (see sqnorm_partial_kernel): only the zero-fill happens
        if (gz) {
            const size_t stride = (size_t)gridDim.x * 256;
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) gz[i] = 0.f;
        }
        return;
    }
    if (step_dev) {   // bias corrections from the device-side step counter (already incremented for this step)
        const float st = (float)step_dev[0];
        bc1 = 1.f - powf(b1, st);
        bc2s = sqrtf(1.f - powf(b2, st));
    }
    float a = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
    const float total = sqrtf(block_sum(a, red));
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(1.f, max_norm / (total + 1e-6f));
    const float step = lr / bc1;
    auto upd = [&](float gi, float pi, float& mi, float& vi) -> float {
        gi *= coef;
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        mi = fmaf(b1, mi, (1.f - b1) * gi);
        vi = fmaf(b2, vi, (1.f - b2) * gi * gi);
        return pi - step * mi / (sqrtf(vi) / bc2s + eps);
    };
    // 16 bytes per lane on all seven streams (the flat buffers are 16-byte aligned), scalar tail
    const size_t n4 = vec ? n >> 2 : 0, stride = (size_t)gridDim.x * 256, t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = t0; i < n4; i += stride) {
        float4 g4;
        if (gh) {
            const bf16x4 h4 = reinterpret_cast<const bf16x4*>(gh)[i];
            g4 = make_float4((float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]);
        } else {
            g4 = reinterpret_cast<const float4*>(g)[i];
        }
        const float4 p4 = reinterpret_cast<const float4*>(p)[i];
        float4 m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i], o;
        o.x = upd(g4.x, p4.x, m4.x, v4.x); o.y = upd(g4.y, p4.y, m4.y, v4.y);
        o.z = upd(g4.z, p4.z, m4.z, v4.z); o.w = upd(g4.w, p4.w, m4.w, v4.w);
        reinterpret_cast<float4*>(m)[i] = m4;
        reinterpret_cast<float4*>(v)[i] = v4;
        reinterpret_cast<float4*>(p)[i] = o;
        if (gz) reinterpret_cast<float4*>(gz)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (twin) {          // bf16 twin of the parameters (the GEMMs' weight operand), kept current here
            bf16x4 h;
            h[0] = (bf16_t)o.x; h[1] = (bf16_t)o.y; h[2] = (bf16_t)o.z; h[3] = (bf16_t)o.w;
            reinterpret_cast<bf16x4*>(twin)[i] = h;
        }
    }
    for (size_t i = (n4 << 2) + t0; i < n; i += stride) {
        float mi = m[i], vi = v[i];
        const float pn = upd(gh ? (float)gh[i] : g[i], p[i], mi, vi);
        m[i] = mi; v[i] = vi; p[i] = pn;
        if (gz) gz[i] = 0.f;
        if (twin) twin[i] = (bf16_t)pn;
    }
}

// fp32 <-> bf16 streams, 16 bytes of fp32 per lane when both pointers allow it (twins, the bf16 gradient wire)
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n, int vec) {
    const size_t n4 = vec ? n >> 2 : 0, stride = (size_t)gridDim.x * 256, t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = t0; i < n4; i += stride) {
        const float4 x = reinterpret_cast<const float4*>(src)[i];
        bf16x4 h;
        h[0] = (bf16_t)x.x; h[1] = (bf16_t)x.y; h[2] = (bf16_t)x.z; h[3] = (bf16_t)x.w;
        reinterpret_cast<bf16x4*>(dst)[i] = h;
    }
    for (size_t i = (n4 << 2) + t0; i < n; i += stride) dst[i] = (bf16_t)src[i];
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, size_t n, int vec) {
    const size_t n4 = vec ? n >> 2 : 0, stride = (size_t)gridDim.x * 256, t0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = t0; i < n4; i += stride) {
        const bf16x4 h = reinterpret_cast<const bf16x4*>(src)[i];
        reinterpret_cast<float4*>(dst)[i] = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    }
    for (size_t i = (n4 << 2) + t0; i < n; i += stride) dst[i] = (float)src[i];
}

}  // namespace

int launch_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t s) {
    if (n == 0) return IMMTSF_OK;
    const unsigned blocks = (unsigned)((n / 4 + 255) / 256 > 4096 ? 4096 : (n / 4 + 255) / 256 + 1);
    const int vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, s, src, static_cast<bf16_t*>(dst), n, vec);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_bf16_to_f32(const void* src, float* dst, size_t n, hipStream_t s) {
    if (n == 0) return IMMTSF_OK;
    const unsigned blocks = (unsigned)((n / 4 + 255) / 256 > 4096 ? 4096 : (n / 4 + 255) / 256 + 1);
    const int vec = (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 7) == 0;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const bf16_t*>(src), dst, n, vec);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_mask_rows(float* x, int rows, int d, const unsigned char* flag, int div, hipStream_t s, void* xh) {
    if (rows <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(mask_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, d, flag, div, static_cast<bf16_t*>(xh));
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_ln_blend_fwd(const float* delta, const float* Y, const unsigned char* mtxt, int BT, int T, int C,
                        const float* gamma, const float* beta, float kappa, float* xhat, float* rstd, float* Yout,
                        DropCfg drop, uint64_t site, hipStream_t s) {
    if (BT <= 0) return IMMTSF_OK;
    hipLaunchKernelGGL(ln_blend_fwd_kernel, dim3(cdiv(BT, 256)), dim3(256), 0, s, delta, Y, mtxt, BT, T, C, gamma, beta, kappa,
                       xhat, rstd, Yout, drop, site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_ln_blend_bwd(const float* dYout, const unsigned char* mtxt, int BT, int T, int C, const float* gamma,
                        const float* xhat, const float* rstd, float kappa, float* dY, float* dn, float* ddelta,
                        DropCfg drop, uint64_t site, hipStream_t s) {
    if (BT <= 0) return IMMTSF_OK;
    if (C <= 64) {
        int CT = 1;
        while (CT < C) CT <<= 1;
        hipLaunchKernelGGL(ln_blend_bwd_lanes_kernel, dim3((unsigned)(((long)BT * CT + 255) / 256)), dim3(256), 0, s, dYout, mtxt, BT, T, C, CT, gamma, xhat,
                           rstd, kappa, dY, dn, ddelta, drop, site);
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    hipLaunchKernelGGL(ln_blend_bwd_kernel, dim3(cdiv(BT, 256)), dim3(256), 0, s, dYout, mtxt, BT, T, C, gamma, xhat, rstd, kappa,
                       dY, dn, ddelta, drop, site);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_mse_sums(const float* truth, const float* pred, const float* mask, int rows, int C, float* err_sum,
                    float* cnt, float* scratch, hipStream_t s) {
    int CT = 1;
    while (CT < C && CT < 64) CT <<= 1;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(kMseSlabs), dim3(256), 0, s, truth, pred, mask, rows, C, CT, scratch);
    IMMTSF_LAUNCH_CHECK();
    hipLaunchKernelGGL(mse_sums_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, scratch, C, err_sum, cnt);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// With the per-variable counts known beforehand (they depend on the mask only: data pipelines reduce them when the batch is
// built) nothing in the gradient waits for a reduction, so the work spreads over several workgroups: every element's gradient
// is final on first touch, the loss is the sum of per-workgroup partials that the LAST workgroup to finish (ticket counter)
// adds up in index order -- deterministic, no zero-fill, the ticket is left at zero for the next call.  C <= 64.
constexpr int MSE_PARTS = 64;
__global__ __launch_bounds__(256) void mse_counted_kernel(const float* __restrict__ truth, const float* __restrict__ pred,
                                                           const float* __restrict__ mask, size_t n, int C,
                                                           const float* __restrict__ cnt, float* __restrict__ partial,
                                                           unsigned int* __restrict__ ticket, float* __restrict__ loss,
                                                           float* __restrict__ dpred, float grad_scale) {
    __shared__ float s_red[4], s_nav;
    __shared__ unsigned int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0) {
        const bool live = lane < C && cnt[lane] != 0.f;
        const unsigned long long b = __ballot(live);
        if (lane == 0) s_nav = (float)__popcll(b);
    }
    __syncthreads();
    const float navail = s_nav;
    float e = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const float dlt = truth[i] - pred[i], m = mask[i], den = cnt[c] + 1e-8f;
        e += dlt * dlt * m / den;
        if (dpred) dpred[i] = -dlt * m * (grad_scale * 2.f / (den * navail));
    }
    e = wave_sum(e);
    if (lane == 0) s_red[wave] = e;
    __syncthreads();
    if (tid == 0) {
        partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (s_last && tid == 0) {
        __threadfence();
        float tot = 0.f;
        for (unsigned int k = 0; k < gridDim.x; ++k) tot += __builtin_nontemporal_load(partial + k);
        if (loss) loss[0] = tot / navail;
        *ticket = 0u;
    }
}

int launch_mse_counted(const float* truth, const float* pred, const float* mask, int rows, int C, const float* cnt, float* partial,
                       unsigned int* ticket, float* loss, float* dpred, float grad_scale, hipStream_t s) {
    const size_t n = (size_t)rows * C;
    int grid = (int)((n + 1023) / 1024);        // four elements per thread
    grid = grid < 1 ? 1 : (grid > MSE_PARTS ? MSE_PARTS : grid);
    hipLaunchKernelGGL(mse_counted_kernel, dim3(grid), dim3(256), 0, s, truth, pred, mask, n, C, cnt, partial, ticket, loss, dpred, grad_scale);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_mse_small(const float* truth, const float* pred, const float* mask, int rows, int C, const float* cnt_in,
                     float* err_sum, float* cnt, float* loss, float* dpred, float grad_scale, hipStream_t s) {
    int CT = 1;
    while (CT < C && CT < 64) CT <<= 1;
    hipLaunchKernelGGL(mse_small_kernel, dim3(1), dim3(1024), (size_t)(2 * C + 2048) * sizeof(float), s, truth, pred, mask, rows, C, CT,
                       cnt_in, err_sum, cnt, loss, dpred, grad_scale);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_mse_finish(const float* truth, const float* pred, const float* mask, int rows, int C, const float* err_sum,
                      const float* cnt, float* loss, float* dpred, float grad_scale, hipStream_t s) {
    const size_t n = (size_t)rows * C;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 1024 ? 1024 : ((n + 255) / 256 ? (n + 255) / 256 : 1));
    hipLaunchKernelGGL(mse_finish_kernel, dim3(blocks), dim3(256), 0, s, truth, pred, mask, rows, C, err_sum, cnt, loss, dpred,
                       grad_scale);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// 16-byte accesses need 16-byte aligned buffers (8 for the bf16 twin): FlatTrainer's are; anything else goes scalar
static int adam_vec_ok(const float* p, const float* g, const float* m, const float* v, const void* twin) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                        reinterpret_cast<uintptr_t>(v);
    return (a & 15) == 0 && (reinterpret_cast<uintptr_t>(twin) & 7) == 0;
}

int launch_adam(float* param, const float* grad, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                float wd, int step, float max_norm, float* norm_scratch, hipStream_t s) {
    if (n == 0) return IMMTSF_OK;
    const int nparts = 1024;
    void* twin = const_cast<void*>(immtsf_twin_lookup(param, n));
    const int vec = adam_vec_ok(param, grad, m, v, twin);
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nparts), dim3(256), 0, s, grad, n, vec, norm_scratch, (long long*)nullptr,
                       (unsigned long long*)nullptr, (const int*)nullptr);
    IMMTSF_LAUNCH_CHECK();
    const float bc1 = 1.f - powf(b1, (float)step), bc2s = sqrtf(1.f - powf(b2, (float)step));
    const unsigned blocks = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, param, grad, m, v, n, lr, b1, b2, eps, wd, bc1, bc2s, max_norm,
                       norm_scratch, nparts, (const long long*)nullptr, reinterpret_cast<bf16_t*>(twin), vec, (float*)nullptr, (const int*)nullptr);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

// the two halves of the step as separate launches, for a SHARDED optimizer: every rank squares its own gradient shard,
// the 1024 partial sums are all-reduced (4 KB), and the update of the shard clips by the GLOBAL norm
int launch_adam_sqnorm(const float* grad, size_t n, float* norm_scratch, long long* step_dev, unsigned long long* drop_dev,
                       hipStream_t s) {
    const int vec = ((reinterpret_cast<uintptr_t>(grad) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(1024), dim3(256), 0, s, grad, n, vec, norm_scratch, step_dev, drop_dev, (const int*)nullptr);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int launch_adam_apply(float* param, const float* grad, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                      float wd, int step, const long long* step_dev, float max_norm, const float* norm_scratch, void* twin,
                      hipStream_t s) {
    if (n == 0) return IMMTSF_OK;
    if (!twin) twin = const_cast<void*>(immtsf_twin_lookup(param, n));
    const int vec = adam_vec_ok(param, grad, m, v, twin);
    const float bc1 = step_dev ? 1.f : 1.f - powf(b1, (float)step), bc2s = step_dev ? 1.f : sqrtf(1.f - powf(b2, (float)step));
    const unsigned blocks = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, param, grad, m, v, n, lr, b1, b2, eps, wd, bc1, bc2s, max_norm,
                       norm_scratch, 1024, step_dev, reinterpret_cast<bf16_t*>(twin), vec, (float*)nullptr, (const int*)nullptr);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_adam_dev(float* param, const float* grad, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                    float wd, long long* step_dev, float max_norm, float* norm_scratch, unsigned long long* drop_dev,
                    hipStream_t s, int zero_grad, const int* skip) {
    if (n == 0) return IMMTSF_OK;
    const int nparts = 1024;
    void* twin = const_cast<void*>(immtsf_twin_lookup(param, n));
    const int vec = adam_vec_ok(param, grad, m, v, twin);
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nparts), dim3(256), 0, s, grad, n, vec, norm_scratch, step_dev, drop_dev, skip);
    IMMTSF_LAUNCH_CHECK();
    const unsigned blocks = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, param, grad, m, v, n, lr, b1, b2, eps, wd, 1.f, 1.f, max_norm,
                       norm_scratch, nparts, (const long long*)step_dev, reinterpret_cast<bf16_t*>(twin), vec,
                       zero_grad ? const_cast<float*>(grad) : nullptr, skip);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}

int launch_adam_prepare(const float* grad, const void* grad_h, size_t n, float* norm_scratch, long long* step_dev, unsigned long long* drop_dev,
                        int* pending, const int* err, const void* guard_h, const float* guard_f, int* skip_out, hipStream_t s) {
    const uintptr_t a = grad_h ? reinterpret_cast<uintptr_t>(grad_h) : reinterpret_cast<uintptr_t>(grad);
    const int vec = (a & 15) == 0;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1024), dim3(256), 0, s, grad, static_cast<const bf16_t*>(grad_h), n, vec, norm_scratch, step_dev,
                       drop_dev, pending, err, static_cast<const bf16_t*>(guard_h), guard_f, skip_out);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int launch_adam_range(float* param, float* grad, const void* grad_h, float* m, float* v, size_t n, size_t lo, size_t hi, float lr, float b1,
                      float b2, float eps, float wd, const long long* step_dev, float max_norm, const float* norm_scratch, int zero_grad,
                      const int* skip, hipStream_t s) {
    if (hi <= lo) return IMMTSF_OK;
    const bf16_t* twin0 = static_cast<const bf16_t*>(immtsf_twin_lookup(param, n));
    bf16_t* twin = twin0 ? const_cast<bf16_t*>(twin0) + lo : nullptr;
    const bf16_t* gh = grad_h ? static_cast<const bf16_t*>(grad_h) + lo : nullptr;
    const size_t k = hi - lo;
    const int vec = adam_vec_ok(param + lo, grad + lo, m + lo, v + lo, twin) && (reinterpret_cast<uintptr_t>(gh) & 7) == 0;
    const unsigned blocks = (unsigned)((k / 4 + 255) / 256 > 2048 ? 2048 : (k / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, param + lo, grad + lo, m + lo, v + lo, k, lr, b1, b2, eps, wd, 1.f, 1.f,
                       max_norm, norm_scratch, 1024, step_dev, twin, vec, zero_grad ? grad + lo : nullptr, skip, gh);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int launch_guard_pack(const int* err, void* slot, int is_bf16, hipStream_t s) {
    hipLaunchKernelGGL(guard_pack_kernel, dim3(1), dim3(1), 0, s, err, slot, is_bf16);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
