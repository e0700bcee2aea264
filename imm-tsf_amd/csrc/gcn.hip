// tPatchGNN's adaptive-graph stage for one (window, patch) per workgroup: node-vector gating, adaptive adjacency
// softmax(relu(NV1 NV2)), `order` hops of graph convolution and the 1x1-conv mixing layer
// (models/tPatchGNN.py:205-238 with gcn/nconv/linear at :29-84).  Every operand of a (b, m) cell -- N variables x D
// channels, N x N adjacency, the ~3k parameters -- fits in LDS, so the ~35 eager launches of the forward collapse
// into one kernel and the ~55 of the backward into one more (which recomputes the forward in LDS instead of reading
// saved activations back from HBM).  Latency-bound work: the win is launches, not bytes.
#include "../../include/immtsf.h"
#include "common.hpp"

namespace {

struct Dims {
    int B, N, M, D, nd, order;
};

// LDS map (float offsets); the same struct is built on host (for the size) and device
struct Lds {
    int X, Xk, Z, dfeat, L1, L2, dL1, dL2, NV1, NV2, dNV1, dNV2, A, S, dA, Wm, vec, total, ldw, F;
    int fwd_total, g_mb, g_mw, g_nv1, g_nv2, g_g1w, g_g2w, g_gb, g_l1w, g_l2w, g_l1b, g_l2b;
    int p_l1w, p_l2w, p_g1w, p_g2w, p_nv1, p_nv2;      // backward: the parameters its phases read, staged once per workgroup
    int saved;           // floats a cell hands from the forward to the backward: X | X_k | Z | L1 | L2 | NV1 | NV2 | A | S | t, g (4 N)
    __host__ __device__ explicit Lds(const Dims& d) {
        const int ND = d.N * d.D, Nn = d.N * d.nd, NN = d.N * d.N;
        F = (d.order + 1) * d.D;            // concatenated feature width
        ldw = F + 1;                        // padded row of the mixing weight
        int o = 0;
        X = o;      o += ND;                // X = feat[:, 0:D]; Xk (k = 1..order) follow contiguously as N x D slabs
        Xk = o;     o += d.order * ND;
        Z = o;      o += ND;                // pre-activation of the mixing layer, later dZ
        dfeat = o;  o += (d.order + 1) * ND;
        L1 = o;     o += Nn;
        L2 = o;     o += Nn;
        dL1 = o;    o += Nn;
        dL2 = o;    o += Nn;
        NV1 = o;    o += Nn;                // (N, nd)
        NV2 = o;    o += Nn;                // (nd, N)
        dNV1 = o;   o += Nn;
        dNV2 = o;   o += Nn;
        A = o;      o += NN;
        S = o;      o += NN;
        dA = o;     o += NN;
        Wm = o;     o += d.D * ldw;
        vec = o;    o += 8 * d.N;           // t1, t2, g1, g2, da1, da2, (2 spare)
        fwd_total = o;
        // backward only: the workgroup's running parameter gradients (it walks several cells; one atomic per element at the end)
        g_mb = o;   o += d.D;
        g_mw = o;   o += d.D * F;
        g_nv1 = o;  o += Nn;
        g_nv2 = o;  o += Nn;
        g_g1w = o;  o += d.D + d.nd;
        g_g2w = o;  o += d.D + d.nd;
        g_gb = o;   o += 2;                 // gate1_b, gate2_b
        g_l1w = o;  o += d.nd * d.D;
        g_l2w = o;  o += d.nd * d.D;
        g_l1b = o;  o += d.nd;
        g_l2b = o;  o += d.nd;
        p_l1w = o;  o += d.nd * d.D;
        p_l2w = o;  o += d.nd * d.D;
        p_g1w = o;  o += d.D + d.nd;
        p_g2w = o;  o += d.D + d.nd;
        p_nv1 = o;  o += Nn;
        p_nv2 = o;  o += Nn;
        total = o;
        saved = (d.order + 2) * ND + 4 * Nn + 2 * NN + 4 * d.N;
    }
};

struct Params {
    const float *nv1, *nv2, *g1w, *g1b, *g2w, *g2b, *l1w, *l1b, *l2w, *l2b, *mw, *mb;
};
struct Grads {
    float *nv1, *nv2, *g1w, *g1b, *g2w, *g2b, *l1w, *l1b, *l2w, *l2b, *mw, *mb;
};

// feat[v, c]: c < D -> X, else the hop slabs
__device__ __forceinline__ float feat_at(const float* sm, const Lds& l, const Dims& d, int v, int c) {
    const int k = c / d.D, f = c - k * d.D;
    return sm[l.X + k * d.N * d.D + v * d.D + f];      // X and Xk are contiguous slabs
}

// the forward of one (b, m) cell entirely in LDS; leaves X, Xk, Z (pre-activation), L1, L2, NV1, NV2, A, S, t/g in sm
__device__ void cell_forward(float* sm, const Lds& l, const Dims& d, const Params& p, const float* __restrict__ x, int b, int m) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int N = d.N, D = d.D, nd = d.nd;
    for (int i = tid; i < N * D; i += nt) {
        const int n = i / D, f = i - n * D;
        sm[l.X + i] = x[(((size_t)b * N + n) * d.M + m) * D + f];
    }
    for (int i = tid; i < D * l.F; i += nt) {
        const int o = i / l.F, c = i - o * l.F;
        sm[l.Wm + o * l.ldw + c] = p.mw[i];
    }
    __syncthreads();
    // per node: two nd-wide linears and two gate logits
    const int per = 2 * nd + 2;
    for (int i = tid; i < N * per; i += nt) {
        const int n = i / per, j = i - n * per;
        const float* xr = sm + l.X + n * D;
        float acc;
        if (j < 2 * nd) {
            const int k = j < nd ? j : j - nd;
            const float* w = (j < nd ? p.l1w : p.l2w) + (size_t)k * D;
            acc = (j < nd ? p.l1b : p.l2b)[k];
#pragma unroll 8
            for (int f = 0; f < D; ++f) acc = fmaf(w[f], xr[f], acc);
            sm[(j < nd ? l.L1 : l.L2) + n * nd + k] = acc;
        } else {
            const bool one = j == 2 * nd;
            const float* w = one ? p.g1w : p.g2w;
            acc = (one ? p.g1b : p.g2b)[0];
#pragma unroll 8
            for (int f = 0; f < D; ++f) acc = fmaf(w[f], xr[f], acc);
#pragma unroll 8
            for (int k = 0; k < nd; ++k) acc = fmaf(w[D + k], one ? p.nv1[n * nd + k] : p.nv2[k * N + n], acc);
            const float t = tanhf(acc);
            sm[l.vec + (one ? 0 : 1) * N + n] = t;
            sm[l.vec + (one ? 2 : 3) * N + n] = fmaxf(t, 0.f);
        }
    }
    __syncthreads();
    for (int i = tid; i < N * nd; i += nt) {
        const int n = i / nd, k = i - n * nd;
        sm[l.NV1 + n * nd + k] = fmaf(sm[l.vec + 2 * N + n], sm[l.L1 + i], p.nv1[n * nd + k]);
        sm[l.NV2 + k * N + n] = fmaf(sm[l.vec + 3 * N + n], sm[l.L2 + i], p.nv2[k * N + n]);
    }
    __syncthreads();
    for (int i = tid; i < N * N; i += nt) {
        const int r = i / N, c = i - r * N;
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < nd; ++k) acc = fmaf(sm[l.NV1 + r * nd + k], sm[l.NV2 + k * N + c], acc);
        sm[l.S + i] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int r = tid; r < N; r += nt) {
        float mx = -INFINITY;
        for (int c = 0; c < N; ++c) mx = fmaxf(mx, sm[l.S + r * N + c]);
        float den = 0.f;
        for (int c = 0; c < N; ++c) {
            const float e = expf(sm[l.S + r * N + c] - mx);
            sm[l.A + r * N + c] = e;
            den += e;
        }
        const float inv = 1.f / den;
        for (int c = 0; c < N; ++c) sm[l.A + r * N + c] *= inv;
    }
    __syncthreads();
    // hops: X_k[v, f] = sum_n A[n, v] X_{k-1}[n, f]
    for (int k = 1; k <= d.order; ++k) {
        const float* src = sm + l.X + (k - 1) * N * D;
        float* dst = sm + l.X + k * N * D;
        for (int i = tid; i < N * D; i += nt) {
            const int v = i / D, f = i - v * D;
            float acc = 0.f;
#pragma unroll 8
            for (int n = 0; n < N; ++n) acc = fmaf(sm[l.A + n * N + v], src[n * D + f], acc);
            dst[i] = acc;
        }
        __syncthreads();
    }
    // mixing layer (1x1 conv over the concatenated features)
    for (int i = tid; i < N * D; i += nt) {
        const int v = i / D, o = i - v * D;
        float acc = p.mb[o];
        const float* w = sm + l.Wm + o * l.ldw;
#pragma unroll 8
        for (int c = 0; c < l.F; ++c) acc = fmaf(w[c], feat_at(sm, l, d, v, c), acc);
        sm[l.Z + i] = acc;
    }
    __syncthreads();
}

// what cell_forward leaves behind that the backward reads, as one contiguous record per cell (`store`: LDS -> memory, else back).  The
// backward used to recompute the cell: 11.7 of its 33 us on the device clock -- every phase of the forward is a barrier and a chain of
// dependent LDS reads, 5 KB per cell come back in one round trip
template <bool STORE>
__device__ __forceinline__ void cell_record(float* sm, const Lds& l, const Dims& d, float* rec) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int ND = d.N * d.D, Nn = d.N * d.nd, NN = d.N * d.N;
    const int off[7] = {l.X, l.L1, l.NV1, l.A, l.S, l.vec, 0}, len[6] = {(d.order + 2) * ND, 2 * Nn, 2 * Nn, NN, NN, 4 * d.N};
    int at = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {           // (X, X_k, Z are contiguous in LDS; so are L1 | L2 and NV1 | NV2)
        for (int i = tid; i < len[q]; i += nt) {
            if (STORE) rec[at + i] = sm[off[q] + i];
            else sm[off[q] + i] = rec[at + i];
        }
        at += len[q];
    }
}

template <bool SAVE>
__global__ __launch_bounds__(256) void gcn_fwd_kernel(Dims d, Params p, const float* __restrict__ x, float* __restrict__ out, float* __restrict__ saved) {
    extern __shared__ float sm[];
    const Lds l(d);
    const int b = blockIdx.x / d.M, m = blockIdx.x - b * d.M;
    cell_forward(sm, l, d, p, x, b, m);
    for (int i = threadIdx.x; i < d.N * d.D; i += blockDim.x) {
        const int v = i / d.D, o = i - v * d.D;
        out[(((size_t)b * d.N + v) * d.M + m) * d.D + o] = fmaxf(sm[l.Z + i], 0.f);
    }
    if (SAVE) cell_record<true>(sm, l, d, saved + (size_t)blockIdx.x * l.saved);
}

template <bool SAVED>
__global__ __launch_bounds__(256) void gcn_bwd_kernel(Dims d, Params p, const float* __restrict__ x,
                                                       const float* __restrict__ dout, float* __restrict__ dx, Grads g, const float* __restrict__ saved) {
    extern __shared__ float sm[];
    const Lds l(d);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int N = d.N, D = d.D, nd = d.nd;
    for (int i = l.g_mb + tid; i < l.p_l1w; i += nt) sm[i] = 0.f;
    // the parameters the phases below read, once per workgroup (from global memory inside the loops they were chains of dependent L2
    // round trips: the gate-weight and dx phases 2.4 us each of a cell's 33 on the device clock)
    for (int i = tid; i < nd * D; i += nt) { sm[l.p_l1w + i] = p.l1w[i]; sm[l.p_l2w + i] = p.l2w[i]; }
    for (int i = tid; i < D + nd; i += nt) { sm[l.p_g1w + i] = p.g1w[i]; sm[l.p_g2w + i] = p.g2w[i]; }
    for (int i = tid; i < N * nd; i += nt) { sm[l.p_nv1 + i] = p.nv1[i]; sm[l.p_nv2 + i] = p.nv2[i]; }
    if (SAVED)          // the mixing weight once per workgroup (cell_forward stages it per cell)
        for (int i = tid; i < D * l.F; i += nt) {
            const int o = i / l.F, c = i - o * l.F;
            sm[l.Wm + o * l.ldw + c] = p.mw[i];
        }
    // a workgroup walks cells blockIdx.x, + gridDim.x, ...: every parameter-gradient element below is owned by one thread
    // (same loop shape in every cell), accumulated in LDS and added to global memory once per workgroup -- with one workgroup
    // per cell the ~3.5 k atomics of each of B * M cells queue up per address and were the whole run time of this kernel
    for (int cell = blockIdx.x; cell < d.B * d.M; cell += gridDim.x) {
    const int b = cell / d.M, m = cell - b * d.M;
    if (SAVED) {
        cell_record<false>(sm, l, d, const_cast<float*>(saved) + (size_t)cell * l.saved);
        __syncthreads();
    } else {
        cell_forward(sm, l, d, p, x, b, m);
    }
    // dZ = dout * relu'(Z)
    for (int i = tid; i < N * D; i += nt) {
        const int v = i / D, o = i - v * D;
        const float go = dout[(((size_t)b * N + v) * d.M + m) * D + o];
        sm[l.Z + i] = sm[l.Z + i] > 0.f ? go : 0.f;
    }
    for (int i = tid; i < N * N; i += nt) sm[l.dA + i] = 0.f;
    __syncthreads();
    // mixing-layer parameter gradients
    for (int o = tid; o < D; o += nt) {
        float acc = 0.f;
#pragma unroll 8
        for (int v = 0; v < N; ++v) acc += sm[l.Z + v * D + o];
        sm[l.g_mb + o] += acc;
    }
    for (int i = tid; i < D * l.F; i += nt) {
        const int o = i / l.F, c = i - o * l.F, k = c / D, f = c - k * D;
        const float* fx = sm + l.X + k * N * D + f;          // feat[v, c] = X_k[v, f]
        float acc = 0.f;
#pragma unroll 8
        for (int v = 0; v < N; ++v) acc = fmaf(sm[l.Z + v * D + o], fx[v * D], acc);
        sm[l.g_mw + i] += acc;
    }
    // dfeat[k][v, f] = sum_o Wm[o, k*D + f] dZ[v, o]     (stored as order+1 slabs of N x D, like X/Xk)
    for (int i = tid; i < (d.order + 1) * N * D; i += nt) {
        const int k = i / (N * D), r = i - k * N * D, v = r / D, f = r - v * D;
        float acc = 0.f;
#pragma unroll 8
        for (int o = 0; o < D; ++o) acc = fmaf(sm[l.Wm + o * l.ldw + k * D + f], sm[l.Z + v * D + o], acc);
        sm[l.dfeat + i] = acc;
    }
    __syncthreads();
    // hops backward: dA[n, v] += sum_f X_{k-1}[n, f] dX_k[v, f];  dX_{k-1}[n, f] += sum_v A[n, v] dX_k[v, f]
    for (int k = d.order; k >= 1; --k) {
        const float* xs = sm + l.X + (k - 1) * N * D;
        const float* dk = sm + l.dfeat + k * N * D;
        float* dprev = sm + l.dfeat + (k - 1) * N * D;
        for (int i = tid; i < N * N; i += nt) {
            const int n = i / N, v = i - n * N;
            float acc = 0.f;
#pragma unroll 8
            for (int f = 0; f < D; ++f) acc = fmaf(xs[n * D + f], dk[v * D + f], acc);
            sm[l.dA + i] += acc;
        }
        for (int i = tid; i < N * D; i += nt) {
            const int n = i / D, f = i - n * D;
            float acc = 0.f;
#pragma unroll 8
            for (int v = 0; v < N; ++v) acc = fmaf(sm[l.A + n * N + v], dk[v * D + f], acc);
            dprev[i] += acc;
        }
        __syncthreads();
    }
    // softmax + relu backward (in place: dA -> dS)
    for (int r = tid; r < N; r += nt) {
        float dot = 0.f;
#pragma unroll 8
        for (int c = 0; c < N; ++c) dot = fmaf(sm[l.A + r * N + c], sm[l.dA + r * N + c], dot);
        for (int c = 0; c < N; ++c) {
            const float v = sm[l.A + r * N + c] * (sm[l.dA + r * N + c] - dot);
            sm[l.dA + r * N + c] = sm[l.S + r * N + c] > 0.f ? v : 0.f;
        }
    }
    __syncthreads();
    for (int i = tid; i < N * nd; i += nt) {
        const int n = i / nd, k = i - n * nd;
        float a1 = 0.f, a2 = 0.f;
#pragma unroll 8
        for (int c = 0; c < N; ++c) {
            a1 = fmaf(sm[l.dA + n * N + c], sm[l.NV2 + k * N + c], a1);      // dNV1[n,k] = sum_j dS[n,j] NV2[k,j]
            a2 = fmaf(sm[l.NV1 + c * nd + k], sm[l.dA + c * N + n], a2);      // dNV2[k,n] = sum_i NV1[i,k] dS[i,n]
        }
        sm[l.dNV1 + n * nd + k] = a1;
        sm[l.dNV2 + n * nd + k] = a2;           // stored (n, k) like p2
        sm[l.dL1 + i] = sm[l.vec + 2 * N + n] * a1;
        sm[l.dL2 + i] = sm[l.vec + 3 * N + n] * a2;
    }
    __syncthreads();
    // gate logits: da = dg * relu'(t) * (1 - t^2),  dg[n] = sum_k dNV[n,k] L[n,k]
    for (int i = tid; i < 2 * N; i += nt) {
        const int which = i / N, n = i - which * N;
        const float* dnv = sm + (which ? l.dNV2 : l.dNV1) + n * nd;
        const float* L = sm + (which ? l.L2 : l.L1) + n * nd;
        float dg = 0.f;
#pragma unroll 8
        for (int k = 0; k < nd; ++k) dg = fmaf(dnv[k], L[k], dg);
        const float t = sm[l.vec + which * N + n];
        sm[l.vec + (4 + which) * N + n] = t > 0.f ? dg * (1.f - t * t) : 0.f;
    }
    __syncthreads();
    const float* da1 = sm + l.vec + 4 * N;
    const float* da2 = sm + l.vec + 5 * N;
    // node vectors
    for (int i = tid; i < N * nd; i += nt) {
        const int n = i / nd, k = i - n * nd;
        sm[l.g_nv1 + n * nd + k] += sm[l.dNV1 + i] + da1[n] * sm[l.p_g1w + D + k];
        sm[l.g_nv2 + k * N + n] += sm[l.dNV2 + i] + da2[n] * sm[l.p_g2w + D + k];
    }
    // gate weights / biases
    for (int i = tid; i < 2 * (D + nd + 1); i += nt) {
        const int which = i / (D + nd + 1), j = i - which * (D + nd + 1);
        const float* da = which ? da2 : da1;
        float acc = 0.f;
        if (j < D) for (int n = 0; n < N; ++n) acc = fmaf(da[n], sm[l.X + n * D + j], acc);
        else if (j < D + nd) for (int n = 0; n < N; ++n) acc = fmaf(da[n], which ? sm[l.p_nv2 + (j - D) * N + n] : sm[l.p_nv1 + n * nd + (j - D)], acc);
        else for (int n = 0; n < N; ++n) acc += da[n];
        if (j < D + nd) sm[(which ? l.g_g2w : l.g_g1w) + j] += acc;
        else sm[l.g_gb + which] += acc;
    }
    // nd-wide linears
    for (int i = tid; i < 2 * nd * (D + 1); i += nt) {
        const int which = i / (nd * (D + 1)), r = i - which * nd * (D + 1), k = r / (D + 1), f = r - k * (D + 1);
        const float* dL = sm + (which ? l.dL2 : l.dL1);
        float acc = 0.f;
        if (f < D) {
#pragma unroll 8
            for (int n = 0; n < N; ++n) acc = fmaf(dL[n * nd + k], sm[l.X + n * D + f], acc);
            sm[(which ? l.g_l2w : l.g_l1w) + k * D + f] += acc;
        } else {
#pragma unroll 8
            for (int n = 0; n < N; ++n) acc += dL[n * nd + k];
            sm[(which ? l.g_l2b : l.g_l1b) + k] += acc;
        }
    }
    // dx
    for (int i = tid; i < N * D; i += nt) {
        const int n = i / D, f = i - n * D;
        float acc = sm[l.dfeat + i];
        acc = fmaf(da1[n], sm[l.p_g1w + f], acc);
        acc = fmaf(da2[n], sm[l.p_g2w + f], acc);
#pragma unroll 2
        for (int k = 0; k < nd; ++k) {
            acc = fmaf(sm[l.dL1 + n * nd + k], sm[l.p_l1w + k * D + f], acc);
            acc = fmaf(sm[l.dL2 + n * nd + k], sm[l.p_l2w + k * D + f], acc);
        }
        dx[(((size_t)b * N + n) * d.M + m) * D + f] = acc;
    }
    __syncthreads();        // the next cell's forward overwrites what the loops above read
    }   // cells
    auto flush = [&](float* dst, int off, int n) __attribute__((always_inline)) {
        for (int i = tid; i < n; i += nt) atomicAdd(dst + i, sm[off + i]);
    };
    flush(g.mb, l.g_mb, D);
    flush(g.mw, l.g_mw, D * l.F);
    flush(g.nv1, l.g_nv1, N * nd);
    flush(g.nv2, l.g_nv2, N * nd);
    flush(g.g1w, l.g_g1w, D + nd);
    flush(g.g2w, l.g_g2w, D + nd);
    flush(g.g1b, l.g_gb, 1);
    flush(g.g2b, l.g_gb + 1, 1);
    flush(g.l1w, l.g_l1w, nd * D);
    flush(g.l2w, l.g_l2w, nd * D);
    flush(g.l1b, l.g_l1b, nd);
    flush(g.l2b, l.g_l2b, nd);
}

constexpr size_t kMaxLds = 160 * 1024 - 256;

int check_dims(const Dims& d, size_t* bytes) {
    if (d.B <= 0 || d.N <= 0 || d.M <= 0 || d.D <= 0 || d.nd <= 0 || d.order < 1) return IMMTSF_EINVAL;
    const Lds l(d);
    *bytes = (size_t)l.total * sizeof(float);
    return *bytes <= kMaxLds ? IMMTSF_OK : IMMTSF_EUNSUPPORTED;
}

Params to_params(const immtsf_gcn_params* p) {
    return Params{p->nodevec1, p->nodevec2, p->gate1_w, p->gate1_b, p->gate2_w, p->gate2_b,
                  p->lin1_w, p->lin1_b, p->lin2_w, p->lin2_b, p->mlp_w, p->mlp_b};
}
bool all_set(const immtsf_gcn_params* p) {
    return p && p->nodevec1 && p->nodevec2 && p->gate1_w && p->gate1_b && p->gate2_w && p->gate2_b && p->lin1_w &&
           p->lin1_b && p->lin2_w && p->lin2_b && p->mlp_w && p->mlp_b;
}

}  // namespace

extern "C" {

size_t immtsf_tpatchgnn_gcn_lds_bytes(int32_t N, int32_t D, int32_t nd, int32_t order) {
    size_t bytes = 0;
    const Dims d{1, N, 1, D, nd, order};
    return check_dims(d, &bytes) == IMMTSF_OK ? bytes : 0;
}

size_t immtsf_tpatchgnn_gcn_saved_floats(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order) {
    size_t bytes = 0;
    const Dims d{B, N, M, D, nd, order};
    return check_dims(d, &bytes) == IMMTSF_OK ? (size_t)B * M * Lds(d).saved : 0;
}

static int gcn_forward(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order, const float* x, const immtsf_gcn_params* p,
                       float* out, float* saved, immtsf_stream_t stream) {
    const Dims d{B, N, M, D, nd, order};
    size_t bytes = 0;
    if (!x || !out || !all_set(p)) return IMMTSF_EINVAL;
    if (int rc = check_dims(d, &bytes)) return rc;
    bytes = (size_t)Lds(d).fwd_total * sizeof(float);       // (the backward's gradient accumulators are not needed)
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(saved ? reinterpret_cast<const void*>(gcn_fwd_kernel<true>) : reinterpret_cast<const void*>(gcn_fwd_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    if (saved) hipLaunchKernelGGL(gcn_fwd_kernel<true>, dim3(B * M), dim3(256), bytes, static_cast<hipStream_t>(stream), d, to_params(p), x, out, saved);
    else hipLaunchKernelGGL(gcn_fwd_kernel<false>, dim3(B * M), dim3(256), bytes, static_cast<hipStream_t>(stream), d, to_params(p), x, out, saved);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int immtsf_tpatchgnn_gcn_forward(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order, const float* x,
                                 const immtsf_gcn_params* p, float* out, immtsf_stream_t stream) {
    return gcn_forward(B, N, M, D, nd, order, x, p, out, nullptr, stream);
}
int immtsf_tpatchgnn_gcn_forward_saved(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order, const float* x,
                                       const immtsf_gcn_params* p, float* out, float* saved, immtsf_stream_t stream) {
    if (!saved) return IMMTSF_EINVAL;
    return gcn_forward(B, N, M, D, nd, order, x, p, out, saved, stream);
}

static int gcn_backward(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order, const float* x, const immtsf_gcn_params* p,
                        const float* dout, float* dx, const immtsf_gcn_params* grads, const float* saved, immtsf_stream_t stream) {
    const Dims d{B, N, M, D, nd, order};
    size_t bytes = 0;
    if ((!x && !saved) || !dout || !dx || !all_set(p) || !all_set(grads)) return IMMTSF_EINVAL;
    if (int rc = check_dims(d, &bytes)) return rc;
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(saved ? reinterpret_cast<const void*>(gcn_bwd_kernel<true>) : reinterpret_cast<const void*>(gcn_bwd_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    const Grads g{grads->nodevec1, grads->nodevec2, grads->gate1_w, grads->gate1_b, grads->gate2_w, grads->gate2_b,
                  grads->lin1_w, grads->lin1_b, grads->lin2_w, grads->lin2_b, grads->mlp_w, grads->mlp_b};
    int per_cu = (int)((160 * 1024) / bytes);      // resident workgroups per CU by LDS, at most 4: the grid is the atomics' fan-in
    per_cu = per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu;
    const int cells = B * M, grid = cells < 256 * per_cu ? cells : 256 * per_cu;
    if (saved) hipLaunchKernelGGL(gcn_bwd_kernel<true>, dim3(grid), dim3(256), bytes, static_cast<hipStream_t>(stream), d, to_params(p), x, dout, dx, g, saved);
    else hipLaunchKernelGGL(gcn_bwd_kernel<false>, dim3(grid), dim3(256), bytes, static_cast<hipStream_t>(stream), d, to_params(p), x, dout, dx, g, saved);
    IMMTSF_LAUNCH_CHECK();
    return IMMTSF_OK;
}
int immtsf_tpatchgnn_gcn_backward(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order, const float* x,
                                  const immtsf_gcn_params* p, const float* dout, float* dx, const immtsf_gcn_params* grads,
                                  immtsf_stream_t stream) {
    if (!x) return IMMTSF_EINVAL;
    return gcn_backward(B, N, M, D, nd, order, x, p, dout, dx, grads, nullptr, stream);
}
int immtsf_tpatchgnn_gcn_backward_saved(int32_t B, int32_t N, int32_t M, int32_t D, int32_t nd, int32_t order, const float* saved,
                                        const immtsf_gcn_params* p, const float* dout, float* dx, const immtsf_gcn_params* grads,
                                        immtsf_stream_t stream) {
    if (!saved) return IMMTSF_EINVAL;
    return gcn_backward(B, N, M, D, nd, order, nullptr, p, dout, dx, grads, saved, stream);
}

}  // extern "C"
