// MMF_XAttn_Add (fusions/MMF_XAttn_Add.py:56-103) as a sequence of MFMA GEMMs + row kernels.
// Q = W_q Y, K = W_k E, V = W_v E -> nn.MultiheadAttention (three more in-projections, T x T attention per
// window and head with all-or-nothing key padding from M_txt, attention-weight dropout, out_proj) -> zero the
// no-text windows -> residual_head -> LN(C) -> dropout -> zero no-text -> (Y + kappa*delta)/(1+kappa).
// QK^T and A*V run as batched MFMA GEMMs over (window, head); scores live in HBM ((B,H,T,T) fp32, L2-resident).
#include "../../include/immtsf.h"
#include "attn.hpp"
#include "gemm.hpp"
#include "rowops.hpp"
#include "tail.hpp"
#include "block_util.hpp"
#include <math.h>

namespace {

// ---- Folded form.  Between proj_{q,k,v} (no bias) and the MHA in-projections there is no nonlinearity, and neither is
// there between out_proj and residual_head (only the zeroing of no-text windows, which commutes with a row-wise linear
// map up to the bias), so per step the block forms the PRODUCT weights once
//     W_Kf = W_in,k W_k   W_Vf = W_in,v W_v   (d x d each, stacked as W_KVf (2d, d))      W_Qf = W_in,q W_q  (d x C)
//     W_HO = W_res W_out  (C x d)             b_HO = W_res b_out + b_res
// and runs ONE (B*T) x 2d x d projection of E_txt (k | v side by side), a (B*T) x d x C projection of Y_ts, the attention,
// and the C-wide head straight on the attention output.  Backward: the data gradients flow through the same products
// and the gradients of the ORIGINAL parameters follow by the chain rule from the product weights' gradients (d x d x d
// GEMMs instead of (B*T) x d x d ones): dW_in,k = dW_Kf W_k^T, dW_k = W_in,k^T dW_Kf, ...  Same function, same
// state_dict, results equal to the unfolded chain up to fp32 reassociation (bf16 mode: one operand rounding of the
// product weight instead of one of the intermediate activation).  Six of the block's eight (B*T) x d x d forward GEMMs
// and eight of its twelve backward ones disappear.
//
// The block is split into a key/value half (depends only on E_txt) and a query half (everything else).  The halves are
// separate C entry points so the host can run the key/value half on the text stream beside the backbone, and start the
// backbone's backward as soon as the query half has produced dY_ts.  The monolithic entry points call both.
inline bool xadd_hf(const immtsf_fusion_cfg* c) { return c->precision == 1 && c->d >= 16 && (c->d % 16) == 0; }

struct KVWs {
    Mat E, WKVf;          // bf16 image of E_txt (hf) ; the stacked product weight (2d, d)
    void *w_k, *w_v, *w_inkv;
    size_t bytes;
};
KVWs carve_kv(const immtsf_fusion_cfg* c, void* base) {
    const size_t BT = (size_t)c->B * c->T, d = c->d;
    const bool hf = xadd_hf(c);
    Carver k(base);
    KVWs w;
    w.E = k.take_mat(BT * d, false, hf);
    w.WKVf = k.take_mat(2 * d * d, !hf, hf);
    w.w_k = w.w_v = w.w_inkv = nullptr;
    if (hf) {
        w.w_k = k.take<unsigned short>(d * d);
        w.w_v = k.take<unsigned short>(d * d);
        w.w_inkv = k.take<unsigned short>(2 * d * d);
    }
    w.bytes = k.bytes();
    return w;
}
struct KVScratch {
    Mat dKV, dWKVf;
    void* sk;         // split-K workspace of the dW_KVf product (empty below 8192 rows)
    size_t skb;
    size_t bytes;
};
KVScratch carve_kv_scratch(const immtsf_fusion_cfg* c, void* base) {
    const size_t BT = (size_t)c->B * c->T, d = c->d;
    const bool hf = xadd_hf(c);
    Carver k(base);
    KVScratch s;
    s.dKV = k.take_mat(BT * 2 * d, false, hf);
    s.dWKVf = k.take_mat(2 * d * d, !hf, hf);
    s.skb = hf ? immtsf_gemm3_tn_ws_bytes((int)(2 * d), (int)d, (int)BT) : 0;
    s.sk = s.skb ? k.take<unsigned char>(s.skb) : nullptr;
    s.bytes = k.bytes();
    return s;
}
struct KVW { Mat k, v, ink, inv; };
int kv_weights(const immtsf_fusion_cfg* c, const immtsf_xadd_params* p, const KVWs& w, hipStream_t s, KVW* o) {
    const bool hf = xadd_hf(c);
    const size_t d = c->d;
    CHECK(weight_mat(hf, p->proj_k_w, d * d, w.w_k, s, &o->k));
    CHECK(weight_mat(hf, p->proj_v_w, d * d, w.w_v, s, &o->v));
    Mat inkv;
    CHECK(weight_mat(hf, p->attn_in_w + d * d, 2 * d * d, w.w_inkv, s, &inkv));
    o->ink = inkv;
    o->inv = mat_off(inkv, d * d);
    return 0;
}

// the query half's per-step product weights: W_Qf (d, C) | W_HO (C, d) | b_HO (C) | t_HO = W_res b_out (C).  They depend on
// parameters only, so a caller can form them ahead of time (immtsf_mmf_xattn_q_fold, e.g. beside the key/value half
// while the backbone still runs) and hand them to q_forward / q_backward; otherwise they live in the forward workspace.
struct QFold {
    float *WQf, *WHO, *bHO, *tHO;
};
inline size_t qfold_floats(const immtsf_fusion_cfg* c) { return (size_t)2 * c->d * c->C + 2 * (size_t)c->C + 16; }
inline QFold qfold_at(const immtsf_fusion_cfg* c, float* base) {
    QFold f;
    const size_t d = c->d, C = c->C;
    f.WQf = base;
    f.WHO = base + d * C;
    f.bHO = f.WHO + C * d;
    f.tHO = f.bHO + ((C + 3) / 4) * 4;
    return f;
}
struct QWs {
    float *fold, *Qi, *Pm, *Am, *O, *delta, *xhatC, *rstdC;
    size_t bytes;
};
QWs carve_q(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, T = c->T, d = c->d, C = c->C, BT = B * T, S = B * c->H * T * T;
    const bool dropping = c->training && c->p_drop > 0.f;
    Carver k(base);
    QWs w;
    w.fold = k.take<float>(qfold_floats(c));
    w.Qi = k.take<float>(BT * d);
    w.Pm = k.take<float>(S);
    w.Am = dropping ? k.take<float>(S) : w.Pm;
    w.O = k.take<float>(BT * d);
    w.delta = k.take<float>(BT * C);
    w.xhatC = k.take<float>(BT * C);
    w.rstdC = k.take<float>(BT);
    w.bytes = k.bytes();
    return w;
}
struct QScratch {
    float *dn, *ddelta, *dO, *dA, *dQi, *dWQf, *dWHO, *slive, *red;
    size_t bytes;
};
QScratch carve_q_scratch(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, T = c->T, d = c->d, C = c->C, BT = B * T, S = B * c->H * T * T;
    Carver k(base);
    QScratch s;
    s.dn = k.take<float>(BT * C);
    s.ddelta = k.take<float>(BT * C);
    s.dO = k.take<float>(BT * d);
    s.dA = k.take<float>(S);
    s.dQi = k.take<float>(BT * d);
    s.dWQf = k.take<float>(d * C);
    s.dWHO = k.take<float>(C * d);
    s.slive = k.take<float>(C);
    s.red = k.take<float>(64 * (d + C + 8) + colsum_scratch_floats(C, 2));
    s.bytes = k.bytes();
    return s;
}
// monolithic call: [query half][key/value half][KV]
struct XAddWs {
    void *q, *kv;
    float* KV;
    size_t qb, kvb, bytes;
};
XAddWs carve_xadd(const immtsf_fusion_cfg* c, void* base) {
    const size_t BT = (size_t)c->B * c->T, d = c->d;
    Carver k(base);
    XAddWs w;
    w.qb = carve_q(c, nullptr).bytes;
    w.kvb = carve_kv(c, nullptr).bytes;
    w.q = k.take<unsigned char>(w.qb);
    w.kv = k.take<unsigned char>(w.kvb);
    w.KV = k.take<float>(BT * 2 * d);
    w.bytes = k.bytes();
    return w;
}
XAddWs carve_xadd_scratch(const immtsf_fusion_cfg* c, void* base) {     // same shape: [q scratch][kv scratch][dKV]
    const size_t BT = (size_t)c->B * c->T, d = c->d;
    Carver k(base);
    XAddWs w;
    w.qb = carve_q_scratch(c, nullptr).bytes;
    w.kvb = carve_kv_scratch(c, nullptr).bytes;
    w.q = k.take<unsigned char>(w.qb);
    w.kv = k.take<unsigned char>(w.kvb);
    w.KV = k.take<float>(BT * 2 * d);
    w.bytes = k.bytes();
    return w;
}

// batched (window, head) view of a (B*T, ld) activation: element stride T*ld per window, hd per head
inline void batch_bh(GemmArgs& g, int B, int H, long sA_o, long sA_i, long sB_o, long sB_i, long sC_o, long sC_i) {
    g.nbatch = B * H;
    g.batch_inner = H;
    g.sA_o = sA_o; g.sA_i = sA_i; g.sB_o = sB_o; g.sB_i = sB_i; g.sC_o = sC_o; g.sC_i = sC_i;
}

bool bad_x(const immtsf_fusion_cfg* cfg) { return bad_cfg(cfg) || cfg->C <= 0; }

// bf16 mode, T x T attention over <= 32 prediction steps: the one-launch MFMA kernels of attn.hip instead of batched GEMMs + row softmax
// (IMMTSF_XATTN_SMALL=0: the GEMM path, for A/B measurements)
bool xattn_small(const immtsf_fusion_cfg* c) {
    constexpr bool on = true;
    return on && c->precision == 1 && xattn_small_supported(c->T, c->H, c->d / c->H);
}

// ---- the three folded products of the query half (parameters only) as ONE launch of three independent jobs:
//   job A, workgroups [0, na):  W_Qf[m][n] = sum_k W_in,q[m][k] W_q[k][n]   (d x C, K = d): a wave per row m, lanes stride k
//   job B, [na, na + nb):       W_HO[c][e] = sum_k W_res[c][k] W_out[k][e]  (C x d, K = d): 64 columns per workgroup, the lanes'
//                               16 k-groups meet by permlane swaps inside a wave and through LDS across the four waves
//   job C, the rest:            t_HO[c] = sum_k W_res[c][k] b_out[k];  b_HO = t_HO + b_res
// (As GEMM launches they were a zero-fill, a skinny product, a split-K product and a mat-vec: four dependent launches of
// parameter-only work on the text side's forward chain.)  Exact fp32.  C <= 16, d % 4 == 0.
constexpr int QF_C = 16;
struct QFoldJob { const float *Win, *Wq, *Wres, *Wout, *bout, *bres; float *WQf, *WHO, *bHO, *tHO; int d, C; };
__global__ __launch_bounds__(256) void qfold_kernel(QFoldJob q, int na, int nb) {
    __shared__ __attribute__((aligned(16))) float red[4][QF_C][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, d = q.d, C = q.C;
    int bid = blockIdx.x;
    if (bid < na) {                 // ---- job A
        const int m = bid * 4 + wave;
        if (m >= d) return;
        float acc[QF_C];
#pragma unroll
        for (int n = 0; n < QF_C; ++n) acc[n] = 0.f;
#pragma unroll 4
        for (int k = lane; k < d; k += 64) {      // (unrolled: a round trip per k step otherwise)
            const float a = q.Win[(size_t)m * d + k];
#pragma unroll
            for (int n = 0; n < QF_C; ++n)
                if (n < C) acc[n] = fmaf(a, q.Wq[(size_t)k * C + n], acc[n]);
        }
#pragma unroll
        for (int n = 0; n < QF_C; ++n)
            if (n < C) {
                const float v = wave_sum(acc[n]);
                if (lane == 0) q.WQf[(size_t)m * C + n] = v;
            }
        return;
    }
    bid -= na;
    if (bid < nb) {                 // ---- job B: columns e0 .. e0 + 64, thread = (4-column group eq, k-group kg)
        const int eq = tid & 15, kg = tid >> 4, e = bid * 64 + eq * 4;
        float4 acc[QF_C];
#pragma unroll
        for (int c = 0; c < QF_C; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < d) {
            constexpr int U = 8;          // k steps whose loads are in flight together (12 workgroups: latency is all there is)
            int k = kg;
            for (; k + 16 * (U - 1) < d; k += 16 * U) {
                float4 w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) w[u] = *reinterpret_cast<const float4*>(q.Wout + (size_t)(k + 16 * u) * d + e);
#pragma unroll
                for (int c = 0; c < QF_C; ++c)
                    if (c < C) {
                        float r[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) r[u] = q.Wres[(size_t)c * d + k + 16 * u];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            acc[c].x = fmaf(r[u], w[u].x, acc[c].x); acc[c].y = fmaf(r[u], w[u].y, acc[c].y);
                            acc[c].z = fmaf(r[u], w[u].z, acc[c].z); acc[c].w = fmaf(r[u], w[u].w, acc[c].w);
                        }
                    }
            }
            for (; k < d; k += 16) {
                const float4 w = *reinterpret_cast<const float4*>(q.Wout + (size_t)k * d + e);
#pragma unroll
                for (int c = 0; c < QF_C; ++c)
                    if (c < C) {
                        const float r = q.Wres[(size_t)c * d + k];
                        acc[c].x = fmaf(r, w.x, acc[c].x); acc[c].y = fmaf(r, w.y, acc[c].y);
                        acc[c].z = fmaf(r, w.z, acc[c].z); acc[c].w = fmaf(r, w.w, acc[c].w);
                    }
            }
        }
        // the four k-groups of a wave sit 16 lanes apart: lane ^ 16, lane ^ 32
#pragma unroll
        for (int c = 0; c < QF_C; ++c)
            if (c < C) {
                acc[c].x = xor32_sum(xor16_sum(acc[c].x)); acc[c].y = xor32_sum(xor16_sum(acc[c].y));
                acc[c].z = xor32_sum(xor16_sum(acc[c].z)); acc[c].w = xor32_sum(xor16_sum(acc[c].w));
                if (lane < 16) *reinterpret_cast<float4*>(&red[wave][c][eq * 4]) = acc[c];
            }
        __syncthreads();
        for (int x = tid; x < C * 64; x += 256) {
            const int c = x >> 6, col = x & 63;
            if (bid * 64 + col < d) q.WHO[(size_t)c * d + bid * 64 + col] = (red[0][c][col] + red[1][c][col]) + (red[2][c][col] + red[3][c][col]);
        }
        return;
    }
    bid -= nb;                      // ---- job C
    const int c = bid * 4 + wave;
    if (c >= C) return;
    float a = 0.f;
    for (int k = lane; k < d; k += 64) a = fmaf(q.Wres[(size_t)c * d + k], q.bout[k], a);
    a = wave_sum(a);
    if (lane == 0) { q.tHO[c] = a; q.bHO[c] = a + q.bres[c]; }
}
bool qfold_one_launch(const immtsf_fusion_cfg* c, const immtsf_xadd_params* p, const float* fold) {
    constexpr bool on = true;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p->attn_out_w) | reinterpret_cast<uintptr_t>(fold);
    return on && c->C <= QF_C && (c->d & 3) == 0 && (a & 15) == 0;
}

// the query projection (forward, backward) and the context gradient (backward) as operands formed inside the tile kernels from
// their C-column inputs: decided from what the forward and the backward both see, so that they agree on whether Qi exists
bool xattn_gen(const immtsf_fusion_cfg* c, const immtsf_xadd_params* p, const float* Y_ts, const QFold& f) {
    constexpr bool on = true;
    const uintptr_t a = reinterpret_cast<uintptr_t>(Y_ts) | reinterpret_cast<uintptr_t>(f.WQf) | reinterpret_cast<uintptr_t>(f.WHO) |
                        reinterpret_cast<uintptr_t>(p->attn_in_b);
    return on && xattn_small(c) && xattn_small_generates(c->d / c->H, c->C) && (a & 15) == 0;
}

}  // namespace

extern "C" {

size_t immtsf_mmf_xattn_kv_workspace_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_kv(cfg, nullptr).bytes; }
size_t immtsf_mmf_xattn_kv_scratch_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_kv_scratch(cfg, nullptr).bytes; }
size_t immtsf_mmf_xattn_q_workspace_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_q(cfg, nullptr).bytes; }
size_t immtsf_mmf_xattn_q_scratch_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_q_scratch(cfg, nullptr).bytes; }
size_t immtsf_mmf_xattn_add_workspace_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_xadd(cfg, nullptr).bytes; }
size_t immtsf_mmf_xattn_add_scratch_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_xadd_scratch(cfg, nullptr).bytes; }

int immtsf_mmf_xattn_kv_forward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt, float* KV,
                                void* workspace, size_t workspace_bytes, immtsf_stream_t stream) {
    if (bad_x(cfg) || !p || !E_txt || !KV || !workspace) return IMMTSF_EINVAL;
    KVWs w = carve_kv(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int d = cfg->d, BT = cfg->B * cfg->T, prec = cfg->precision;
    const bool hf = xadd_hf(cfg);
    KVW W;
    CHECK(kv_weights(cfg, p, w, s, &W));
    Mat E = cmat(E_txt);
    if (hf && cfg->in_h) {
        E.h = const_cast<void*>(cfg->in_h);
    } else if (hf) {
        CHECK(launch_f32_to_bf16(E_txt, w.E.h, (size_t)BT * d, s));
        E.h = w.E.h;
    }
    {   // W_KVf = [W_in,k W_k ; W_in,v W_v]   (two d x d x d products, one launch)
        GemmArgs g = gemm_args(d, d, d, d, d, d);
        g.nprob = 2;
        set_problem2(g, 0, W.ink, W.k, w.WKVf, nullptr);
        set_problem2(g, 1, W.inv, W.v, mat_off(w.WKVf, (size_t)d * d), nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    {   // (k | v) = E W_KVf^T + (b_k | b_v)
        GemmArgs g = gemm_args(BT, 2 * d, d, d, d, 2 * d);
        set_problem2(g, 0, E, w.WKVf, mat(KV), p->attn_in_b + d);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    return IMMTSF_OK;
}

size_t immtsf_mmf_xattn_q_fold_floats(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : qfold_floats(cfg); }

int immtsf_mmf_xattn_q_fold(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, float* fold, immtsf_stream_t stream) {
    if (bad_x(cfg) || !p || !fold) return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int d = cfg->d, C = cfg->C, prec = cfg->precision;
    const QFold f = qfold_at(cfg, fold);
    if (qfold_one_launch(cfg, p, fold)) {
        const QFoldJob q{p->attn_in_w, p->proj_q_w, p->res_w, p->attn_out_w, p->attn_out_b, p->res_b, f.WQf, f.WHO, f.bHO, f.tHO, d, C};
        const int na = (d + 3) / 4, nb = (d + 63) / 64, nc = (C + 3) / 4;
        hipLaunchKernelGGL(qfold_kernel, dim3(na + nb + nc), dim3(256), 0, s, q, na, nb);
        IMMTSF_LAUNCH_CHECK();
        return IMMTSF_OK;
    }
    {   // both folded products split K (12 tiles each otherwise): ONE zero-fill for the two adjacent outputs
        CHECK(launch_fill(f.WQf, 0.f, (size_t)2 * d * C, s));       // (a kernel: memset nodes misbehave under graph replay)
    }
    {   // W_Qf = W_in,q W_q  (d x C)
        GemmArgs g = gemm_args(d, C, d, d, C, C);
        set_problem(g, 0, p->attn_in_w, p->proj_q_w, f.WQf, nullptr);
        g.c_prezeroed = 1;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    {   // W_HO = W_res W_out  (C x d);  t = W_res b_out;  b_HO = t + b_res
        GemmArgs g = gemm_args(C, d, d, d, d, d);
        set_problem(g, 0, p->res_w, p->attn_out_w, f.WHO, nullptr);
        g.c_prezeroed = 1;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    return launch_matvec(p->res_w, d, p->attn_out_b, p->res_b, C, d, f.bHO, nullptr, 1.f, s, f.tHO);
}

int immtsf_mmf_xattn_q_forward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts, const float* KV,
                               const uint8_t* M_txt, const float* fold, float* Y_out, void* workspace, size_t workspace_bytes,
                               immtsf_stream_t stream) {
    if (bad_x(cfg) || !p || !Y_ts || !KV || !M_txt || !Y_out || !workspace) return IMMTSF_EINVAL;
    QWs w = carve_q(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, T = cfg->T, d = cfg->d, C = cfg->C, H = cfg->H, hd = d / H, BT = B * T, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    const long TT2 = (long)T * T;
    const float* Ki = KV;
    const float* Vi = KV + d;
    if (!fold) {
        CHECK(immtsf_mmf_xattn_q_fold(cfg, p, w.fold, stream));
        fold = w.fold;
    }
    const QFold f = qfold_at(cfg, const_cast<float*>(fold));
    const bool gen = xattn_gen(cfg, p, Y_ts, f);      // the tile kernel forms Qi = Y W_Qf^T + b_q itself
    if (!gen) {   // Qi = Y W_Qf^T + b_q
        GemmArgs g = gemm_args(BT, d, C, C, C, d);
        set_problem(g, 0, Y_ts, f.WQf, w.Qi, p->attn_in_b);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    if (xattn_small(cfg)) {      // few prediction steps: scores, softmax, dropout and A V in one launch
        const XattnGen xg{Y_ts, f.WQf, p->attn_in_b, nullptr, nullptr, gen ? C : 0};
        CHECK(launch_xattn_small_fwd(gen ? nullptr : w.Qi, KV, M_txt, B, T, H, hd, sqrtf(1.0f / (float)hd), drop, SITE_XADD_ATTN, w.Pm, w.Am, w.O,
                                     s, &xg));
    } else {
        {   // scores[b,h] = scale * Qi_h Ki_h^T
            GemmArgs g = gemm_args(T, T, hd, d, 2 * d, T);
            set_problem(g, 0, w.Qi, Ki, w.Pm, nullptr);
            g.alpha = sqrtf(1.0f / (float)hd);
            batch_bh(g, B, H, (long)T * d, hd, (long)T * 2 * d, hd, (long)H * TT2, TT2);
            CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
        }
        CHECK(launch_softmax_rows_fwd(w.Pm, w.Am, B, H, T, T, M_txt, drop, SITE_XADD_ATTN, 0, s));
        {   // O_h = A V_h   (zero for the no-text windows: their attention rows are zero)
            GemmArgs g = gemm_args(T, hd, T, T, 2 * d, d);
            set_problem(g, 0, w.Am, Vi, w.O, nullptr);
            batch_bh(g, B, H, (long)H * TT2, TT2, (long)T * 2 * d, hd, (long)T * d, hd);
            CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        }
    }
    // delta = M ? O W_HO^T + b_HO : b_res   (residual_head(where(M, out_proj(O), 0)))
    if (xadd_head_supported(C, d))        // head + LayerNorm(C) + dropout + blend in one kernel (w.delta stays unused)
        return launch_xadd_head_fwd(w.O, f.WHO, f.bHO, p->res_b, Y_ts, M_txt, BT, T, C, d, p->ln_w, p->ln_b, cfg->kappa, w.xhatC, w.rstdC,
                                    Y_out, drop, SITE_XADD_OUT, s);
    {
        GemmArgs g = gemm_args(BT, C, d, d, d, C);
        set_problem(g, 0, w.O, f.WHO, w.delta, f.tHO);
        g.row_flag = M_txt; g.row_flag_div = T; g.add_vec = p->res_b;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    return launch_ln_blend_fwd(w.delta, Y_ts, M_txt, BT, T, C, p->ln_w, p->ln_b, cfg->kappa, w.xhatC, w.rstdC, Y_out, drop,
                               SITE_XADD_OUT, s);
}

// parameter gradients of the query half from what its data-path backward left in `scratch` (ddelta, dQi) and the
// forward workspace (O): everything here is off the path to dY_ts / dKV
int immtsf_mmf_xattn_q_backward_params(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts,
                                       const uint8_t* M_txt, const float* fold, void* workspace, size_t workspace_bytes,
                                       void* scratch, size_t scratch_bytes, const immtsf_xadd_params* gr, immtsf_stream_t stream) {
    if (bad_x(cfg) || !p || !gr || !Y_ts || !M_txt || !workspace || !scratch) return IMMTSF_EINVAL;
    QWs w = carve_q(cfg, workspace);
    QScratch sc = carve_q_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, T = cfg->T, d = cfg->d, C = cfg->C, BT = B * T, prec = cfg->precision;
    (void)fold;
    // the two split-K weight-gradient products below land in scratch (dW_Qf, dW_HO: carved back to back): ONE zero-fill for
    // both instead of the launcher's own fill per output and per bias gradient
    int pz = 0;
    {
        const size_t nbytes = (size_t)((char*)(sc.dWHO + (size_t)C * d) - (char*)sc.dWQf);
        CHECK(launch_fill(sc.dWQf, 0.f, nbytes / sizeof(float), s));
        pz = 1;
        if (!cfg->grads_prezeroed) {       // the bias gradients those two GEMMs also reduce must read zero as well
            CHECK(launch_fill(gr->res_b, 0.f, (size_t)C, s));
            CHECK(launch_fill(gr->attn_in_b, 0.f, (size_t)d, s));
        }
    }
    // LayerNorm(C)'s two parameter gradients and s_live (column sums of ddelta over the windows with text) in one launch, then
    // d b_out = W_res^T s_live and the first term of dW_res = s_live b_out^T + dW_HO W_out^T in another
    constexpr bool head_small = true;
    const bool hs = head_small && head_sums_supported(BT, C);
    if (hs) {
        CHECK(launch_head_sums(sc.dn, w.xhatC, sc.ddelta, M_txt, T, BT, C, gr->ln_w, gr->ln_b, sc.slive, s));
        CHECK(launch_head_outer(p->res_w, d, sc.slive, p->attn_out_b, C, gr->attn_out_b, gr->res_w, s));
    } else {
        CHECK(launch_colsum2(sc.dn, w.xhatC, BT, C, C, gr->ln_w, gr->ln_b, sc.red, s, true));
    }
    {   // dW_HO = ddelta^T O (O is zero in the no-text windows);  d b_res = column sums of ddelta over ALL rows
        GemmArgs h = gemm_args(C, d, BT, C, d, d);
        set_problem(h, 0, sc.ddelta, w.O, sc.dWHO, nullptr, gr->res_b);
        h.c_prezeroed = pz;
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, s));
    }
    if (!hs) {
        CHECK(launch_mask_rows(sc.ddelta, BT, C, M_txt, T, s));       // from here on only the windows with text
        CHECK(launch_colsum(sc.ddelta, nullptr, BT, nullptr, C, C, sc.slive, 0, sc.red, s, true));
    }
    {   // chain rule through W_HO = W_res W_out, b_HO = W_res b_out + b_res
        if (!hs) {
            CHECK(launch_matvec_t(p->res_w, d, sc.slive, C, d, gr->attn_out_b, 0, s));      // d b_out = W_res^T s_live
            CHECK(launch_outer(sc.slive, p->attn_out_b, C, d, gr->res_w, d, s));             // dW_res = s_live b_out^T + dW_HO W_out^T
        }
        GemmArgs g = gemm_args(C, d, d, d, d, d);
        set_problem(g, 0, sc.dWHO, p->attn_out_w, gr->res_w, nullptr);
        g.accumulate = 1;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
        GemmArgs h = gemm_args(d, d, C, d, d, d);                                          // dW_out = W_res^T dW_HO
        set_problem(h, 0, p->res_w, sc.dWHO, gr->attn_out_w, nullptr);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, s));
    }
    {   // dW_Qf = dQi^T Y, d b_q = column sums of dQi;  then dW_in,q = dW_Qf W_q^T, dW_q = W_in,q^T dW_Qf
        GemmArgs h = gemm_args(d, C, BT, d, C, C);
        set_problem(h, 0, sc.dQi, Y_ts, sc.dWQf, nullptr, gr->attn_in_b);
        h.c_prezeroed = pz;
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, s));
        GemmArgs g = gemm_args(d, d, C, C, C, d);
        set_problem(g, 0, sc.dWQf, p->proj_q_w, gr->attn_in_w, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
        GemmArgs h2 = gemm_args(d, C, d, d, C, C);
        set_problem(h2, 0, p->attn_in_w, sc.dWQf, gr->proj_q_w, nullptr);
        prezeroed(h2, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h2, s));
    }
    return IMMTSF_OK;
}

// defer_params != 0: only the data path (dY_ts, dKV) runs; the caller must keep `workspace` and `scratch` alive and call
// immtsf_mmf_xattn_q_backward_params with them later (any stream ordered after this call) to get the parameter gradients
int immtsf_mmf_xattn_q_backward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts, const float* KV,
                                const uint8_t* M_txt, const float* fold, const float* dY_out, float* dY_ts, float* dKV,
                                void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                const immtsf_xadd_params* gr, int32_t defer_params, immtsf_stream_t stream) {
    if (bad_x(cfg) || !p || !gr || !Y_ts || !KV || !M_txt || !dY_out || !dY_ts || !dKV || !workspace || !scratch)
        return IMMTSF_EINVAL;
    QWs w = carve_q(cfg, workspace);
    QScratch sc = carve_q_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, T = cfg->T, d = cfg->d, C = cfg->C, H = cfg->H, hd = d / H, BT = B * T, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    const long TT2 = (long)T * T;
    const float scale = sqrtf(1.0f / (float)hd);
    const float* Ki = KV;
    const float* Vi = KV + d;
    if (!fold) fold = w.fold;
    const QFold f = qfold_at(cfg, const_cast<float*>(fold));
    unsigned short* dKV_h = (xadd_hf(cfg) && cfg->out_h) ? static_cast<unsigned short*>(cfg->out_h) : nullptr;   // bf16 image of dKV

    CHECK(launch_ln_blend_bwd(dY_out, M_txt, BT, T, C, p->ln_w, w.xhatC, w.rstdC, cfg->kappa, dY_ts, sc.dn, sc.ddelta, drop,
                              SITE_XADD_OUT, s));
    const bool gen = xattn_gen(cfg, p, Y_ts, f);      // the tile kernel forms dO (and Qi again) itself
    if (!gen) {   // dO = where(M, ddelta W_HO, 0)
        GemmArgs g = gemm_args(BT, d, C, C, d, d);
        set_problem(g, 0, sc.ddelta, f.WHO, sc.dO, nullptr);
        g.row_flag = M_txt; g.row_flag_div = T;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    if (xattn_small(cfg)) {
        const XattnGen xg{Y_ts, f.WQf, p->attn_in_b, sc.ddelta, f.WHO, gen ? C : 0};
        CHECK(launch_xattn_small_bwd(gen ? nullptr : w.Qi, KV, gen ? nullptr : sc.dO, w.Pm, w.Am, M_txt, B, T, H, hd, scale, drop, SITE_XADD_ATTN,
                                     sc.dQi, dKV, dKV_h, s, &xg));
    } else {
        {   // dA[b,h] = dO_h V_h^T ;  dV_h = A^T dO_h
            GemmArgs g = gemm_args(T, T, hd, d, 2 * d, T);
            set_problem(g, 0, sc.dO, Vi, sc.dA, nullptr);
            batch_bh(g, B, H, (long)T * d, hd, (long)T * 2 * d, hd, (long)H * TT2, TT2);
            CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
            GemmArgs h = gemm_args(T, hd, T, T, d, 2 * d);
            set_problem(h, 0, w.Am, sc.dO, dKV + d, nullptr);
            if (dKV_h) h.p[0].Ch = dKV_h + d;
            batch_bh(h, B, H, (long)H * TT2, TT2, (long)T * d, hd, (long)T * 2 * d, hd);
            CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, s));
        }
        CHECK(launch_softmax_rows_bwd(sc.dA, w.Pm, B, H, T, T, drop, SITE_XADD_ATTN, s));
        {   // dQ_h = scale dS K_h ; dK_h = scale dS^T Q_h
            GemmArgs g = gemm_args(T, hd, T, T, 2 * d, d);
            set_problem(g, 0, sc.dA, Ki, sc.dQi, nullptr);
            g.alpha = scale;
            batch_bh(g, B, H, (long)H * TT2, TT2, (long)T * 2 * d, hd, (long)T * d, hd);
            CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
            GemmArgs h = gemm_args(T, hd, T, T, d, 2 * d);
            set_problem(h, 0, sc.dA, w.Qi, dKV, nullptr);
            if (dKV_h) h.p[0].Ch = dKV_h;
            h.alpha = scale;
            batch_bh(h, B, H, (long)H * TT2, TT2, (long)T * d, hd, (long)T * 2 * d, hd);
            CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, s));
        }
    }
    {   // dY += dQi W_Qf
        GemmArgs g = gemm_args(BT, C, d, d, C, C);
        set_problem(g, 0, sc.dQi, f.WQf, dY_ts, nullptr);
        g.accumulate = 1;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    if (defer_params) return IMMTSF_OK;
    Fork fk(s);
    CHECK(immtsf_mmf_xattn_q_backward_params(cfg, p, Y_ts, M_txt, fold, workspace, workspace_bytes, scratch, scratch_bytes, gr, fk.fork()));
    return fk.join();
}

int immtsf_mmf_xattn_kv_backward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt, const float* dKV,
                                 float* dE_txt, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                 const immtsf_xadd_params* gr, immtsf_stream_t stream) {
    if (bad_x(cfg) || !p || !gr || !E_txt || !dKV || !dE_txt || !workspace || !scratch) return IMMTSF_EINVAL;
    KVWs w = carve_kv(cfg, workspace);
    KVScratch sc = carve_kv_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int d = cfg->d, BT = cfg->B * cfg->T, prec = cfg->precision;
    const bool hf = xadd_hf(cfg);
    KVW W;
    CHECK(kv_weights(cfg, p, w, s, &W));
    Mat dK = cmat(dKV), E = cmat(E_txt, (hf && cfg->aux_h) ? cfg->aux_h : w.E.h);
    if (hf && cfg->in_h) {
        dK.h = const_cast<void*>(cfg->in_h);
    } else if (hf) {
        CHECK(launch_f32_to_bf16(dKV, sc.dKV.h, (size_t)BT * 2 * d, s));
        dK.h = sc.dKV.h;
    }
    Fork fk(s);
    {   // dE = (dK | dV) W_KVf
        GemmArgs g = gemm_args(BT, d, 2 * d, 2 * d, d, d);
        set_problem2(g, 0, dK, w.WKVf, mat(dE_txt, hf ? cfg->out_h : nullptr), nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    {   // dW_KVf = (dK | dV)^T E ; (d b_k | d b_v) = column sums;  then the chain rule through W_{K,V}f = W_in W_proj
        hipStream_t f = fk.fork();
        GemmArgs h = gemm_args(2 * d, d, BT, 2 * d, d, d);
        set_problem2(h, 0, dK, E, sc.dWKVf, nullptr, gr->attn_in_b + d);
        h.c_prezeroed = 0;           // scratch output: see the query half
        h.ws = sc.sk; h.ws_bytes = sc.skb;
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, f));
        GemmArgs g = gemm_args(d, d, d, d, d, d);                 // dW_in,{k,v} = dW_{K,V}f W_{k,v}^T
        g.nprob = 2;
        set_problem2(g, 0, sc.dWKVf, W.k, mat(gr->attn_in_w + (size_t)d * d), nullptr);
        set_problem2(g, 1, mat_off(sc.dWKVf, (size_t)d * d), W.v, mat(gr->attn_in_w + (size_t)2 * d * d), nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, f));
        GemmArgs h2 = gemm_args(d, d, d, d, d, d);                // dW_{k,v} = W_in,{k,v}^T dW_{K,V}f
        h2.nprob = 2;
        set_problem2(h2, 0, W.ink, sc.dWKVf, mat(gr->proj_k_w), nullptr);
        set_problem2(h2, 1, W.inv, mat_off(sc.dWKVf, (size_t)d * d), mat(gr->proj_v_w), nullptr);
        prezeroed(h2, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h2, f));
    }
    return fk.join();
}

int immtsf_mmf_xattn_add_forward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts,
                                 const float* E_txt, const uint8_t* M_txt, float* Y_out, void* workspace,
                                 size_t workspace_bytes, immtsf_stream_t stream) {
    if (bad_x(cfg) || !workspace) return IMMTSF_EINVAL;
    XAddWs w = carve_xadd(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    CHECK(immtsf_mmf_xattn_kv_forward(cfg, p, E_txt, w.KV, w.kv, w.kvb, stream));
    return immtsf_mmf_xattn_q_forward(cfg, p, Y_ts, w.KV, M_txt, nullptr, Y_out, w.q, w.qb, stream);
}

int immtsf_mmf_xattn_add_backward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts,
                                  const float* E_txt, const uint8_t* M_txt, const float* dY_out, float* dY_ts,
                                  float* dE_txt, void* workspace, size_t workspace_bytes, void* scratch,
                                  size_t scratch_bytes, const immtsf_xadd_params* gr, immtsf_stream_t stream) {
    if (bad_x(cfg) || !workspace || !scratch) return IMMTSF_EINVAL;
    XAddWs w = carve_xadd(cfg, workspace);
    XAddWs sc = carve_xadd_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    CHECK(immtsf_mmf_xattn_q_backward(cfg, p, Y_ts, w.KV, M_txt, nullptr, dY_out, dY_ts, sc.KV, w.q, w.qb, sc.q, sc.qb, gr, 0, stream));
    return immtsf_mmf_xattn_kv_backward(cfg, p, E_txt, sc.KV, dE_txt, w.kv, w.kvb, sc.kv, sc.kvb, gr, stream);
}

}  // extern "C"
