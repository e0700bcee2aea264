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

// ---- the block is split into a key/value half (depends only on E_txt: proj_k/proj_v + their MHA in-projections) and
// a query half (everything else).  The halves are separate C entry points so the host can run the key/value half --
// four of the block's six 2048x768x768 GEMMs forward, eight backward -- on the text stream beside the backbone, and
// start the backbone's backward as soon as the query half has produced dY_ts.  The monolithic entry points call both.
struct KVWs {
    float *K0, *V0;
    size_t bytes;
};
KVWs carve_kv(const immtsf_fusion_cfg* c, void* base) {
    const size_t BT = (size_t)c->B * c->T, d = c->d;
    Carver k(base);
    KVWs w;
    w.K0 = k.take<float>(BT * d);
    w.V0 = k.take<float>(BT * d);
    w.bytes = k.bytes();
    return w;
}
struct KVScratch {
    float *dK0, *dV0;
    size_t bytes;
};
KVScratch carve_kv_scratch(const immtsf_fusion_cfg* c, void* base) {
    const size_t BT = (size_t)c->B * c->T, d = c->d;
    Carver k(base);
    KVScratch s;
    s.dK0 = k.take<float>(BT * 2 * d);      // [dK0 | dV0] side by side, row pitch 2d: one K = 2d GEMM can read both
    s.dV0 = s.dK0 ? s.dK0 + d : nullptr;
    s.bytes = k.bytes();
    return s;
}
struct QWs {
    float *Q0, *Qi, *Pm, *Am, *O, *U, *delta, *xhatC, *rstdC;
    size_t bytes;
};
QWs carve_q(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, T = c->T, d = c->d, C = c->C, BT = B * T, S = B * c->H * T * T;
    const bool dropping = c->training && c->p_drop > 0.f;
    Carver k(base);
    QWs w;
    w.Q0 = k.take<float>(BT * d);
    w.Qi = k.take<float>(BT * d);
    w.Pm = k.take<float>(S);
    w.Am = dropping ? k.take<float>(S) : w.Pm;
    w.O = k.take<float>(BT * d);
    w.U = k.take<float>(BT * d);
    w.delta = k.take<float>(BT * C);
    w.xhatC = k.take<float>(BT * C);
    w.rstdC = k.take<float>(BT);
    w.bytes = k.bytes();
    return w;
}
struct QScratch {
    float *dn, *ddelta, *dU, *dO, *dA, *dQi, *dQ0, *red;
    size_t bytes;
};
QScratch carve_q_scratch(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, T = c->T, d = c->d, C = c->C, BT = B * T, S = B * c->H * T * T;
    Carver k(base);
    QScratch s;
    s.dn = k.take<float>(BT * C);
    s.ddelta = k.take<float>(BT * C);
    s.dU = k.take<float>(BT * d);
    s.dO = k.take<float>(BT * d);
    s.dA = k.take<float>(S);
    s.dQi = k.take<float>(BT * d);
    s.dQ0 = k.take<float>(BT * d);
    s.red = k.take<float>(64 * (d + C + 8));
    s.bytes = k.bytes();
    return s;
}
// monolithic call: [query half][key/value half][Ki][Vi]
struct XAddWs {
    void *q, *kv;
    float *Ki, *Vi;
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
    w.Ki = k.take<float>(BT * d);
    w.Vi = k.take<float>(BT * d);
    w.bytes = k.bytes();
    return w;
}
XAddWs carve_xadd_scratch(const immtsf_fusion_cfg* c, void* base) {     // same shape: [q scratch][kv scratch][dKi][dVi]
    const size_t BT = (size_t)c->B * c->T, d = c->d;
    Carver k(base);
    XAddWs w;
    w.qb = carve_q_scratch(c, nullptr).bytes;
    w.kvb = carve_kv_scratch(c, nullptr).bytes;
    w.q = k.take<unsigned char>(w.qb);
    w.kv = k.take<unsigned char>(w.kvb);
    w.Ki = k.take<float>(BT * d);
    w.Vi = k.take<float>(BT * d);
    w.bytes = k.bytes();
    return w;
}

// batched (window, head) view of a (B*T, d) activation: element stride T*d per window, hd per head
inline void batch_bh(GemmArgs& g, int B, int H, long sA_o, long sA_i, long sB_o, long sB_i, long sC_o, long sC_i) {
    g.nbatch = B * H;
    g.batch_inner = H;
    g.sA_o = sA_o; g.sA_i = sA_i; g.sB_o = sB_o; g.sB_i = sB_i; g.sC_o = sC_o; g.sC_i = sC_i;
}

bool bad_x(const immtsf_fusion_cfg* cfg) { return bad_cfg(cfg) || cfg->C <= 0; }

}  // namespace

extern "C" {

size_t immtsf_mmf_xattn_kv_workspace_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_kv(cfg, nullptr).bytes; }
size_t immtsf_mmf_xattn_kv_scratch_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_kv_scratch(cfg, nullptr).bytes; }
size_t immtsf_mmf_xattn_q_workspace_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_q(cfg, nullptr).bytes; }
size_t immtsf_mmf_xattn_q_scratch_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_q_scratch(cfg, nullptr).bytes; }
size_t immtsf_mmf_xattn_add_workspace_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_xadd(cfg, nullptr).bytes; }
size_t immtsf_mmf_xattn_add_scratch_bytes(const immtsf_fusion_cfg* cfg) { return bad_x(cfg) ? 0 : carve_xadd_scratch(cfg, nullptr).bytes; }

int immtsf_mmf_xattn_kv_forward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt, float* Ki,
                                float* Vi, void* workspace, size_t workspace_bytes, immtsf_stream_t stream) {
    if (bad_x(cfg) || !p || !E_txt || !Ki || !Vi || !workspace) return IMMTSF_EINVAL;
    KVWs w = carve_kv(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int d = cfg->d, BT = cfg->B * cfg->T, prec = cfg->precision;
    {   // K0, V0 = E {W_k, W_v}^T   (two problems, one launch)
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        g.nprob = 2;
        set_problem(g, 0, E_txt, p->proj_k_w, w.K0, nullptr);
        set_problem(g, 1, E_txt, p->proj_v_w, w.V0, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    {   // MHA in-projections of k, v
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        g.nprob = 2;
        set_problem(g, 0, w.K0, p->attn_in_w + (size_t)d * d, Ki, p->attn_in_b + d);
        set_problem(g, 1, w.V0, p->attn_in_w + (size_t)2 * d * d, Vi, p->attn_in_b + 2 * d);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    return IMMTSF_OK;
}

int immtsf_mmf_xattn_q_forward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts, const float* Ki,
                               const float* Vi, const uint8_t* M_txt, float* Y_out, void* workspace, size_t workspace_bytes,
                               immtsf_stream_t stream) {
    if (bad_x(cfg) || !p || !Y_ts || !Ki || !Vi || !M_txt || !Y_out || !workspace) return IMMTSF_EINVAL;
    QWs w = carve_q(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, T = cfg->T, d = cfg->d, C = cfg->C, H = cfg->H, hd = d / H, BT = B * T, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    const long TT2 = (long)T * T;
    {   // Q0 = Y W_q^T
        GemmArgs g = gemm_args(BT, d, C, C, C, d);
        set_problem(g, 0, Y_ts, p->proj_q_w, w.Q0, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    {   // MHA in-projection of q
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem(g, 0, w.Q0, p->attn_in_w, w.Qi, p->attn_in_b);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    {   // scores[b,h] = scale * Qi_h Ki_h^T
        GemmArgs g = gemm_args(T, T, hd, d, d, T);
        set_problem(g, 0, w.Qi, Ki, w.Pm, nullptr);
        g.alpha = sqrtf(1.0f / (float)hd);
        batch_bh(g, B, H, (long)T * d, hd, (long)T * d, hd, (long)H * TT2, TT2);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    CHECK(launch_softmax_rows_fwd(w.Pm, w.Am, B, H, T, T, M_txt, drop, SITE_XADD_ATTN, 0, s));
    {   // O_h = A V_h
        GemmArgs g = gemm_args(T, hd, T, T, d, d);
        set_problem(g, 0, w.Am, Vi, w.O, nullptr);
        batch_bh(g, B, H, (long)H * TT2, TT2, (long)T * d, hd, (long)T * d, hd);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    {   // out_proj, zero no-text windows (attn_out = where(M, attn_out, 0))
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem(g, 0, w.O, p->attn_out_w, w.U, p->attn_out_b);
        g.row_flag = M_txt; g.row_flag_div = T;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    if (xadd_head_supported(C, d))        // residual_head + LayerNorm(C) + dropout + blend in one kernel (w.delta stays unused)
        return launch_xadd_head_fwd(w.U, p->res_w, p->res_b, Y_ts, M_txt, BT, T, C, d, p->ln_w, p->ln_b, cfg->kappa, w.xhatC, w.rstdC,
                                    Y_out, drop, SITE_XADD_OUT, s);
    {   // residual_head
        GemmArgs g = gemm_args(BT, C, d, d, d, C);
        set_problem(g, 0, w.U, p->res_w, w.delta, p->res_b);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    return launch_ln_blend_fwd(w.delta, Y_ts, M_txt, BT, T, C, p->ln_w, p->ln_b, cfg->kappa, w.xhatC, w.rstdC, Y_out, drop,
                               SITE_XADD_OUT, s);
}

int immtsf_mmf_xattn_q_backward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts, const float* Ki,
                                const float* Vi, const uint8_t* M_txt, const float* dY_out, float* dY_ts, float* dKi,
                                float* dVi, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                const immtsf_xadd_params* gr, immtsf_stream_t stream) {
    if (bad_x(cfg) || !p || !gr || !Y_ts || !Ki || !Vi || !M_txt || !dY_out || !dY_ts || !dKi || !dVi || !workspace || !scratch)
        return IMMTSF_EINVAL;
    QWs w = carve_q(cfg, workspace);
    QScratch sc = carve_q_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, T = cfg->T, d = cfg->d, C = cfg->C, H = cfg->H, hd = d / H, BT = B * T, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    const long TT2 = (long)T * T;
    const float scale = sqrtf(1.0f / (float)hd);
    Fork fk(s);   // weight-gradient GEMMs may run on the side stream, joined before returning

    CHECK(launch_ln_blend_bwd(dY_out, M_txt, BT, T, C, p->ln_w, w.xhatC, w.rstdC, cfg->kappa, dY_ts, sc.dn, sc.ddelta, drop,
                              SITE_XADD_OUT, s));
    CHECK(launch_colsum2(sc.dn, w.xhatC, BT, C, C, gr->ln_w, gr->ln_b, sc.red, s));
    {   // residual_head
        GemmArgs g = gemm_args(BT, d, C, C, d, d);
        set_problem(g, 0, sc.ddelta, p->res_w, sc.dU, nullptr);
        g.row_flag = M_txt; g.row_flag_div = T;      // where(M, attn_out, 0): no gradient into no-text windows
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(C, d, BT, C, d, d);
        set_problem(h, 0, sc.ddelta, w.U, gr->res_w, nullptr, gr->res_b);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    {   // out_proj
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem(g, 0, sc.dU, p->attn_out_w, sc.dO, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(d, d, BT, d, d, d);
        set_problem(h, 0, sc.dU, w.O, gr->attn_out_w, nullptr, gr->attn_out_b);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    {   // dA[b,h] = dO_h V_h^T ;  dV_h = A^T dO_h
        GemmArgs g = gemm_args(T, T, hd, d, d, T);
        set_problem(g, 0, sc.dO, Vi, sc.dA, nullptr);
        batch_bh(g, B, H, (long)T * d, hd, (long)T * d, hd, (long)H * TT2, TT2);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
        GemmArgs h = gemm_args(T, hd, T, T, d, d);
        set_problem(h, 0, w.Am, sc.dO, dVi, nullptr);
        batch_bh(h, B, H, (long)H * TT2, TT2, (long)T * d, hd, (long)T * d, hd);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, s));
    }
    CHECK(launch_softmax_rows_bwd(sc.dA, w.Pm, B, H, T, T, drop, SITE_XADD_ATTN, s));
    {   // dQ_h = scale dS K_h ; dK_h = scale dS^T Q_h
        GemmArgs g = gemm_args(T, hd, T, T, d, d);
        set_problem(g, 0, sc.dA, Ki, sc.dQi, nullptr);
        g.alpha = scale;
        batch_bh(g, B, H, (long)H * TT2, TT2, (long)T * d, hd, (long)T * d, hd);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(T, hd, T, T, d, d);
        set_problem(h, 0, sc.dA, w.Qi, dKi, nullptr);
        h.alpha = scale;
        batch_bh(h, B, H, (long)H * TT2, TT2, (long)T * d, hd, (long)T * d, hd);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, s));
    }
    {   // MHA in-projection of q
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem(g, 0, sc.dQi, p->attn_in_w, sc.dQ0, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(d, d, BT, d, d, d);
        set_problem(h, 0, sc.dQi, w.Q0, gr->attn_in_w, nullptr, gr->attn_in_b);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    {   // proj_q: dY += dQ0 W_q ; dW_q = dQ0^T Y
        GemmArgs g = gemm_args(BT, C, d, d, C, C);
        set_problem(g, 0, sc.dQ0, p->proj_q_w, dY_ts, nullptr);
        g.accumulate = 1;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(d, C, BT, d, C, C);
        set_problem(h, 0, sc.dQ0, Y_ts, gr->proj_q_w, nullptr);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    return fk.join();
}

int immtsf_mmf_xattn_kv_backward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* E_txt,
                                 const float* dKi, const float* dVi, float* dE_txt, void* workspace, size_t workspace_bytes,
                                 void* scratch, size_t scratch_bytes, const immtsf_xadd_params* gr, immtsf_stream_t stream) {
    if (bad_x(cfg) || !p || !gr || !E_txt || !dKi || !dVi || !dE_txt || !workspace || !scratch) return IMMTSF_EINVAL;
    KVWs w = carve_kv(cfg, workspace);
    KVScratch sc = carve_kv_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int d = cfg->d, BT = cfg->B * cfg->T, prec = cfg->precision;
    Fork fk(s);
    {   // MHA in-projections of k, v
        GemmArgs g = gemm_args(BT, d, d, d, d, 2 * d);
        g.nprob = 2;
        set_problem(g, 0, dKi, p->attn_in_w + (size_t)d * d, sc.dK0, nullptr);
        set_problem(g, 1, dVi, p->attn_in_w + (size_t)2 * d * d, sc.dV0, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(d, d, BT, d, d, d);
        h.nprob = 2;
        set_problem(h, 0, dKi, w.K0, gr->attn_in_w + (size_t)d * d, nullptr, gr->attn_in_b + d);
        set_problem(h, 1, dVi, w.V0, gr->attn_in_w + (size_t)2 * d * d, nullptr, gr->attn_in_b + 2 * d);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    {   // proj_k / proj_v: dE = dK0 W_k + dV0 W_v ; dW_k = dK0^T E ; dW_v = dV0^T E
        if (p->proj_v_w == p->proj_k_w + (size_t)d * d) {
            // the two weights are adjacent (FlatTrainer's flat buffer, or any state-dict-order allocation): [W_k ; W_v] is one
            // (2d, d) matrix and dE = [dK0 | dV0] [W_k ; W_v] ONE GEMM with K = 2d instead of two K = d launches
            GemmArgs g = gemm_args(BT, d, 2 * d, 2 * d, d, d);
            set_problem(g, 0, sc.dK0, p->proj_k_w, dE_txt, nullptr);
            CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        } else {
            GemmArgs g = gemm_args(BT, d, d, 2 * d, d, d);
            set_problem(g, 0, sc.dK0, p->proj_k_w, dE_txt, nullptr);
            CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
            GemmArgs g2 = gemm_args(BT, d, d, 2 * d, d, d);
            set_problem(g2, 0, sc.dV0, p->proj_v_w, dE_txt, nullptr);
            g2.accumulate = 1;
            CHECK(immtsf_launch_gemm(GEMM_NN, prec, g2, s));
        }
        GemmArgs h = gemm_args(d, d, BT, 2 * d, d, d);
        h.nprob = 2;
        set_problem(h, 0, sc.dK0, E_txt, gr->proj_k_w, nullptr);
        set_problem(h, 1, sc.dV0, E_txt, gr->proj_v_w, nullptr);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    return fk.join();
}

int immtsf_mmf_xattn_add_forward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts,
                                 const float* E_txt, const uint8_t* M_txt, float* Y_out, void* workspace,
                                 size_t workspace_bytes, immtsf_stream_t stream) {
    if (bad_x(cfg) || !workspace) return IMMTSF_EINVAL;
    XAddWs w = carve_xadd(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    CHECK(immtsf_mmf_xattn_kv_forward(cfg, p, E_txt, w.Ki, w.Vi, w.kv, w.kvb, stream));
    return immtsf_mmf_xattn_q_forward(cfg, p, Y_ts, w.Ki, w.Vi, M_txt, Y_out, w.q, w.qb, stream);
}

int immtsf_mmf_xattn_add_backward(const immtsf_fusion_cfg* cfg, const immtsf_xadd_params* p, const float* Y_ts,
                                  const float* E_txt, const uint8_t* M_txt, const float* dY_out, float* dY_ts,
                                  float* dE_txt, void* workspace, size_t workspace_bytes, void* scratch,
                                  size_t scratch_bytes, const immtsf_xadd_params* gr, immtsf_stream_t stream) {
    if (bad_x(cfg) || !workspace || !scratch) return IMMTSF_EINVAL;
    XAddWs w = carve_xadd(cfg, workspace);
    XAddWs sc = carve_xadd_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    CHECK(immtsf_mmf_xattn_q_backward(cfg, p, Y_ts, w.Ki, w.Vi, M_txt, dY_out, dY_ts, sc.Ki, sc.Vi, w.q, w.qb, sc.q, sc.qb, gr,
                                      stream));
    return immtsf_mmf_xattn_kv_backward(cfg, p, E_txt, sc.Ki, sc.Vi, dE_txt, w.kv, w.kvb, sc.kv, sc.kvb, gr, stream);
}

}  // extern "C"
