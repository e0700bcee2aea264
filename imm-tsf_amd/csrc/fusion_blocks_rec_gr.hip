// Block-level entry points for TTF_RecAvg (fusions/TTF_RecAvg.py:54-112) and MMF_GR_Add (fusions/MMF_GR_Add.py:31-61).
#include "../../include/immtsf.h"
#include "gemm.hpp"
#include "gru.hpp"
#include "recavg.hpp"
#include "rowops.hpp"
#include "tail.hpp"
#include "block_util.hpp"
#include <math.h>

namespace {

// ------------------------------------------------------------------------------------------------ TTF_RecAvg
struct RecWs {
    unsigned char *mask, *mtxt;
    int *lengths, *offsets, *rowmap, *seg;
    float *Vp, *Eraw, *denom, *xhat, *rstd, *z;
    size_t bytes;
};
RecWs carve_rec(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, N = c->N, T = c->T, d = c->d, R = B * N, BT = B * T;
    Carver k(base);
    RecWs w;
    w.mask = k.take<unsigned char>(R);
    w.mtxt = k.take<unsigned char>(B);
    w.lengths = k.take<int>(B);
    w.offsets = k.take<int>(B + 1);
    w.rowmap = k.take<int>(R);
    w.seg = k.take<int>(R);
    w.Vp = k.take<float>(R * d);
    w.Eraw = k.take<float>(BT * d);
    w.denom = k.take<float>(BT);
    w.xhat = k.take<float>(BT * d);
    w.rstd = k.take<float>(BT);
    w.z = k.take<float>(BT * d);
    w.bytes = k.bytes();
    return w;
}
struct RecScratch {
    float *dz, *dEraw, *dVp, *dls_part, *red;
    size_t bytes;
};
RecScratch carve_rec_scratch(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, N = c->N, T = c->T, d = c->d, R = B * N, BT = B * T;
    Carver k(base);
    RecScratch s;
    s.dz = k.take<float>(BT * d);
    s.dEraw = k.take<float>(BT * d);
    s.dVp = k.take<float>(R * d);
    s.dls_part = k.take<float>(B);
    s.red = k.take<float>(ln_sums_scratch_floats(d, 2));        // (>= colsum_scratch_floats(d, 2): the fallback's scratch)
    s.bytes = k.bytes();
    return s;
}

// ------------------------------------------------------------------------------------------------ MMF_GR_Add
struct GRWs {
    float *x, *gi, *gl, *r, *z, *n, *hn, *h, *hprev, *xhat, *rstd, *g, *dd;
    size_t bytes;
};
GRWs carve_gr(const immtsf_fusion_cfg* c, int Hd, void* base) {
    const size_t BT = (size_t)c->B * c->T, C = c->C, d = c->d;
    Carver k(base);
    GRWs w;
    w.x = k.take<float>(BT * (C + d));
    w.gi = k.take<float>(BT * 3 * Hd);
    w.gl = k.take<float>(BT * C);
    w.r = k.take<float>(BT * Hd);
    w.z = k.take<float>(BT * Hd);
    w.n = k.take<float>(BT * Hd);
    w.hn = k.take<float>(BT * Hd);
    w.h = k.take<float>(BT * Hd);
    w.hprev = k.take<float>(BT * Hd);
    w.xhat = k.take<float>(BT * C);
    w.rstd = k.take<float>(BT);
    w.g = k.take<float>(BT * C);
    w.dd = k.take<float>(BT * C);
    w.bytes = k.bytes();
    return w;
}
struct GRScratch {
    float *dn, *ddelta, *dgl, *dh_in, *dgi, *dgh, *dx, *red;
    size_t bytes;
};
GRScratch carve_gr_scratch(const immtsf_fusion_cfg* c, int Hd, void* base) {
    const size_t BT = (size_t)c->B * c->T, C = c->C, d = c->d;
    Carver k(base);
    GRScratch s;
    s.dn = k.take<float>(BT * C);
    s.ddelta = k.take<float>(BT * C);
    s.dgl = k.take<float>(BT * C);
    s.dh_in = k.take<float>(BT * Hd);
    s.dgi = k.take<float>(BT * 3 * Hd);
    s.dgh = k.take<float>(BT * 3 * Hd);
    s.dx = k.take<float>(BT * (C + d));
    s.red = k.take<float>(64 * (C + d + 3 * Hd + 8) + colsum_scratch_floats(C, 2));
    s.bytes = k.bytes();
    return s;
}

}  // namespace

extern "C" {

size_t immtsf_ttf_recavg_workspace_bytes(const immtsf_fusion_cfg* cfg) { return bad_cfg(cfg) ? 0 : carve_rec(cfg, nullptr).bytes; }
size_t immtsf_ttf_recavg_scratch_bytes(const immtsf_fusion_cfg* cfg) { return bad_cfg(cfg) ? 0 : carve_rec_scratch(cfg, nullptr).bytes; }

int immtsf_ttf_recavg_forward(const immtsf_fusion_cfg* cfg, const immtsf_recavg_params* p, const float* notes,
                              const float* tau, const float* t_hat, float* E_txt, uint8_t* M_txt, void* workspace,
                              size_t workspace_bytes, int32_t* nan_flag, immtsf_stream_t stream) {
    if (bad_cfg(cfg) || !p || !notes || !tau || !t_hat || !E_txt || !M_txt || !workspace) return IMMTSF_EINVAL;
    if (cfg->N <= 0 || cfg->d_m <= 0) return IMMTSF_EINVAL;
    if (!p->input_proj_w && cfg->d != cfg->d_m) return IMMTSF_EINVAL;
    RecWs w = carve_rec(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, N = cfg->N, T = cfg->T, d = cfg->d, R = B * N, BT = B * T, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    const int* total = w.offsets + B;
    CHECK(launch_note_mask(notes, R, cfg->d_m, w.mask, nan_flag, s));
    CHECK(launch_ragged_index(w.mask, B, N, w.lengths, w.offsets, w.rowmap, w.seg, w.mtxt, s, M_txt));
    if (p->input_proj_w) {
        GemmArgs g = gemm_args(R, d, cfg->d_m, cfg->d_m, cfg->d_m, d);
        set_problem(g, 0, notes, p->input_proj_w, w.Vp, p->input_proj_b);
        g.dyn = total; g.dyn_which = 0; g.a_rowmap = w.rowmap;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    } else {
        CHECK(launch_gather_rows(notes, cfg->d_m, w.rowmap, total, R, d, w.Vp, d, s));
    }
    CHECK(launch_recavg_fwd(B, T, d, N, w.offsets, w.rowmap, tau, t_hat, p->log_recency_sigma, w.Vp, w.Eraw, w.denom, s));
    CHECK(launch_layernorm_fwd(w.Eraw, BT, d, p->ln_w, p->ln_b, 1e-5f, w.xhat, w.rstd, w.z, drop, SITE_REC_OUT, s));
    {
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem(g, 0, w.z, p->proj_w, E_txt, p->proj_b);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    return IMMTSF_OK;
}

int immtsf_ttf_recavg_backward(const immtsf_fusion_cfg* cfg, const immtsf_recavg_params* p, const float* notes,
                               const float* tau, const float* t_hat, const float* dE_txt, void* workspace,
                               size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                               const immtsf_recavg_params* gr, immtsf_stream_t stream) {
    if (bad_cfg(cfg) || !p || !gr || !notes || !tau || !t_hat || !dE_txt || !workspace || !scratch) return IMMTSF_EINVAL;
    RecWs w = carve_rec(cfg, workspace);
    RecScratch sc = carve_rec_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, N = cfg->N, T = cfg->T, d = cfg->d, R = B * N, BT = B * T, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    const int* total = w.offsets + B;
    Fork fk(s);
    {
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem(g, 0, dE_txt, p->proj_w, sc.dz, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(d, d, BT, d, d, d);
        set_problem(h, 0, dE_txt, w.z, gr->proj_w, nullptr, gr->proj_b);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    {   // LayerNorm backward with its parameter-gradient sums in the same pass; small / unaligned cases: the two separate passes
        const int rc = launch_layernorm_bwd_sums(sc.dz, BT, d, p->ln_w, w.xhat, w.rstd, sc.dEraw, drop, SITE_REC_OUT, gr->ln_w, gr->ln_b,
                                                 nullptr, sc.red, nullptr, 1, nullptr, s);
        if (rc == IMMTSF_EUNSUPPORTED) {
            CHECK(launch_layernorm_bwd(sc.dz, BT, d, p->ln_w, w.xhat, w.rstd, sc.dEraw, drop, SITE_REC_OUT, s));
            CHECK(launch_colsum2(sc.dz, w.xhat, BT, d, d, gr->ln_w, gr->ln_b, sc.red, s, true));
        } else {
            CHECK(rc);
        }
    }
    CHECK(launch_recavg_bwd(B, T, d, w.offsets, w.rowmap, tau, t_hat, p->log_recency_sigma, w.Vp, w.Eraw, w.denom, sc.dEraw,
                            sc.dVp, sc.dls_part, s, cfg->precision));
    CHECK(launch_colsum(sc.dls_part, nullptr, B, nullptr, 1, 1, gr->log_recency_sigma, 0, sc.red, s));
    if (p->input_proj_w) {
        GemmArgs h = gemm_args(d, cfg->d_m, R, d, cfg->d_m, cfg->d_m);
        set_problem(h, 0, sc.dVp, notes, gr->input_proj_w, nullptr, gr->input_proj_b);
        h.dyn = total; h.dyn_which = 1; h.b_rowmap = w.rowmap;
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    return fk.join();
}

size_t immtsf_mmf_gr_add_workspace_bytes(const immtsf_fusion_cfg* cfg, int32_t hidden) {
    return (bad_cfg(cfg) || cfg->C <= 0 || hidden <= 0) ? 0 : carve_gr(cfg, hidden, nullptr).bytes;
}
size_t immtsf_mmf_gr_add_scratch_bytes(const immtsf_fusion_cfg* cfg, int32_t hidden) {
    return (bad_cfg(cfg) || cfg->C <= 0 || hidden <= 0) ? 0 : carve_gr_scratch(cfg, hidden, nullptr).bytes;
}

int immtsf_mmf_gr_add_forward(const immtsf_fusion_cfg* cfg, int32_t Hd, const immtsf_gr_params* p, const float* Y_ts,
                              const float* E_txt, const uint8_t* M_txt, float* Y_out, void* workspace,
                              size_t workspace_bytes, immtsf_stream_t stream) {
    if (bad_cfg(cfg) || cfg->C <= 0 || Hd <= 0 || !p || !Y_ts || !E_txt || !M_txt || !Y_out || !workspace) return IMMTSF_EINVAL;
    GRWs w = carve_gr(cfg, Hd, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, T = cfg->T, d = cfg->d, C = cfg->C, BT = B * T, I = C + d, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    CHECK(launch_concat2(Y_ts, C, E_txt, d, BT, w.x, s));
    {   // input side of the GRU for every (b,t): gi = W_ih x + b_ih
        GemmArgs g = gemm_args(BT, 3 * Hd, I, I, I, 3 * Hd);
        set_problem(g, 0, w.x, p->w_ih, w.gi, p->b_ih);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    {   // gate logits
        GemmArgs g = gemm_args(BT, C, I, I, I, C);
        set_problem(g, 0, w.x, p->gate_w, w.gl, p->gate_b);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    CHECK(launch_gru_fwd(B, T, Hd, w.gi, p->w_hh, p->b_hh, w.r, w.z, w.n, w.hn, w.h, w.hprev, s));
    return launch_gr_tail_fwd(BT, T, C, Hd, w.h, p->res_w, p->res_b, p->ln_w, p->ln_b, w.gl, Y_ts, M_txt, w.xhat, w.rstd, w.g,
                              w.dd, Y_out, drop, SITE_GR_OUT, s);
}

int immtsf_mmf_gr_add_backward(const immtsf_fusion_cfg* cfg, int32_t Hd, const immtsf_gr_params* p, const float* Y_ts,
                               const float* E_txt, const uint8_t* M_txt, const float* dY_out, float* dY_ts, float* dE_txt,
                               void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                               const immtsf_gr_params* gr, immtsf_stream_t stream) {
    if (bad_cfg(cfg) || cfg->C <= 0 || Hd <= 0 || !p || !gr || !Y_ts || !E_txt || !M_txt || !dY_out || !dY_ts || !dE_txt ||
        !workspace || !scratch)
        return IMMTSF_EINVAL;
    GRWs w = carve_gr(cfg, Hd, workspace);
    GRScratch sc = carve_gr_scratch(cfg, Hd, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, T = cfg->T, d = cfg->d, C = cfg->C, BT = B * T, I = C + d, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    Fork fk(s);
    CHECK(launch_gr_tail_bwd(BT, T, C, Hd, dY_out, p->res_w, p->ln_w, w.xhat, w.rstd, w.g, w.dd, M_txt, sc.dn, sc.ddelta,
                             sc.dgl, sc.dh_in, drop, SITE_GR_OUT, s));
    CHECK(launch_colsum2(sc.dn, w.xhat, BT, C, C, gr->ln_w, gr->ln_b, sc.red, s, true));
    {   // residual_head: dW_r = ddelta^T h ; db_r
        GemmArgs h = gemm_args(C, Hd, BT, C, Hd, Hd);
        set_problem(h, 0, sc.ddelta, w.h, gr->res_w, nullptr, gr->res_b);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    CHECK(launch_gru_bwd(B, T, Hd, sc.dh_in, p->w_hh, w.r, w.z, w.n, w.hn, w.hprev, sc.dgi, sc.dgh, s));
    {   // recurrent weights: dW_hh = dgh^T h_prev ; db_hh
        GemmArgs h = gemm_args(3 * Hd, Hd, BT, 3 * Hd, Hd, Hd);
        set_problem(h, 0, sc.dgh, w.hprev, gr->w_hh, nullptr, gr->b_hh);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    {   // input weights: dW_ih = dgi^T x ; db_ih ; dx = dgi W_ih
        GemmArgs h = gemm_args(3 * Hd, I, BT, 3 * Hd, I, I);
        set_problem(h, 0, sc.dgi, w.x, gr->w_ih, nullptr, gr->b_ih);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
        GemmArgs g = gemm_args(BT, I, 3 * Hd, 3 * Hd, I, I);
        set_problem(g, 0, sc.dgi, p->w_ih, sc.dx, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    {   // gate net: dW_g = dgl^T x ; db_g ; dx += dgl W_g
        GemmArgs h = gemm_args(C, I, BT, C, I, I);
        set_problem(h, 0, sc.dgl, w.x, gr->gate_w, nullptr, gr->gate_b);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
        GemmArgs g = gemm_args(BT, I, C, C, I, I);
        set_problem(g, 0, sc.dgl, p->gate_w, sc.dx, nullptr);
        g.accumulate = 1;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    // dY = dYout (direct path: out = Y + (1-g) dd) + dx[:, :C] ; dE = dx[:, C:]
    CHECK(launch_axpy(dY_out, 1.f, dY_ts, BT * C, 0, s));       // (a copy kernel, not a memcpy node: see launch_fill's note in attn.hip)
    CHECK(launch_split2(sc.dx, C, d, BT, dY_ts, 1, dE_txt, s));
    return fk.join();
}

}  // extern "C"
