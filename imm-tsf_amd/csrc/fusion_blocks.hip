// Block-level entry points of the C ABI: each function enqueues the whole forward (or backward) of one reference
// fusion module as a short sequence of HIP kernels on the caller's stream.  No allocation, no sync, no global
// state: graph-capturable.  Workspaces are carved from caller memory by `Carver` (256-byte aligned slices).
#include "../../include/immtsf.h"
#include "attn.hpp"
#include "gemm.hpp"
#include "rowops.hpp"
#include "tail.hpp"
#include "block_util.hpp"
#include <math.h>
#include <string.h>

namespace {

// ================================================================================================ TTF_T2V_XAttn
struct T2VWs {
    unsigned char *mask, *mtxt;
    int *lengths, *offsets, *rowmap, *seg;
    float *Xcat, *KV, *KVp, *q, *qs, *P, *ctx, *xpre, *xhat, *rstd, *z;
    size_t bytes;
};
T2VWs carve_t2v(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, N = c->N, T = c->T, d = c->d, dt = c->d / 2, R = B * N, BT = B * T;
    Carver k(base);
    T2VWs w;
    w.mask = k.take<unsigned char>(R);
    w.mtxt = k.take<unsigned char>(B);
    w.lengths = k.take<int>(B);
    w.offsets = k.take<int>(B + 1);
    w.rowmap = k.take<int>(R);
    w.seg = k.take<int>(R);
    w.Xcat = k.take<float>(R * (d + dt));
    w.KV = k.take<float>(R * d);
    w.KVp = k.take<float>(R * 2 * d);
    w.q = k.take<float>(d);
    w.qs = k.take<float>(d);
    w.P = k.take<float>(R * c->H);
    w.ctx = k.take<float>(BT * d);
    w.xpre = k.take<float>(BT * d);
    w.xhat = k.take<float>(BT * d);
    w.rstd = k.take<float>(BT);
    w.z = k.take<float>(BT * d);
    w.bytes = k.bytes();
    return w;
}
struct T2VScratch {
    float *dz, *dx, *dctx, *dKVp, *dKV, *dXcat, *dqs_part, *dqs, *dq, *dp, *red;
    size_t bytes;
};
T2VScratch carve_t2v_scratch(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, N = c->N, T = c->T, d = c->d, dt = c->d / 2, R = B * N, BT = B * T;
    Carver k(base);
    T2VScratch s;
    s.dz = k.take<float>(BT * d);
    s.dx = k.take<float>(BT * d);
    s.dctx = k.take<float>(BT * d);
    s.dKVp = k.take<float>(R * 2 * d);
    s.dKV = k.take<float>(R * d);
    s.dXcat = k.take<float>(R * (d + dt));
    s.dqs_part = k.take<float>(B * d);
    s.dqs = k.take<float>(d);
    s.dq = k.take<float>(d);
    s.dp = k.take<float>(R * c->H);
    s.red = k.take<float>(64 * (d + dt + 8));
    s.bytes = k.bytes();
    return s;
}

}  // namespace

extern "C" {

int immtsf_abi_version(void) { return IMMTSF_ABI_VERSION; }

int immtsf_ragged_index(const float* notes, int32_t B, int32_t N, int32_t d_m, uint8_t* note_mask, int32_t* lengths,
                        int32_t* offsets, int32_t* rowmap, int32_t* seg, uint8_t* m_txt, int32_t* nan_flag,
                        immtsf_stream_t stream) {
    if (!notes || !note_mask || !lengths || !offsets || !rowmap || !seg || !m_txt || B <= 0 || N < 0 || d_m <= 0)
        return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    CHECK(launch_note_mask(notes, B * N, d_m, note_mask, nan_flag, s));
    return launch_ragged_index(note_mask, B, N, lengths, offsets, rowmap, seg, m_txt, s);
}

size_t immtsf_ttf_t2v_xattn_workspace_bytes(const immtsf_fusion_cfg* cfg) { return bad_cfg(cfg) ? 0 : carve_t2v(cfg, nullptr).bytes; }
size_t immtsf_ttf_t2v_xattn_scratch_bytes(const immtsf_fusion_cfg* cfg) { return bad_cfg(cfg) ? 0 : carve_t2v_scratch(cfg, nullptr).bytes; }

// `src_rows` == null: `notes` is the zero-padded (B,N,d_m) tensor and the ragged index is derived from it (reference
// semantics).  Otherwise `notes` is the resident embedding matrix, src_rows[packed row] its row and `lengths_in` the
// per-window note counts from the batch builder: no padded tensor, no |V|-sum scan.
static int t2v_forward(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* notes, const int32_t* src_rows,
                       const int32_t* lengths_in, const float* tau, float* E_txt, uint8_t* M_txt, void* workspace,
                       size_t workspace_bytes, int32_t* nan_flag, immtsf_stream_t stream) {
    if (bad_cfg(cfg) || !p || !notes || !tau || !E_txt || !M_txt || !workspace) return IMMTSF_EINVAL;
    if (cfg->d < 4 || cfg->N <= 0 || cfg->d_m <= 0) return IMMTSF_EINVAL;
    if (!p->input_proj_w && cfg->d != cfg->d_m) return IMMTSF_EINVAL;
    T2VWs w = carve_t2v(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, N = cfg->N, T = cfg->T, d = cfg->d, dt = d / 2, dcat = d + dt, H = cfg->H, hd = d / H;
    const int R = B * N, BT = B * T, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    const int* total = w.offsets + B;

    if (src_rows) CHECK(launch_mask_from_lengths(lengths_in, B, N, w.mask, s));
    else CHECK(launch_note_mask(notes, R, cfg->d_m, w.mask, nan_flag, s));
    CHECK(launch_ragged_index(w.mask, B, N, w.lengths, w.offsets, w.rowmap, w.seg, w.mtxt, s));
    const int* gather = src_rows ? src_rows : w.rowmap;
    // [input_proj(V) ; time2vec(tau)] on the packed rows
    if (p->input_proj_w) {
        GemmArgs g = gemm_args(R, d, cfg->d_m, cfg->d_m, cfg->d_m, dcat);
        set_problem(g, 0, notes, p->input_proj_w, w.Xcat, p->input_proj_b);
        g.dyn = total; g.dyn_which = 0; g.a_rowmap = gather;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    } else {
        CHECK(launch_gather_rows(notes, cfg->d_m, gather, total, R, d, w.Xcat, dcat, s));
    }
    CHECK(launch_time2vec_fwd(tau, w.rowmap, total, R, dt, p->t2v_lin_w, p->t2v_lin_b, p->t2v_per_w, p->t2v_per_b,
                              w.Xcat + d, dcat, s));
    {   // KV = KV_proj([V;tau])
        GemmArgs g = gemm_args(R, d, dcat, dcat, dcat, d);
        set_problem(g, 0, w.Xcat, p->kv_w, w.KV, p->kv_b);
        g.dyn = total;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    {   // packed k|v in-projection (rows d..3d of attn.in_proj_weight), once per note
        GemmArgs g = gemm_args(R, 2 * d, d, d, d, 2 * d);
        set_problem(g, 0, w.KV, p->attn_in_w + (size_t)d * d, w.KVp, p->attn_in_b + d);
        g.dyn = total;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    CHECK(launch_matvec(p->attn_in_w, d, p->Q_param, p->attn_in_b, d, d, w.q, w.qs, sqrtf(1.0f / (float)hd), s));
    RaggedAttnDims dm; dm.B = B; dm.T = T; dm.H = H; dm.hd = hd; dm.N = N;
    CHECK(launch_ragged_attn_fwd(dm, w.offsets, w.rowmap, w.KVp, w.qs, w.P, w.ctx, drop, SITE_T2V_ATTN, s));
    {   // out_proj, zero the windows without notes, + Q_param residual
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem(g, 0, w.ctx, p->attn_out_w, w.xpre, p->attn_out_b);
        g.row_flag = w.mtxt; g.row_flag_div = T; g.add_vec = p->Q_param;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    CHECK(launch_layernorm_fwd(w.xpre, BT, d, p->ln_w, p->ln_b, 1e-5f, w.xhat, w.rstd, w.z, drop, SITE_T2V_OUT, s));
    {
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem(g, 0, w.z, p->proj_out_w, E_txt, p->proj_out_b);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    hipError_t e = hipMemcpyAsync(M_txt, w.mtxt, B, hipMemcpyDeviceToDevice, s);
    return e == hipSuccess ? IMMTSF_OK : (int)e;
}

int immtsf_ttf_t2v_xattn_forward(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* notes,
                                 const float* tau, float* E_txt, uint8_t* M_txt, void* workspace, size_t workspace_bytes,
                                 int32_t* nan_flag, immtsf_stream_t stream) {
    return t2v_forward(cfg, p, notes, nullptr, nullptr, tau, E_txt, M_txt, workspace, workspace_bytes, nan_flag, stream);
}

int immtsf_ttf_t2v_xattn_forward_packed(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* emb,
                                        const int32_t* src_rows, const int32_t* lengths, const float* tau, float* E_txt,
                                        uint8_t* M_txt, void* workspace, size_t workspace_bytes, immtsf_stream_t stream) {
    if (!src_rows || !lengths) return IMMTSF_EINVAL;
    return t2v_forward(cfg, p, emb, src_rows, lengths, tau, E_txt, M_txt, workspace, workspace_bytes, nullptr, stream);
}

static int t2v_backward(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* notes, const int32_t* src_rows,
                        const float* tau, const float* dE_txt, void* workspace, size_t workspace_bytes,
                        void* scratch, size_t scratch_bytes, const immtsf_t2v_params* gr,
                        immtsf_stream_t stream) {
    if (bad_cfg(cfg) || !p || !gr || !notes || !tau || !dE_txt || !workspace || !scratch) return IMMTSF_EINVAL;
    T2VWs w = carve_t2v(cfg, workspace);
    T2VScratch sc = carve_t2v_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, N = cfg->N, T = cfg->T, d = cfg->d, dt = d / 2, dcat = d + dt, H = cfg->H, hd = d / H;
    const int R = B * N, BT = B * T, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    const int* total = w.offsets + B;
    const float scale = sqrtf(1.0f / (float)hd);
    Fork fk(s);   // weight-gradient GEMMs run on the side stream, joined before returning

    {   // proj_out: dz = dE W_po ; dW_po = dE^T z ; db_po = colsum dE
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem(g, 0, dE_txt, p->proj_out_w, sc.dz, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(d, d, BT, d, d, d);
        set_problem(h, 0, dE_txt, w.z, gr->proj_out_w, nullptr, gr->proj_out_b);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    CHECK(launch_layernorm_bwd(sc.dz, BT, d, p->ln_w, w.xhat, w.rstd, sc.dx, drop, SITE_T2V_OUT, s));
    CHECK(launch_colsum2(sc.dz, w.xhat, BT, d, d, gr->ln_w, gr->ln_b, sc.red, s));
    // residual: dQ_param = sum over all (b,t) rows; then only windows with notes feed the attention branch
    CHECK(launch_colsum(sc.dx, nullptr, BT, nullptr, d, d, gr->Q_param, 0, sc.red, s));
    CHECK(launch_mask_rows(sc.dx, BT, d, w.mtxt, T, s));
    {   // out_proj
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem(g, 0, sc.dx, p->attn_out_w, sc.dctx, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(d, d, BT, d, d, d);
        set_problem(h, 0, sc.dx, w.ctx, gr->attn_out_w, nullptr, gr->attn_out_b);
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    RaggedAttnDims dm; dm.B = B; dm.T = T; dm.H = H; dm.hd = hd; dm.N = N;
    CHECK(launch_ragged_attn_bwd(dm, w.offsets, w.rowmap, w.KVp, w.qs, w.P, sc.dctx, sc.dKVp, sc.dqs_part, sc.dp, drop,
                                 SITE_T2V_ATTN, s));
    // query path: q = W_q Q_param + b_q, qs = q * scale
    CHECK(launch_colsum(sc.dqs_part, nullptr, B, nullptr, d, d, sc.dqs, 0, sc.red, s));
    CHECK(launch_axpy(sc.dqs, scale, sc.dq, d, 0, s));
    CHECK(launch_outer(sc.dq, p->Q_param, d, d, gr->attn_in_w, d, s));              // rows 0..d of in_proj_weight
    CHECK(launch_axpy(sc.dq, 1.f, gr->attn_in_b, d, 0, s));
    CHECK(launch_matvec_t(p->attn_in_w, d, sc.dq, d, d, gr->Q_param, 1, s));        // += W_q^T dq
    {   // k|v in-projection
        GemmArgs g = gemm_args(R, d, 2 * d, 2 * d, d, d);
        set_problem(g, 0, sc.dKVp, p->attn_in_w + (size_t)d * d, sc.dKV, nullptr);
        g.dyn = total; g.dyn_which = 0;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(2 * d, d, R, 2 * d, d, d);
        set_problem(h, 0, sc.dKVp, w.KV, gr->attn_in_w + (size_t)d * d, nullptr, gr->attn_in_b + d);
        h.dyn = total; h.dyn_which = 1;
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    {   // KV_proj
        GemmArgs g = gemm_args(R, dcat, d, d, dcat, dcat);
        set_problem(g, 0, sc.dKV, p->kv_w, sc.dXcat, nullptr);
        g.dyn = total; g.dyn_which = 0;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(d, dcat, R, d, dcat, dcat);
        set_problem(h, 0, sc.dKV, w.Xcat, gr->kv_w, nullptr, gr->kv_b);
        h.dyn = total; h.dyn_which = 1;
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    if (p->input_proj_w) {   // dW_in = dVp^T V(gathered) ; db_in = colsum dVp
        GemmArgs h = gemm_args(d, cfg->d_m, R, dcat, cfg->d_m, cfg->d_m);
        set_problem(h, 0, sc.dXcat, notes, gr->input_proj_w, nullptr, gr->input_proj_b);
        h.dyn = total; h.dyn_which = 1; h.b_rowmap = src_rows ? src_rows : w.rowmap;
        prezeroed(h, cfg);
        CHECK(immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork()));
    }
    CHECK(launch_time2vec_bwd(tau, w.rowmap, total, R, dt, p->t2v_per_w, p->t2v_per_b, sc.dXcat + d, dcat, gr->t2v_lin_w,
                              gr->t2v_lin_b, gr->t2v_per_w, gr->t2v_per_b, sc.red, 0, s));
    return fk.join();
}

int immtsf_ttf_t2v_xattn_backward(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* notes,
                                  const float* tau, const float* dE_txt, void* workspace, size_t workspace_bytes,
                                  void* scratch, size_t scratch_bytes, const immtsf_t2v_params* gr,
                                  immtsf_stream_t stream) {
    return t2v_backward(cfg, p, notes, nullptr, tau, dE_txt, workspace, workspace_bytes, scratch, scratch_bytes, gr, stream);
}

int immtsf_ttf_t2v_xattn_backward_packed(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* emb,
                                         const int32_t* src_rows, const float* tau, const float* dE_txt, void* workspace,
                                         size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                         const immtsf_t2v_params* gr, immtsf_stream_t stream) {
    if (!src_rows) return IMMTSF_EINVAL;
    return t2v_backward(cfg, p, emb, src_rows, tau, dE_txt, workspace, workspace_bytes, scratch, scratch_bytes, gr, stream);
}

}  // extern "C"
