// Block-level entry points of the C ABI: each function enqueues the whole forward (or backward) of one reference
// fusion module as a short sequence of HIP kernels on the caller's stream.  No allocation, no sync, no global
// state: graph-capturable.  Workspaces are carved from caller memory by `Carver` (256-byte aligned slices).
#include "../../include/immtsf.h"
#include "attn.hpp"
#include "gemm.hpp"
#include "rowops.hpp"
#include "tail.hpp"
#include "block_util.hpp"
#include "t2v_fold.hpp"
#include <stdlib.h>
#include <math.h>
#include <string.h>

namespace {

// ================================================================================================ TTF_T2V_XAttn
// bf16 dataflow (`hf`): in bf16 mode, when every reduction length is a whole number of 8-element chunks, the operands of
// the projections live in HBM as bf16 (written by their producers' epilogues) and the GEMMs run on the LDS-DMA kernel
// (gemm2.hip); tensors only row kernels read stay fp32.  Otherwise (fp32 parity mode, odd dims) every tensor is fp32 and
// the round-1 kernel converts while staging.
inline bool t2v_hf(const immtsf_fusion_cfg* c) { return c->precision == 1 && c->d >= 16 && (c->d % 16) == 0 && (c->d_m % 8) == 0; }

struct T2VWs {
    unsigned char *mask, *mtxt;
    int *lengths, *offsets, *rowmap, *seg;
    Mat Xcat, KV, ctx, z;
    void* Vh;        // bf16 image of the gathered (packed) note embeddings [R, d_m] (hf with an input projection)
    Mat KVp;                  // packed k | v in-projection: fp32, or -- bf16 dataflow -- only the bf16 image (its one reader is the ragged attention)
    float *q, *qs, *P, *xpre, *xhat, *rstd, *part;
    void *w_in, *w_kv, *w_inkv, *w_out, *w_po;     // bf16 weight images when no twin is registered (hf only)
    size_t bytes;
};
T2VWs carve_t2v(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, N = c->N, T = c->T, d = c->d, dt = c->d / 2, R = B * N, BT = B * T;
    const bool hf = t2v_hf(c);
    Carver k(base);
    T2VWs w;
    w.mask = k.take<unsigned char>(R);
    w.mtxt = k.take<unsigned char>(B);
    w.lengths = k.take<int>(B);
    w.offsets = k.take<int>(B + 1);
    w.rowmap = k.take<int>(R);
    w.seg = k.take<int>(R);
    w.Vh = hf ? k.take<unsigned short>(R * (size_t)c->d_m) : nullptr;
    w.Xcat = k.take_mat(R * (d + dt), !hf, hf);
    w.KV = k.take_mat(R * d, !hf, hf);
    w.KVp = k.take_mat(R * 2 * d, !hf, hf);
    w.q = k.take<float>(d);
    w.qs = k.take<float>(d);
    w.P = k.take<float>(R * c->H);
    w.ctx = k.take_mat(BT * d, !hf, hf);
    w.xpre = k.take<float>(BT * d);
    w.xhat = k.take<float>(BT * d);
    w.rstd = k.take<float>(BT);
    w.z = k.take_mat(BT * d, !hf, hf);
    w.part = k.take<float>(ragged_attn_part_floats(c->B, c->T, c->d, c->N));     // long windows only (0 otherwise)
    w.w_in = w.w_kv = w.w_inkv = w.w_out = w.w_po = nullptr;
    if (hf) {
        w.w_in = k.take<unsigned short>(d * (size_t)c->d_m);
        w.w_kv = k.take<unsigned short>(d * (d + dt));
        w.w_inkv = k.take<unsigned short>(2 * d * d);
        w.w_out = k.take<unsigned short>(d * d);
        w.w_po = k.take<unsigned short>(d * d);
    }
    w.bytes = k.bytes();
    return w;
}
struct T2VScratch {
    Mat dE, dx, dKVp, dKV;
    float *dz, *dctx, *dXcat, *dqs_part, *dqs, *dq, *dp, *red, *red_t2v;
    void* dXcat_h;
    int t2v_slabs;
    // split-K workspaces of the five weight-gradient GEMMs (they run concurrently on forked streams): empty below 8192 rows
    void* sk[5];
    size_t skb[5];
    size_t bytes;
};
T2VScratch carve_t2v_scratch(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, N = c->N, T = c->T, d = c->d, dt = c->d / 2, R = B * N, BT = B * T;
    const bool hf = t2v_hf(c);
    Carver k(base);
    T2VScratch s;
    s.dE = k.take_mat(BT * d, false, hf);
    s.dz = k.take<float>(BT * d);
    s.dx = k.take_mat(BT * d, true, hf);
    s.dctx = k.take<float>(BT * d);
    s.dKVp = k.take_mat(R * 2 * d, !hf, hf);
    s.dKV = k.take_mat(R * d, !hf, hf);
    s.dXcat = k.take<float>(R * (d + dt));
    s.dXcat_h = hf ? k.take<unsigned short>(R * (d + dt)) : nullptr;
    s.dqs_part = k.take<float>(B * d);
    s.dqs = k.take<float>(d);
    s.dq = k.take<float>(d);
    s.dp = k.take<float>(ragged_attn_dp_floats(c->B, c->N, c->H, c->d / c->H));
    s.red = k.take<float>(ln_sums_scratch_floats(d, 3));        // (>= colsum_scratch_floats(d, 3): the fallback's scratch)
    s.t2v_slabs = (int)(R / 256 < 32 ? 32 : (R / 256 > 1024 ? 1024 : R / 256));       // time2vec backward: ~256 packed rows per slab
    s.red_t2v = k.take<float>((size_t)s.t2v_slabs * 2 * dt);
    {
        const int di = (int)d, dc = (int)(d + dt), dm = c->d_m, Rb = (int)R, BTb = (int)BT;
        const size_t need[5] = {hf ? immtsf_gemm3_tn_ws_bytes(di, di, BTb) : 0, hf ? immtsf_gemm3_tn_ws_bytes(di, di, BTb) : 0,
                                hf ? immtsf_gemm3_tn_ws_bytes(2 * di, di, Rb) : 0, hf ? immtsf_gemm3_tn_ws_bytes(di, dc, Rb) : 0,
                                hf ? immtsf_gemm3_tn_ws_bytes(di, dm, Rb) : 0};
        for (int i = 0; i < 5; ++i) {
            s.skb[i] = need[i];
            s.sk[i] = need[i] ? k.take<unsigned char>(need[i]) : nullptr;
        }
    }
    s.bytes = k.bytes();
    return s;
}

// a batch's ragged index built ahead of the call (immtsf_fusion_cfg.note_index) replaces the workspace's own index arrays
template <typename WS> inline void use_note_index(const immtsf_fusion_cfg* c, WS& w) {
    const immtsf_note_index* ix = c->note_index;
    if (!ix) return;
    w.mask = ix->mask; w.mtxt = ix->mtxt; w.lengths = ix->lengths; w.offsets = ix->offsets; w.rowmap = ix->rowmap; w.seg = ix->seg;
}
inline bool bad_note_index(const immtsf_fusion_cfg* c) {
    const immtsf_note_index* ix = c->note_index;
    return ix && (!ix->mask || !ix->mtxt || !ix->lengths || !ix->offsets || !ix->rowmap || !ix->seg);
}

// the block's GEMM weights as (fp32, bf16) pairs
struct T2VW { Mat in, kv, inkv, out, po; };
int t2v_weights(const immtsf_fusion_cfg* c, const immtsf_t2v_params* p, const T2VWs& w, hipStream_t s, T2VW* o) {
    const bool hf = t2v_hf(c);
    const size_t d = c->d, dcat = d + d / 2;
    CHECK(weight_mat(hf, p->input_proj_w, d * (size_t)c->d_m, w.w_in, s, &o->in));
    CHECK(weight_mat(hf, p->kv_w, d * dcat, w.w_kv, s, &o->kv));
    CHECK(weight_mat(hf, p->attn_in_w + d * d, 2 * d * d, w.w_inkv, s, &o->inkv));
    CHECK(weight_mat(hf, p->attn_out_w, d * d, w.w_out, s, &o->out));
    CHECK(weight_mat(hf, p->proj_out_w, d * d, w.w_po, s, &o->po));
    return 0;
}


// ================================================================================================ TTF_T2V_XAttn, folded form
// (csrc/t2v_fold.hip has the algebra and the non-GEMM kernels; this is the launch sequence.)  Taken wherever its limits hold unless the
// caller asks for the chain as written (immtsf_fusion_cfg.form = 1, the cross-check).
// the MIX-FIRST variant of the folded form (t2v_premix.hip): windows with more notes than the fold's mix kernels hold (N > 64) in bf16
// mode, one head, T <= 32 -- or form = 3 at any N
inline bool t2v_premix_on(const immtsf_fusion_cfg* c) {
    const int form = c->form & 3;
    if (form == 1 || form == 2) return false;
    if (!(c->precision == 1 && t2v_hf(c)) || !t2v_premix_shape_ok(c->T, c->d, c->H, c->d_m)) return false;
    return form == 3 || c->N > 64;
}
inline bool t2v_fold_on(const immtsf_fusion_cfg* c) {
    if (t2v_premix_on(c)) return true;
    const int form = c->form & 3;
    if (form == 1 || !t2v_fold_shape_ok(c->N, c->T, c->d, c->H)) return false;
    if (form == 0 && (long)c->B * c->N < IMMTSF_T2V_FOLD_MIN_ROWS) return false;      // small batches: the chain's GEMMs are as cheap as the fold's fixed cost
    if ((c->d_m % 8) || (c->d % 16) || c->d_m <= 0) return false;            // X rows and the Time2Vec half in 16-byte pieces
    if (c->precision == 1 && !(t2v_hf(c) && ((c->d / c->H) % 8) == 0)) return false;
    return true;
}
struct T2VFoldWs {
    unsigned char *mask, *mtxt;
    int *lengths, *offsets, *rowmap, *seg;
    Mat X, z, zln;            // [R, dmc] notes | Time2Vec ; [R, H d] folded value rows ; [BT, d] LayerNorm output
    float *S, *P, *q, *qs, *xpre, *xhat, *rstd;
    unsigned long long* keep;      // the output dropout's keep bits (the wide mix + LayerNorm kernel writes, the low-rank LayerNorm backward reads)
    void* At;                 // mix-first: [R, 32] bf16, the dropped attention weights of (note, step)
    float *wbar, *wpart;      // mix-first: [BT] their sums; [B, chunks, 32] the sums per chunk of notes
    Mat xbar;                 // mix-first: [BT, dmc] bf16, the mix of the raw rows
    Mat OVa;                  // [H d + 8, d]: W_o[:, h] W_v[h, :] per head, then G_h = (scale q_h)^T W_k,h, then zero rows
    Mat Ab;                   // [d, dmc] = [W_KV[:, :d] W_in | W_KV[:, d:]] (with an input projection; else W_KV itself)
    Mat Wa;                   // [H d + 8, dmc]: W_tot per head, then the score vectors u_h, then zero rows
    float *bvec1, *bvec2, *cvec;
    void *w_in, *w_kv, *w_att, *w_out, *w_po;
    size_t bytes;
};
T2VFoldWs carve_t2v_fold(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, N = c->N, T = c->T, d = c->d, dt = d / 2, R = B * N, BT = B * T, H = c->H, Hd = H * d, Ma = Hd + 8;
    const size_t dmc = (size_t)c->d_m + dt;
    const bool hf = t2v_hf(c) && c->precision == 1;
    Carver k(base);
    T2VFoldWs w;
    w.mask = k.take<unsigned char>(R);
    w.mtxt = k.take<unsigned char>(B);
    w.lengths = k.take<int>(B);
    w.offsets = k.take<int>(B + 1);
    w.rowmap = k.take<int>(R);
    w.seg = k.take<int>(R);
    const bool pm = t2v_premix_on(c);
    w.X = k.take_mat(R * dmc, !hf, hf);
    w.z = pm ? Mat{nullptr, nullptr} : k.take_mat(R * Hd, !hf, hf);
    w.S = k.take<float>(R * H);
    w.P = k.take<float>(R * H);
    w.q = k.take<float>(d);
    w.qs = k.take<float>(d);
    const bool wide = t2v_mix_wide_ok((int)d) && !pm;
    w.xpre = wide ? nullptr : k.take<float>(BT * d);      // (the wide mix + LayerNorm kernel keeps x_pre in registers)
    w.xhat = k.take<float>(BT * d);
    w.rstd = k.take<float>(BT);
    w.keep = wide ? k.take<unsigned long long>(B * 2 * 256) : nullptr;
    w.At = pm ? k.take<unsigned short>(R * 32) : nullptr;
    w.wbar = pm ? k.take<float>(BT) : nullptr;
    w.wpart = pm ? k.take<float>(B * (size_t)t2v_premix_chunks((int)N) * 32) : nullptr;
    w.xbar = pm ? k.take_mat(BT * dmc, false, true) : Mat{nullptr, nullptr};
    w.zln = k.take_mat(BT * d, !hf, hf);
    w.OVa = k.take_mat(Ma * d, true, hf);
    w.Ab = k.take_mat(d * dmc, true, hf);
    w.Wa = k.take_mat(Ma * dmc, true, hf);
    w.bvec1 = k.take<float>(d);
    w.bvec2 = k.take<float>(d);
    w.cvec = k.take<float>(Hd);
    w.w_in = w.w_kv = w.w_att = w.w_out = w.w_po = nullptr;
    if (hf) {
        w.w_in = k.take<unsigned short>(d * (size_t)c->d_m);
        w.w_kv = k.take<unsigned short>(d * (d + dt));
        w.w_att = k.take<unsigned short>(3 * d * d);
        w.w_out = k.take<unsigned short>(d * d);
        w.w_po = k.take<unsigned short>(d * d);
    }
    w.bytes = k.bytes();
    return w;
}
struct T2VFoldScratch {
    Mat dE, dza, dWa, dOVa, dA;
    float *dzln, *dx, *dbo_part, *dXt, *dcv, *dbvec1, *dbvec2, *dqs, *red, *red_t2v, *red_bo;
    void* dxh;                // mix-first: bf16 image of dx, [BT, d]
    Mat dxbar;                // mix-first: [BT, dmc] bf16
    float *dwbar, *g, *ds, *du_slab;      // mix-first: [BT], [R], [R], the score vector's gradient per row block
    int t2v_slabs;
    void* sk[2];
    size_t skb[2];
    size_t bytes;
};
T2VFoldScratch carve_t2v_fold_scratch(const immtsf_fusion_cfg* c, void* base) {
    const size_t B = c->B, N = c->N, T = c->T, d = c->d, dt = d / 2, R = B * N, BT = B * T, H = c->H, Hd = H * d, Ma = Hd + 8;
    const size_t dmc = (size_t)c->d_m + dt;
    const bool hf = t2v_hf(c) && c->precision == 1;
    Carver k(base);
    T2VFoldScratch s;
    s.dE = k.take_mat(BT * d, false, hf);
    s.dzln = k.take<float>(BT * d);
    s.dx = k.take<float>(BT * d);
    const bool pm = t2v_premix_on(c);
    s.dza = pm ? Mat{nullptr, nullptr} : k.take_mat(R * Ma, !hf, hf);
    s.dxh = pm ? k.take<unsigned short>(BT * d) : nullptr;
    s.dxbar = pm ? k.take_mat(BT * dmc, false, true) : Mat{nullptr, nullptr};
    s.dwbar = pm ? k.take<float>(BT) : nullptr;
    s.g = pm ? k.take<float>(R) : nullptr;
    s.ds = pm ? k.take<float>(R) : nullptr;
    s.du_slab = pm ? k.take<float>(t2v_premix_du_scratch_floats((int)dmc)) : nullptr;
    s.dbo_part = k.take<float>(B * d);
    s.dXt = k.take<float>(R * dt);
    s.dWa = k.take_mat(Ma * dmc, true, hf);
    s.dcv = k.take<float>(Ma);
    s.dOVa = k.take_mat(Ma * d, true, hf);
    s.dA = k.take_mat(d * dmc, true, hf);
    s.dbvec1 = k.take<float>(d);
    s.dbvec2 = k.take<float>(d);
    s.dqs = k.take<float>(d);
    s.red = k.take<float>(ln_sums_scratch_floats(d, 3));
    s.red_bo = k.take<float>(colsum_scratch_floats(d, 1) + 64 * d);
    s.t2v_slabs = (int)(R / 256 < 32 ? 32 : (R / 256 > 1024 ? 1024 : R / 256));
    s.red_t2v = k.take<float>((size_t)s.t2v_slabs * 2 * dt);
    {
        const size_t need[2] = {hf ? immtsf_gemm3_tn_ws_bytes((int)d, (int)d, (int)BT) : 0,
                                (hf && !pm) ? immtsf_gemm3_tn_ws_bytes((int)Ma, (int)dmc, (int)R) : 0};
        for (int i = 0; i < 2; ++i) {
            s.skb[i] = need[i];
            s.sk[i] = need[i] ? k.take<unsigned char>(need[i]) : nullptr;
        }
    }
    s.bytes = k.bytes();
    return s;
}
// the block's weights as (fp32, bf16) pairs: in = input_proj, kv = KV_proj, att = attn.in_proj (3d, d), out, po
struct T2VFW { Mat in, kv, att, out, po; };
int t2v_fold_weights(const immtsf_fusion_cfg* c, const immtsf_t2v_params* p, const T2VFoldWs& w, hipStream_t s, T2VFW* o) {
    const bool hf = t2v_hf(c) && c->precision == 1;
    const size_t d = c->d, dcat = d + d / 2;
    CHECK(weight_mat(hf, p->input_proj_w, d * (size_t)c->d_m, w.w_in, s, &o->in));
    CHECK(weight_mat(hf, p->kv_w, d * dcat, w.w_kv, s, &o->kv));
    CHECK(weight_mat(hf, p->attn_in_w, 3 * d * d, w.w_att, s, &o->att));
    CHECK(weight_mat(hf, p->attn_out_w, d * d, w.w_out, s, &o->out));
    CHECK(weight_mat(hf, p->proj_out_w, d * d, w.w_po, s, &o->po));
    return 0;
}
inline int fold_gemm(int layout, int prec, int M, int N, int K, Mat A, int lda, Mat B, int ldb, Mat C, int ldc, const float* bias, hipStream_t s) {
    GemmArgs g = gemm_args(M, N, K, lda, ldb, ldc);
    set_problem2(g, 0, A, B, C, bias);
    return immtsf_launch_gemm(layout, prec, g, s);
}

int t2v_fold_forward(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* notes, const int32_t* src_rows,
                     const int32_t* lengths_in, const float* tau, float* E_txt, uint8_t* M_txt, void* workspace, size_t workspace_bytes,
                     int32_t* nan_flag, hipStream_t s) {
    T2VFoldWs w = carve_t2v_fold(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    const bool indexed = src_rows && cfg->note_index;
    if (indexed) use_note_index(cfg, w);
    const int B = cfg->B, N = cfg->N, T = cfg->T, d = cfg->d, dt = d / 2, dcat = d + dt, H = cfg->H, hd = d / H, d_m = cfg->d_m;
    const int R = B * N, BT = B * T, prec = cfg->precision, dmc = d_m + dt, Hd = H * d, Ma = Hd + 8;
    const bool hf = t2v_hf(cfg) && prec == 1, inp = p->input_proj_w != nullptr;
    const DropCfg drop = drop_of(cfg);
    const int* total = w.offsets + B;
    T2VFW W;
    CHECK(t2v_fold_weights(cfg, p, w, s, &W));
    if (!indexed) {
        if (src_rows) CHECK(launch_mask_from_lengths(lengths_in, B, N, w.mask, s));
        else CHECK(launch_note_mask(notes, R, d_m, w.mask, nan_flag, s));
        CHECK(launch_ragged_index(w.mask, B, N, w.lengths, w.offsets, w.rowmap, w.seg, w.mtxt, s, M_txt));
    }
    const int* gather = src_rows ? src_rows : w.rowmap;
    const float scale = sqrtf(1.0f / (float)hd);
    // X = [V ; Time2Vec(tau)] on the packed rows; in the same launch the two parameter-only mat-vecs the fold starts from: the learned
    // query's in-projection and bvec1 = W_KV[:, :d] b_in + b_KV
    const bool pm = t2v_premix_on(cfg);
    if (hf && !pm) {
        CHECK(launch_notes_stage(notes, d_m, gather, total, R, d_m, w.X.h, dmc, tau, w.rowmap, dt, p->t2v_lin_w, p->t2v_lin_b, p->t2v_per_w,
                                 p->t2v_per_b, nullptr, dmc, mat_off(w.X, d_m).h, p->attn_in_w, d, p->Q_param, p->attn_in_b, d, d, w.q, w.qs, scale, s,
                                 inp ? p->kv_w : nullptr, dcat, p->input_proj_b, p->kv_b, d, d, w.bvec1));
    } else if (pm) {
        // mix-first: the notes are staged BEHIND the fold, with the score u . x fused into the staging pass (one read of X less); the two
        // mat-vecs the fold starts from go first
        VecJobList l;
        VecJob& q = l.add(VJ_MV, p->attn_in_w, d, p->Q_param, p->attn_in_b, w.qs, d, d);
        q.scale = scale;
        if (inp) l.add(VJ_MV, p->kv_w, dcat, p->input_proj_b, p->kv_b, w.bvec1, d, d);
        CHECK(launch_vecjobs(l, s));
    } else {
        CHECK(launch_gather_rows(notes, d_m, gather, total, R, d_m, w.X.f, dmc, s, nullptr));
        CHECK(launch_time2vec_fwd(tau, w.rowmap, total, R, dt, p->t2v_lin_w, p->t2v_lin_b, p->t2v_per_w, p->t2v_per_b, w.X.f + d_m, dmc, s, nullptr));
        VecJobList l;
        VecJob& q = l.add(VJ_MV, p->attn_in_w, d, p->Q_param, p->attn_in_b, w.qs, d, d);
        q.scale = scale;
        if (inp) l.add(VJ_MV, p->kv_w, dcat, p->input_proj_b, p->kv_b, w.bvec1, d, d);
        CHECK(launch_vecjobs(l, s));
    }
    // ---- the fold (parameters only): five launches
    const Mat Wq_k = mat_off(W.att, (size_t)d * d), Wv = mat_off(W.att, (size_t)2 * d * d);
    const Mat Abuf = inp ? w.Ab : W.kv;
    const int ldab = inp ? dmc : dcat;         // (without an input projection d_m == d: the two pitches are the same number)
    const float* bvec1 = inp ? w.bvec1 : p->kv_b;
    for (int h = 0; h < H; ++h)         // OV_h = W_o[:, h] W_v[h, :]
        CHECK(fold_gemm(GEMM_NN, prec, d, d, hd, mat_off(W.out, (size_t)h * hd), d, mat_off(Wv, (size_t)h * hd * d), d, mat_off(w.OVa, (size_t)h * d * d), d, nullptr, s));
    if (inp) CHECK(fold_gemm(GEMM_NN, prec, d, d_m, d, W.kv, dcat, W.in, d_m, w.Ab, dmc, nullptr, s));
    {
        VecJobList l;
        for (int h = 0; h < H; ++h) {
            VecJob& g = l.add(VJ_MVT, Wq_k.f + (size_t)h * hd * d, d, w.qs + h * hd, nullptr, w.OVa.f + (size_t)(Hd + h) * d, hd, d);
            g.yh = hf ? mat_off(w.OVa, (size_t)(Hd + h) * d).h : nullptr;
        }
        VecJob& z2 = l.add(VJ_COPY, nullptr, 0, nullptr, nullptr, w.OVa.f + (size_t)(Hd + H) * d, 8 - H, d);       // the pad rows
        z2.yh = hf ? mat_off(w.OVa, (size_t)(Hd + H) * d).h : nullptr; z2.ldy = d;
        l.add(VJ_MV, Wv.f, d, bvec1, p->attn_in_b + 2 * d, w.bvec2, d, d);
        if (inp) {
            VecJob& c = l.add(VJ_COPY, p->kv_w + d, dcat, nullptr, nullptr, w.Ab.f + d_m, d, dt);
            c.yh = hf ? mat_off(w.Ab, d_m).h : nullptr; c.ldy = dmc;
        }
        CHECK(launch_vecjobs(l, s));
    }
    CHECK(fold_gemm(GEMM_NN, prec, Hd, dmc, d, w.OVa, d, Abuf, ldab, w.Wa, dmc, nullptr, s));
    {
        VecJobList l;
        for (int h = 0; h < H; ++h) {
            VecJob& u = l.add(VJ_MVT, Abuf.f, ldab, w.OVa.f + (size_t)(Hd + h) * d, nullptr, w.Wa.f + (size_t)(Hd + h) * dmc, d, dmc);
            u.yh = hf ? mat_off(w.Wa, (size_t)(Hd + h) * dmc).h : nullptr;
            l.add(VJ_MV, p->attn_out_w + h * hd, d, w.bvec2 + h * hd, nullptr, w.cvec + h * d, d, hd);
        }
        VecJob& z1 = l.add(VJ_COPY, nullptr, 0, nullptr, nullptr, w.Wa.f + (size_t)(Hd + H) * dmc, 8 - H, dmc);     // the pad rows
        z1.yh = hf ? mat_off(w.Wa, (size_t)(Hd + H) * dmc).h : nullptr; z1.ldy = dmc;
        CHECK(launch_vecjobs(l, s));
    }
    // ---- the data path
    if (pm) {   // mix first (raw rows), then ONE B T-row product with W_tot
        const int rc = launch_notes_stage(notes, d_m, gather, total, R, d_m, w.X.h, dmc, tau, w.rowmap, dt, p->t2v_lin_w, p->t2v_lin_b, p->t2v_per_w,
                                          p->t2v_per_b, nullptr, dmc, mat_off(w.X, d_m).h, nullptr, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, 1.f, s,
                                          nullptr, 0, nullptr, nullptr, 0, 0, nullptr, w.Wa.f + (size_t)Hd * dmc, w.S);
        if (rc == IMMTSF_EUNSUPPORTED) {   // (rows that are not 16-byte aligned: staging and score as two passes)
            CHECK(launch_notes_stage(notes, d_m, gather, total, R, d_m, w.X.h, dmc, tau, w.rowmap, dt, p->t2v_lin_w, p->t2v_lin_b, p->t2v_per_w,
                                     p->t2v_per_b, nullptr, dmc, mat_off(w.X, d_m).h, nullptr, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, 1.f, s));
            CHECK(launch_t2v_scores(w.X.h, 1, dmc, w.Wa.f + (size_t)Hd * dmc, dmc, H, total, R, w.S, s));
        } else {
            CHECK(rc);
        }
        CHECK(launch_t2v_premix_weights(B, T, N, w.offsets, w.rowmap, w.S, w.P, w.At, w.wpart, drop, SITE_T2V_ATTN, s));
        CHECK(launch_t2v_premix_fwd(B, T, dmc, w.offsets, w.X.h, w.At, w.xbar.h, s));
        CHECK(fold_gemm(GEMM_NT, prec, BT, d, dmc, w.xbar, dmc, w.Wa, dmc, mat(w.xpre), d, nullptr, s));
        CHECK(launch_t2v_premix_finish(BT, T, N, d, w.xpre, p->attn_out_b, w.cvec, w.wpart, w.wbar, p->Q_param, w.mtxt, s));
    } else {
        CHECK(launch_t2v_scores(hf ? w.X.h : (const void*)w.X.f, hf ? 1 : 0, dmc, w.Wa.f + (size_t)Hd * dmc, dmc, H, total, R, w.S, s));
        GemmArgs g = gemm_args(R, Hd, dmc, dmc, dmc, Hd);
        set_problem2(g, 0, w.X, w.Wa, w.z, w.cvec);
        g.dyn = total; g.dyn_which = 0;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    T2VFoldDims dm; dm.B = B; dm.T = T; dm.H = H; dm.d = d; dm.N = N; dm.dmc = dmc;
    // bf16 dataflow: x_hat's only reader is the LayerNorm backward with sums, which widens on load -- it is stored as bf16 alone (in the
    // fp32 image's place) wherever that kernel will take the shape (the backward makes the same test)
    const bool compact = hf && ln_sums_compact_ok(BT, d);
    float* xhat_f = compact ? nullptr : w.xhat;
    void* xhat_h = compact ? static_cast<void*>(w.xhat) : nullptr;
    const bool noproj = (cfg->form & IMMTSF_FORM_NO_PROJ) != 0;
    // what leaves the LayerNorm: Z itself to the caller (proj_out is the consumer's: immtsf_mmf_xrank_p_forward_z) -- fp32 unless the
    // caller reads the bf16 image only (IMMTSF_FORM_HALF_OUT) -- or the operand of proj_out
    void* z_h = noproj ? (hf ? (cfg->out_h ? cfg->out_h : w.zln.h) : nullptr) : w.zln.h;
    float* z_f = noproj ? (((cfg->form & IMMTSF_FORM_HALF_OUT) && hf && cfg->out_h) ? nullptr : E_txt) : w.zln.f;
    if (pm) {
        const DropCfg nodrop = DropCfg{0, 0.f, 1.f, nullptr};
        CHECK(launch_layernorm_fwd(w.xpre, BT, d, p->ln_w, p->ln_b, 1e-5f, xhat_f, w.rstd, z_f, drop, SITE_T2V_OUT, s, z_h, nullptr, nodrop, 0, xhat_h));
    } else if (t2v_mix_wide_ok(d)) {
        CHECK(launch_t2v_mix_ln_fwd(dm, w.offsets, w.rowmap, w.S, hf ? w.z.h : (const void*)w.z.f, hf ? 1 : 0, p->attn_out_b, p->Q_param, w.P,
                                    p->ln_w, p->ln_b, 1e-5f, xhat_f, xhat_h, w.rstd, z_f, z_h, drop, SITE_T2V_ATTN, drop, SITE_T2V_OUT, s, w.keep));
    } else {
        CHECK(launch_t2v_mix_fwd(dm, w.offsets, w.rowmap, w.S, hf ? w.z.h : (const void*)w.z.f, hf ? 1 : 0, p->attn_out_b, p->Q_param, w.P, w.xpre,
                                 drop, SITE_T2V_ATTN, s));
        const DropCfg nodrop = DropCfg{0, 0.f, 1.f, nullptr};
        CHECK(launch_layernorm_fwd(w.xpre, BT, d, p->ln_w, p->ln_b, 1e-5f, xhat_f, w.rstd, z_f, drop, SITE_T2V_OUT, s, z_h, nullptr, nodrop, 0, xhat_h));
    }
    if (noproj) return IMMTSF_OK;
    {
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem2(g, 0, w.zln, W.po, mat(E_txt, hf ? cfg->out_h : nullptr), p->proj_out_b);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    (void)Ma;
    return IMMTSF_OK;
}

int t2v_fold_backward(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* tau, const float* dE_txt, void* workspace,
                      size_t workspace_bytes, void* scratch, size_t scratch_bytes, const immtsf_t2v_params* gr, hipStream_t s) {
    T2VFoldWs w = carve_t2v_fold(cfg, workspace);
    T2VFoldScratch sc = carve_t2v_fold_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    use_note_index(cfg, w);
    const int B = cfg->B, N = cfg->N, T = cfg->T, d = cfg->d, dt = d / 2, dcat = d + dt, H = cfg->H, hd = d / H, d_m = cfg->d_m;
    const int R = B * N, BT = B * T, prec = cfg->precision, dmc = d_m + dt, Hd = H * d, Ma = Hd + 8;
    const bool hf = t2v_hf(cfg) && prec == 1, inp = p->input_proj_w != nullptr;
    const DropCfg drop = drop_of(cfg);
    const int* total = w.offsets + B;
    const float scale = sqrtf(1.0f / (float)hd);
    T2VFW W;
    CHECK(t2v_fold_weights(cfg, p, w, s, &W));
    const Mat Wq_k = mat_off(W.att, (size_t)d * d), Wv = mat_off(W.att, (size_t)2 * d * d);
    const Mat Abuf = inp ? w.Ab : W.kv;
    const int ldab = inp ? dmc : dcat;
    const float* bvec1 = inp ? w.bvec1 : p->kv_b;
    GemmArgs wg[2];
    int nwg = 0;
    const bool noproj = (cfg->form & IMMTSF_FORM_NO_PROJ) != 0;
    float* dzln = sc.dzln;
    if (noproj) {        // the incoming gradient IS dZ (the consumer owns proj_out): LayerNorm's backward works on it in place
        dzln = const_cast<float*>(dE_txt);
    } else {
        Mat dE = cmat(dE_txt);
        if (hf && cfg->in_h) {
            dE.h = const_cast<void*>(cfg->in_h);
        } else if (hf) {
            CHECK(launch_f32_to_bf16(dE_txt, sc.dE.h, (size_t)BT * d, s));
            dE.h = sc.dE.h;
        }
        // proj_out: dzln = dE W_po ; dW_po = dE^T zln ; db_po = colsum dE
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem2(g, 0, dE, W.po, mat(sc.dzln), nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        GemmArgs h = gemm_args(d, d, BT, d, d, d);
        set_problem2(h, 0, dE, w.zln, mat(gr->proj_out_w), nullptr, gr->proj_out_b);
        prezeroed(h, cfg);
        h.ws = sc.sk[0]; h.ws_bytes = sc.skb[0];
        wg[nwg++] = h;
    }
    // compact: the forward stored x_hat as bf16 alone.  dx: as bf16 alone for the WIDE mix backward (8-byte loads; the narrow kernel's
    // two-byte loads made it slower on a bf16 dx -- 421 us instead of 291 at 4096 windows -- and keeps the fp32 one)
    const bool compact = hf && ln_sums_compact_ok(BT, d);
    const bool pm = t2v_premix_on(cfg);
    const bool dx_half = compact && t2v_mix_wide_ok(d) && t2v_mix_bwd_wide && !pm;
    float* dx_f = dx_half ? nullptr : sc.dx;
    void* dx_h = dx_half ? static_cast<void*>(sc.dx) : nullptr;
    const immtsf_lowrank_grad* lr = cfg->lr_grad;
    if (lr) {   // the upstream gradient is dZ = coef basis: formed inside the LayerNorm backward, never written
        if (!noproj || !lr->coef || !lr->basis || lr->rank <= 0 || lr->ld < lr->rank || !compact) return IMMTSF_EINVAL;
        const int rc = launch_layernorm_bwd_lr(lr->coef, lr->ld, lr->rank, lr->basis, BT, d, p->ln_w, nullptr, w.xhat, w.rstd, dx_f, dx_h, drop,
                                               SITE_T2V_OUT, gr->ln_w, gr->ln_b, gr->Q_param, sc.red, w.mtxt, T, s,
                                               (t2v_mix_wide_ok(d) && !pm) ? w.keep : nullptr, T);
        if (rc != IMMTSF_OK) return rc == IMMTSF_EUNSUPPORTED ? IMMTSF_EINVAL : rc;      // (immtsf_ttf_t2v_xattn_accepts_lowrank said otherwise)
    } else {   // LayerNorm backward + its parameter gradients + dQ_param = sum of dx over ALL rows; rows of windows without notes zeroed after
        const int rc = compact ? launch_layernorm_bwd_sums(dzln, BT, d, p->ln_w, nullptr, w.rstd, dx_f, drop, SITE_T2V_OUT, gr->ln_w, gr->ln_b,
                                                           gr->Q_param, sc.red, w.mtxt, T, dx_h, s, w.xhat)
                               : launch_layernorm_bwd_sums(dzln, BT, d, p->ln_w, w.xhat, w.rstd, sc.dx, drop, SITE_T2V_OUT, gr->ln_w, gr->ln_b,
                                                           gr->Q_param, sc.red, w.mtxt, T, nullptr, s);
        if (compact && rc != IMMTSF_OK) return rc == IMMTSF_EUNSUPPORTED ? IMMTSF_EINVAL : rc;      // (the forward's test promised this path)
        if (rc == IMMTSF_EUNSUPPORTED) {
            CHECK(launch_layernorm_bwd(dzln, BT, d, p->ln_w, w.xhat, w.rstd, sc.dx, drop, SITE_T2V_OUT, s));
            CHECK(launch_colsum3(dzln, w.xhat, sc.dx, BT, d, d, gr->ln_w, gr->ln_b, gr->Q_param, sc.red, w.mtxt, T, nullptr, s, true));
        } else {
            CHECK(rc);
        }
    }
    T2VFoldDims dm; dm.B = B; dm.T = T; dm.H = H; dm.d = d; dm.N = N; dm.dmc = dmc;
    if (pm) {
        // mix-first: everything with d in it on B T rows -- dc = dx^T wbar, dwbar = dx c, dxbar = dx W_tot, dW_tot = dx^T xbar -- then the
        // per-note half (t2v_premix.hip): g, ds, the Time2Vec columns' gradient, du
        {
            VecJobList l;
            l.add(VJ_MVT, sc.dx, d, w.wbar, nullptr, sc.dcv, BT, d);
            l.add(VJ_MV, sc.dx, d, w.cvec, nullptr, sc.dwbar, BT, d);
            CHECK(launch_vecjobs(l, s));
        }
        CHECK(launch_f32_to_bf16(sc.dx, sc.dxh, (size_t)BT * d, s));
        const Mat dxm = mat(sc.dx, sc.dxh);
        CHECK(fold_gemm(GEMM_NN, prec, BT, dmc, d, dxm, d, w.Wa, dmc, sc.dxbar, dmc, nullptr, s));
        if (cfg->sched_flag) CHECK(immtsf_flag_set(cfg->sched_flag, s));
        CHECK(fold_gemm(GEMM_TN, prec, d, dmc, BT, dxm, d, w.xbar, dmc, mat(sc.dWa.f), dmc, nullptr, s));
        { const hipError_t e = hipMemsetAsync(sc.dWa.f + (size_t)Hd * dmc, 0, (size_t)8 * dmc * sizeof(float), s); if (e != hipSuccess) return (int)e; }
        CHECK(launch_t2v_premix_bwd(B, T, N, dmc, d_m, w.offsets, total, w.X.h, w.At, w.P, sc.dxbar.h, sc.dwbar, w.Wa.f + (size_t)Hd * dmc, sc.g,
                                    sc.ds, sc.dXt, sc.dWa.f + (size_t)Hd * dmc, sc.du_slab, s));
        CHECK(launch_time2vec_bwd(tau, w.rowmap, total, R, dt, p->t2v_per_w, p->t2v_per_b, sc.dXt, dt, gr->t2v_lin_w, gr->t2v_lin_b, gr->t2v_per_w,
                                  gr->t2v_per_b, sc.red_t2v, sc.t2v_slabs, s));
        CHECK(immtsf_launch_gemm_tn_list(prec, wg, nwg, s));
    } else {
    CHECK(launch_t2v_mix_bwd(dm, w.offsets, w.rowmap, w.P, hf ? w.z.h : (const void*)w.z.f, hf ? 1 : 0, dx_half ? dx_h : (const void*)sc.dx,
                             dx_half ? 1 : 0, hf ? sc.dza.h : (void*)sc.dza.f, sc.dbo_part, drop, SITE_T2V_ATTN, s));
    // (the row-bound kernels of the block are behind us: see the header.  Measured at 4096 windows: no hint 4.75 ms, here 4.63, in front
    // of the mix 4.96, behind the last weight-gradient GEMM 5.30)
    if (cfg->sched_flag) CHECK(immtsf_flag_set(cfg->sched_flag, s));

    {   // Time2Vec rows: dX_tau = dz_aug W_aug[:, d_m:]  (the score path rides in the augmented column)
        GemmArgs g = gemm_args(R, dt, Ma, Ma, dmc, dt);
        set_problem2(g, 0, sc.dza, mat_off(w.Wa, d_m), mat(sc.dXt), nullptr);
        g.dyn = total; g.dyn_which = 0;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    CHECK(launch_time2vec_bwd(tau, w.rowmap, total, R, dt, p->t2v_per_w, p->t2v_per_b, sc.dXt, dt, gr->t2v_lin_w, gr->t2v_lin_b, gr->t2v_per_w,
                              gr->t2v_per_b, sc.red_t2v, sc.t2v_slabs, s));
    {   // dW_aug = dz_aug^T X (+ dc = column sums of dz_aug)
        GemmArgs h = gemm_args(Ma, dmc, R, Ma, dmc, dmc);
        set_problem2(h, 0, sc.dza, w.X, mat(sc.dWa.f), nullptr, sc.dcv);
        h.dyn = total; h.dyn_which = 1;
        h.ws = sc.sk[1]; h.ws_bytes = sc.skb[1];
        wg[nwg++] = h;
    }
    CHECK(immtsf_launch_gemm_tn_list(prec, wg, nwg, s));
    }
    // ---- chain rule through the fold (parameters only): vector jobs, two products, vector jobs, 2 H + 2 products, one last launch
    {
        VecJobList l;
        for (int h = 0; h < H; ++h) l.add(VJ_MVT, p->attn_out_w + h * hd, d, sc.dcv + h * d, nullptr, sc.dbvec2 + h * hd, d, hd);
        // d b_o = sum over the windows: as a job of this launch for a few hundred windows; beyond that the job's 12 workgroups walked
        // thousands of rows each (92 us at 4096 windows, the longest launch of the parameter chain) -- the slabbed column sum instead
        // (mix-first: straight from dx, whose rows of windows without notes are zero)
        const float* bo_src = pm ? sc.dx : sc.dbo_part;
        const int bo_rows = pm ? BT : B;
        if (bo_rows < 512) l.add(VJ_MVT, bo_src, d, nullptr, nullptr, gr->attn_out_b, bo_rows, d);
        else CHECK(launch_colsum(bo_src, nullptr, bo_rows, nullptr, d, d, gr->attn_out_b, 0, sc.red_bo, s, true));
        if (hf) {
            VecJob& c = l.add(VJ_COPY, sc.dWa.f, dmc, nullptr, nullptr, nullptr, Ma, dmc);      // bf16 image of dW_aug for the two products
            c.yh = sc.dWa.h; c.ldy = dmc;
        }
        CHECK(launch_vecjobs(l, s));
    }
    CHECK(fold_gemm(GEMM_NT, prec, Ma, d, dmc, sc.dWa, dmc, Abuf, ldab, sc.dOVa, d, nullptr, s));          // rows H d ..: dG_h
    CHECK(fold_gemm(GEMM_TN, prec, d, dmc, Ma, w.OVa, d, sc.dWa, dmc, inp ? sc.dA : mat(gr->kv_w), inp ? dmc : dcat, nullptr, s));
    {
        VecJobList l;
        VecJob& q = l.add(VJ_MV, Wq_k.f, d, sc.dOVa.f + (size_t)Hd * d, nullptr, sc.dqs, d, d);
        q.xdiv = hd; q.xld = d;
        l.add(VJ_MVT, Wv.f, d, sc.dbvec2, nullptr, sc.dbvec1, d, d);
        VecJob& bv = l.add(VJ_COPY, sc.dbvec2, d, nullptr, nullptr, gr->attn_in_b + 2 * d, 1, d);
        bv.ldy = d;
        VecJob& bk = l.add(VJ_COPY, nullptr, 0, nullptr, nullptr, gr->attn_in_b + d, 1, d);
        bk.ldy = d;
        CHECK(launch_vecjobs(l, s));
    }
    for (int h = 0; h < H; ++h) {
        CHECK(fold_gemm(GEMM_NT, prec, d, hd, d, mat_off(sc.dOVa, (size_t)h * d * d), d, mat_off(Wv, (size_t)h * hd * d), d,
                        mat(gr->attn_out_w + h * hd), d, nullptr, s));
        CHECK(fold_gemm(GEMM_TN, prec, hd, d, d, mat_off(W.out, (size_t)h * hd), d, mat_off(sc.dOVa, (size_t)h * d * d), d,
                        mat(gr->attn_in_w + (size_t)(2 * d + h * hd) * d), d, nullptr, s));
    }
    if (inp) {
        CHECK(fold_gemm(GEMM_NT, prec, d, d, d_m, sc.dA, dmc, W.in, d_m, mat(gr->kv_w), dcat, nullptr, s));
        CHECK(fold_gemm(GEMM_TN, prec, d, d_m, d, W.kv, dcat, sc.dA, dmc, mat(gr->input_proj_w), d_m, nullptr, s));
    }
    {   // the rank-1 terms on top of the products, the remaining bias gradients, the learned query's backward (dq = scale dqs:
        // dW_q = dq Q^T, db_q = dq, dQ += W_q^T dq)
        VecJobList l;
        if (inp) l.add(VJ_MVT, p->kv_w, dcat, sc.dbvec1, nullptr, gr->input_proj_b, d, d);
        VecJob& bk = l.add(VJ_COPY, sc.dbvec1, d, nullptr, nullptr, gr->kv_b, 1, d);
        bk.ldy = d;
        l.rank1(gr->attn_in_w, d, d, d, 0, sc.dqs, p->Q_param).scale = scale;
        VecJob& bq = l.add(VJ_COPY, sc.dqs, d, nullptr, nullptr, gr->attn_in_b, 1, d);
        bq.ldy = d; bq.scale = scale;
        VecJob& dq = l.add(VJ_MVT, p->attn_in_w, d, sc.dqs, nullptr, gr->Q_param, d, d);
        dq.scale = scale; dq.acc = 1;
        for (int h = 0; h < H; ++h) {
            l.rank1(gr->attn_out_w + h * hd, d, d, hd, 1, sc.dcv + h * d, w.bvec2 + h * hd);
            l.rank1(gr->attn_in_w + (size_t)(d + h * hd) * d, d, hd, d, 0, w.qs + h * hd, sc.dOVa.f + (size_t)(Hd + h) * d);
        }
        l.rank1(gr->attn_in_w + (size_t)2 * d * d, d, d, d, 1, sc.dbvec2, bvec1);
        if (inp) {
            l.rank1(gr->kv_w, dcat, d, d, 1, sc.dbvec1, p->input_proj_b);
            l.rank1(gr->kv_w + d, dcat, d, dt, 2, nullptr, nullptr, sc.dA.f + d_m, dmc);
        }
        CHECK(launch_vecjobs(l, s));
    }
    return IMMTSF_OK;
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

int immtsf_note_index_build(const int32_t* lengths_in, int32_t B, int32_t N, const immtsf_note_index* out, immtsf_stream_t stream) {
    if (!lengths_in || !out || B <= 0 || N <= 0 || !out->mask || !out->mtxt || !out->lengths || !out->offsets || !out->rowmap || !out->seg)
        return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    CHECK(launch_mask_from_lengths(lengths_in, B, N, out->mask, s));
    return launch_ragged_index(out->mask, B, N, out->lengths, out->offsets, out->rowmap, out->seg, out->mtxt, s);
}

size_t immtsf_ttf_t2v_xattn_workspace_bytes(const immtsf_fusion_cfg* cfg) {
    if (bad_cfg(cfg)) return 0;
    return t2v_fold_on(cfg) ? carve_t2v_fold(cfg, nullptr).bytes : carve_t2v(cfg, nullptr).bytes;
}
size_t immtsf_ttf_t2v_xattn_scratch_bytes(const immtsf_fusion_cfg* cfg) {
    if (bad_cfg(cfg)) return 0;
    return t2v_fold_on(cfg) ? carve_t2v_fold_scratch(cfg, nullptr).bytes : carve_t2v_scratch(cfg, nullptr).bytes;
}
int immtsf_ttf_t2v_xattn_folded(const immtsf_fusion_cfg* cfg) { return (!bad_cfg(cfg) && t2v_fold_on(cfg)) ? 1 : 0; }

int immtsf_ttf_t2v_xattn_accepts_lowrank(const immtsf_fusion_cfg* cfg, int32_t rank) {
    if (bad_cfg(cfg) || !(cfg->form & IMMTSF_FORM_NO_PROJ)) return 0;
    const int BT = cfg->B * cfg->T, d = cfg->d;
    if (!ln_lr_ok(BT, d, rank)) return 0;
    if (t2v_fold_on(cfg)) return (t2v_hf(cfg) && cfg->precision == 1 && ln_sums_compact_ok(BT, d)) ? 1 : 0;
    return 1;
}

// `src_rows` == null: `notes` is the zero-padded (B,N,d_m) tensor and the ragged index is derived from it (reference
// semantics).  Otherwise `notes` is the resident embedding matrix, src_rows[packed row] its row and `lengths_in` the
// per-window note counts from the batch builder: no padded tensor, no |V|-sum scan.
static int t2v_forward(const immtsf_fusion_cfg* cfg, const immtsf_t2v_params* p, const float* notes, const int32_t* src_rows,
                       const int32_t* lengths_in, const float* tau, float* E_txt, uint8_t* M_txt, void* workspace,
                       size_t workspace_bytes, int32_t* nan_flag, immtsf_stream_t stream) {
    if (bad_cfg(cfg) || !p || !notes || !tau || !E_txt || !workspace || bad_note_index(cfg)) return IMMTSF_EINVAL;
    if (!M_txt && !(src_rows && cfg->note_index)) return IMMTSF_EINVAL;
    if (cfg->note_index && !src_rows) return IMMTSF_EINVAL;         // (an index comes with packed notes only)
    if (cfg->d < 4 || cfg->N <= 0 || cfg->d_m <= 0) return IMMTSF_EINVAL;
    if (!p->input_proj_w && cfg->d != cfg->d_m) return IMMTSF_EINVAL;
    if (t2v_fold_on(cfg))
        return t2v_fold_forward(cfg, p, notes, src_rows, lengths_in, tau, E_txt, M_txt, workspace, workspace_bytes, nan_flag,
                                static_cast<hipStream_t>(stream));
    T2VWs w = carve_t2v(cfg, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    const bool indexed = src_rows && cfg->note_index;
    if (indexed) use_note_index(cfg, w);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = cfg->B, N = cfg->N, T = cfg->T, d = cfg->d, dt = d / 2, dcat = d + dt, H = cfg->H, hd = d / H;
    const int R = B * N, BT = B * T, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    const int* total = w.offsets + B;

    const bool hf = t2v_hf(cfg);
    T2VW W;
    CHECK(t2v_weights(cfg, p, w, s, &W));
    if (!indexed) {
        if (src_rows) CHECK(launch_mask_from_lengths(lengths_in, B, N, w.mask, s));
        else CHECK(launch_note_mask(notes, R, cfg->d_m, w.mask, nan_flag, s));
        CHECK(launch_ragged_index(w.mask, B, N, w.lengths, w.offsets, w.rowmap, w.seg, w.mtxt, s, M_txt));      // M_txt written here: no copy at the end
    }
    const int* gather = src_rows ? src_rows : w.rowmap;
    // [input_proj(V) ; time2vec(tau)] on the packed rows.  The note embeddings are fp32 in memory (gathered rows of the
    // padded tensor or of the resident matrix): this one GEMM converts while staging and emits the bf16 image directly
    constexpr int notes_image = 1;
    bool staged = false;       // Time2Vec and the query projection already done by the notes-stage launch
    if (p->input_proj_w && hf && notes_image) {
        // bf16 mode: ONE gather + cast pass writes the packed bf16 image of the notes; the projection (here) and its weight
        // gradient (backward) then run on the bf16-in-memory GEMM instead of re-reading the fp32 rows through a row map
        // (cfg5: 145 k x 4096 rows -- the row-mapped fp32 weight gradient was the slowest kernel of the fusion path, 6.4 ms)
        // (with Time2Vec of the time stamps and the learned query's in-projection in the same launch: three independent row jobs)
        CHECK(launch_notes_stage(notes, cfg->d_m, gather, total, R, cfg->d_m, w.Vh, cfg->d_m, tau, w.rowmap, dt, p->t2v_lin_w, p->t2v_lin_b,
                                 p->t2v_per_w, p->t2v_per_b, w.Xcat.f ? w.Xcat.f + d : nullptr, dcat, w.Xcat.h ? mat_off(w.Xcat, d).h : nullptr,
                                 p->attn_in_w, d, p->Q_param, p->attn_in_b, d, d, w.q, w.qs, sqrtf(1.0f / (float)hd), s));
        staged = true;
        GemmArgs g = gemm_args(R, d, cfg->d_m, cfg->d_m, cfg->d_m, dcat);
        set_problem2(g, 0, mat(nullptr, w.Vh), W.in, w.Xcat, p->input_proj_b);
        g.dyn = total; g.dyn_which = 0;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    } else if (p->input_proj_w) {
        GemmArgs g = gemm_args(R, d, cfg->d_m, cfg->d_m, cfg->d_m, dcat);
        set_problem2(g, 0, cmat(notes), W.in, w.Xcat, p->input_proj_b);
        g.dyn = total; g.dyn_which = 0; g.a_rowmap = gather;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    } else {
        CHECK(launch_gather_rows(notes, cfg->d_m, gather, total, R, d, w.Xcat.f, dcat, s, w.Xcat.h));
    }
    if (!staged)
        CHECK(launch_time2vec_fwd(tau, w.rowmap, total, R, dt, p->t2v_lin_w, p->t2v_lin_b, p->t2v_per_w, p->t2v_per_b,
                                  w.Xcat.f ? w.Xcat.f + d : nullptr, dcat, s, w.Xcat.h ? mat_off(w.Xcat, d).h : nullptr));
    {   // KV = KV_proj([V;tau])
        GemmArgs g = gemm_args(R, d, dcat, dcat, dcat, d);
        set_problem2(g, 0, w.Xcat, W.kv, w.KV, p->kv_b);
        g.dyn = total;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    {   // packed k|v in-projection (rows d..3d of attn.in_proj_weight), once per note
        GemmArgs g = gemm_args(R, 2 * d, d, d, d, 2 * d);
        set_problem2(g, 0, w.KV, W.inkv, w.KVp, p->attn_in_b + d);
        g.dyn = total;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    if (!staged) CHECK(launch_matvec(p->attn_in_w, d, p->Q_param, p->attn_in_b, d, d, w.q, w.qs, sqrtf(1.0f / (float)hd), s));
    RaggedAttnDims dm; dm.B = B; dm.T = T; dm.H = H; dm.hd = hd; dm.N = N;
    CHECK(launch_ragged_attn_fwd(dm, w.offsets, w.rowmap, w.KVp.f, w.qs, w.P, w.ctx.f, drop, SITE_T2V_ATTN, s, w.ctx.h, w.part, w.KVp.h));
    {   // out_proj, zero the windows without notes, + Q_param residual
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem2(g, 0, w.ctx, W.out, mat(w.xpre), p->attn_out_b);
        g.row_flag = w.mtxt; g.row_flag_div = T; g.add_vec = p->Q_param;
        g.row_flag32 = w.lengths;         // same zero pattern as M_txt
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    if (cfg->form & IMMTSF_FORM_NO_PROJ)         // proj_out is the consumer's: hand over Z itself (fp32 unless the caller reads the bf16 image only)
        return launch_layernorm_fwd(w.xpre, BT, d, p->ln_w, p->ln_b, 1e-5f, w.xhat, w.rstd,
                                    ((cfg->form & IMMTSF_FORM_HALF_OUT) && hf && cfg->out_h) ? nullptr : E_txt, drop, SITE_T2V_OUT, s,
                                    hf ? (cfg->out_h ? cfg->out_h : w.z.h) : nullptr);
    CHECK(launch_layernorm_fwd(w.xpre, BT, d, p->ln_w, p->ln_b, 1e-5f, w.xhat, w.rstd, w.z.f, drop, SITE_T2V_OUT, s, w.z.h));
    {
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem2(g, 0, w.z, W.po, mat(E_txt, hf ? cfg->out_h : nullptr), p->proj_out_b);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    return IMMTSF_OK;
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
    if (bad_cfg(cfg) || !p || !gr || !notes || !tau || !dE_txt || !workspace || !scratch || bad_note_index(cfg)) return IMMTSF_EINVAL;
    if (cfg->note_index && !src_rows) return IMMTSF_EINVAL;
    // phases (immtsf_fusion_cfg.bwd_phase): 0 = everything, every weight gradient in ONE grouped launch at the end (the best form on one
    // stream); a mask = the data paths (bits 0..2) and parameter gradients (bits 4..6) of phases A, B, C this call runs -- the products
    // of one call leave as one grouped launch
    if (cfg->bwd_phase < 0 || (cfg->bwd_phase & ~0x77) || cfg->reserved0 != 0) return IMMTSF_EINVAL;
    const int ph = cfg->bwd_phase ? cfg->bwd_phase : 0x77;
    if (t2v_fold_on(cfg)) {
        // the folded form's parameter gradients all come out of its chain rule at the end: a phased caller gets the work with phase C
        if (!(ph & IMMTSF_BWD_PHASE_C)) return IMMTSF_OK;
        return t2v_fold_backward(cfg, p, tau, dE_txt, workspace, workspace_bytes, scratch, scratch_bytes, gr, static_cast<hipStream_t>(stream));
    }
    T2VWs w = carve_t2v(cfg, workspace);
    T2VScratch sc = carve_t2v_scratch(cfg, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    if (src_rows) use_note_index(cfg, w);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool phA = ph & IMMTSF_BWD_PHASE_A, phB = ph & IMMTSF_BWD_PHASE_B, phC = ph & IMMTSF_BWD_PHASE_C;
    const bool wgA = ph & IMMTSF_BWD_WGRAD_A, wgB = ph & IMMTSF_BWD_WGRAD_B, wgC = ph & IMMTSF_BWD_WGRAD_C;
    // (the chain as written has no small-launch tail to share the chip with: whoever waits for the hint goes ahead at once)
    if (cfg->sched_flag && phA) CHECK(immtsf_flag_set(cfg->sched_flag, s));
    const int B = cfg->B, N = cfg->N, T = cfg->T, d = cfg->d, dt = d / 2, dcat = d + dt, H = cfg->H, hd = d / H;
    const int R = B * N, BT = B * T, prec = cfg->precision;
    const DropCfg drop = drop_of(cfg);
    const int* total = w.offsets + B;
    const float scale = sqrtf(1.0f / (float)hd);
    Fork fk(s);   // (A/B option) weight-gradient GEMMs on the side stream, joined before returning
    // The weight gradients wait on nothing but their dY and nothing in this call waits on them: with IMMTSF_GEMM_GROUP=1 they are
    // collected and go out as ONE grouped launch at the end (gemm2_group_kernel)
    GemmArgs wg[6];
    int nwg = 0;
    const bool defer_wg = !fk.forking() && immtsf_gemm_group_enabled();       // (off by default: see gemm2.hip)
    auto wgrad = [&](GemmArgs& h) -> int {
        if (defer_wg && nwg < 6) { wg[nwg++] = h; return 0; }
        return immtsf_launch_gemm(GEMM_TN, prec, h, fk.fork());
    };

    const bool hf = t2v_hf(cfg);
    T2VW W;
    CHECK(t2v_weights(cfg, p, w, s, &W));
    const bool noproj = (cfg->form & IMMTSF_FORM_NO_PROJ) != 0;
    float* dzp = sc.dz;
    if (noproj) dzp = const_cast<float*>(dE_txt);        // the incoming gradient IS dZ (the consumer owns proj_out): LayerNorm's backward works on it in place
    // ---- phase A: proj_out, LayerNorm, out_proj
    if (!noproj && (phA || wgA)) {
        Mat dE = cmat(dE_txt);
        if (hf && cfg->in_h) {
            dE.h = const_cast<void*>(cfg->in_h);      // the producer (MMF key/value backward) wrote the bf16 image already
        } else if (hf) {       // the upstream gradient arrives as fp32 (block boundary): one cast for its two GEMMs
            if (phA) CHECK(launch_f32_to_bf16(dE_txt, sc.dE.h, (size_t)BT * d, s));
            dE.h = sc.dE.h;
        }
        if (phA) {   // proj_out: dz = dE W_po
            GemmArgs g = gemm_args(BT, d, d, d, d, d);
            set_problem2(g, 0, dE, W.po, mat(sc.dz), nullptr);
            CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
        }
        if (wgA) {   // dW_po = dE^T z ; db_po = colsum dE
            GemmArgs h = gemm_args(d, d, BT, d, d, d);
            set_problem2(h, 0, dE, w.z, mat(gr->proj_out_w), nullptr, gr->proj_out_b);
            prezeroed(h, cfg);
            h.ws = sc.sk[0]; h.ws_bytes = sc.skb[0];
            CHECK(wgrad(h));
        }
    }
    if (phA) {
        // LayerNorm backward; its parameter gradients; residual: dQ_param = sum of dx over ALL (b,t) rows; then only the windows with
        // notes feed the attention branch (rows zeroed, bf16 image written) -- ONE pass over the rows (launch_layernorm_bwd_sums), or,
        // for small / unaligned cases, the LayerNorm backward and the three sums as two passes
        const immtsf_lowrank_grad* lr = cfg->lr_grad;
        if (lr && (!noproj || !lr->coef || !lr->basis || lr->rank <= 0 || lr->ld < lr->rank)) return IMMTSF_EINVAL;
        // (lr: the upstream gradient is dZ = coef basis, formed inside the LayerNorm backward -- immtsf_ttf_t2v_xattn_accepts_lowrank)
        const int rc = lr ? launch_layernorm_bwd_lr(lr->coef, lr->ld, lr->rank, lr->basis, BT, d, p->ln_w, w.xhat, nullptr, w.rstd, sc.dx.f, sc.dx.h,
                                                    drop, SITE_T2V_OUT, gr->ln_w, gr->ln_b, gr->Q_param, sc.red, w.mtxt, T, s)
                          : launch_layernorm_bwd_sums(dzp, BT, d, p->ln_w, w.xhat, w.rstd, sc.dx.f, drop, SITE_T2V_OUT, gr->ln_w, gr->ln_b,
                                                      gr->Q_param, sc.red, w.mtxt, T, sc.dx.h, s);
        if (lr && rc != IMMTSF_OK) return rc == IMMTSF_EUNSUPPORTED ? IMMTSF_EINVAL : rc;
        if (rc == IMMTSF_EUNSUPPORTED) {
            CHECK(launch_layernorm_bwd(dzp, BT, d, p->ln_w, w.xhat, w.rstd, sc.dx.f, drop, SITE_T2V_OUT, s));
            CHECK(launch_colsum3(dzp, w.xhat, sc.dx.f, BT, d, d, gr->ln_w, gr->ln_b, gr->Q_param, sc.red, w.mtxt, T, sc.dx.h, s, true));
        } else {
            CHECK(rc);
        }
        // out_proj: dctx = dx W_out
        GemmArgs g = gemm_args(BT, d, d, d, d, d);
        set_problem2(g, 0, sc.dx, W.out, mat(sc.dctx), nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    if (wgA) {
        GemmArgs h = gemm_args(d, d, BT, d, d, d);
        set_problem2(h, 0, sc.dx, w.ctx, mat(gr->attn_out_w), nullptr, gr->attn_out_b);
        prezeroed(h, cfg);
        h.ws = sc.sk[1]; h.ws_bytes = sc.skb[1];
        CHECK(wgrad(h));
    }
    // ---- phase B: the ragged attention, the k|v in-projection, the query path
    if (phB) {
        RaggedAttnDims dm; dm.B = B; dm.T = T; dm.H = H; dm.hd = hd; dm.N = N;
        CHECK(launch_ragged_attn_bwd(dm, w.offsets, w.rowmap, w.KVp.f, w.qs, w.P, sc.dctx, sc.dKVp.f, sc.dqs_part, sc.dp, drop,
                                     SITE_T2V_ATTN, s, sc.dKVp.h, w.KVp.h));
        GemmArgs g = gemm_args(R, d, 2 * d, 2 * d, d, d);
        set_problem2(g, 0, sc.dKVp, W.inkv, sc.dKV, nullptr);
        g.dyn = total; g.dyn_which = 0;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    if (wgB) {
        GemmArgs h = gemm_args(2 * d, d, R, 2 * d, d, d);
        set_problem2(h, 0, sc.dKVp, w.KV, mat(gr->attn_in_w + (size_t)d * d), nullptr, gr->attn_in_b + d);
        h.dyn = total; h.dyn_which = 1;
        h.ws = sc.sk[2]; h.ws_bytes = sc.skb[2];
        prezeroed(h, cfg);
        CHECK(wgrad(h));
    }
    // ---- phase C: KV_proj, input_proj, Time2Vec
    if (phC) {
        GemmArgs g = gemm_args(R, dcat, d, d, dcat, dcat);
        set_problem2(g, 0, sc.dKV, W.kv, mat(sc.dXcat, p->input_proj_w ? sc.dXcat_h : nullptr), nullptr);
        g.dyn = total; g.dyn_which = 0;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    if (wgC) {
        {
            GemmArgs h = gemm_args(d, dcat, R, d, dcat, dcat);
            set_problem2(h, 0, sc.dKV, w.Xcat, mat(gr->kv_w), nullptr, gr->kv_b);
            h.dyn = total; h.dyn_which = 1;
            h.ws = sc.sk[3]; h.ws_bytes = sc.skb[3];
            prezeroed(h, cfg);
            CHECK(wgrad(h));
        }
        constexpr int notes_image_b = 1;
        if (p->input_proj_w && hf && notes_image_b) {   // dW_in = dVp^T V ; db_in = colsum dVp: both operands are bf16 images (dXcat's first d columns, the packed notes)
            GemmArgs h = gemm_args(d, cfg->d_m, R, dcat, cfg->d_m, cfg->d_m);
            set_problem2(h, 0, mat(nullptr, sc.dXcat_h), mat(nullptr, w.Vh), mat(gr->input_proj_w), nullptr, gr->input_proj_b);
            h.dyn = total; h.dyn_which = 1;
            h.ws = sc.sk[4]; h.ws_bytes = sc.skb[4];
            prezeroed(h, cfg);
            CHECK(wgrad(h));
        } else if (p->input_proj_w) {   // row-mapped, fp32 operands: the round-1 kernel
            GemmArgs h = gemm_args(d, cfg->d_m, R, dcat, cfg->d_m, cfg->d_m);
            set_problem(h, 0, sc.dXcat, notes, gr->input_proj_w, nullptr, gr->input_proj_b);
            h.dyn = total; h.dyn_which = 1; h.b_rowmap = src_rows ? src_rows : w.rowmap;
            prezeroed(h, cfg);
            CHECK(wgrad(h));
        }
    }
    // the small parameter-gradient kernels: the query path (q = W_q Q_param + b_q, qs = q * scale: dW_q -- rows 0..d of in_proj_weight --,
    // db_q, dQ_param += W_q^T dq) belongs to phase B's parameter gradients, Time2Vec's to phase C's; in one call they share a launch
    if (wgB && wgC) {
        CHECK(launch_query_t2v_bwd(sc.dqs_part, B, d, scale, p->attn_in_w, d, p->Q_param, gr->attn_in_w, d, gr->attn_in_b, gr->Q_param, tau,
                                   w.rowmap, total, R, dt, p->t2v_per_w, p->t2v_per_b, sc.dXcat + d, dcat, gr->t2v_lin_w, gr->t2v_lin_b,
                                   gr->t2v_per_w, gr->t2v_per_b, sc.red_t2v, sc.t2v_slabs, s));
    } else if (wgB) {
        CHECK(launch_query_bwd(sc.dqs_part, B, d, scale, p->attn_in_w, d, p->Q_param, gr->attn_in_w, d, gr->attn_in_b, gr->Q_param, s));
    } else if (wgC) {
        CHECK(launch_time2vec_bwd(tau, w.rowmap, total, R, dt, p->t2v_per_w, p->t2v_per_b, sc.dXcat + d, dcat, gr->t2v_lin_w, gr->t2v_lin_b,
                                  gr->t2v_per_w, gr->t2v_per_b, sc.red_t2v, sc.t2v_slabs, s));
    }
    CHECK(immtsf_launch_gemm_tn_list(prec, wg, nwg, s));
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
