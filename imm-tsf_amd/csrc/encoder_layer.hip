// One post-norm transformer encoder layer (self-attention over short sequences + ReLU feed-forward) as a block-level entry
// point: tPatchGNN's nn.TransformerEncoderLayer over the M patches of a variable (reference models/tPatchGNN.py:118-121,
// 200-205; d_model = hid_dim = 32, dim_feedforward = 2048, sequences of M = 2 patches at the benchmark configuration).
//
//   qkv = x W_in^T + b_in ; a = softmax(q k^T / sqrt(E)) v  (attention-weight dropout) ; sa = a W_o^T + b_o
//   x1  = LayerNorm1(x + Dropout(sa))
//   h   = Dropout(relu(x1 W_1^T + b_1)) ; ff = h W_2^T + b_2
//   out = LayerNorm2(x1 + Dropout(ff))
//
// Composed op by op from autograd Functions this is 13 launches forward and ~24 backward (three dropout kernels, two
// adds, mask multiplies, a relu mask, clones and the residual-gradient adds of autograd) around GEMMs that take ~8 us
// each: the layer is launch-chain bound.  Here the residual add and the branch dropout live inside the LayerNorm
// kernels, the feed-forward dropout inside the GEMM epilogue (Philox, regenerated nowhere: the saved h carries it),
// relu' and that dropout's mask are one epilogue mask of the data-gradient GEMM (h == 0), and the residual gradients are
// accumulated by the GEMMs that produce them: 7 launches forward, 15 backward.
#include "../../include/immtsf.h"
#include "attn.hpp"
#include "block_util.hpp"
#include "enc_head.hpp"
#include "ffn32.hpp"
#include "rowops.hpp"
#include "skinny_tn.hpp"
#include <stdlib.h>

namespace {

struct ELWs {
    float *qkv, *a, *sa, *xhat1, *rstd1, *x1, *h, *ff, *xhat2, *rstd2;
    size_t bytes;
};
ELWs carve_el(size_t R, size_t D, size_t F, void* base) {
    Carver k(base);
    ELWs w;
    w.qkv = k.take<float>(R * 3 * D);
    w.a = k.take<float>(R * D);
    w.sa = k.take<float>(R * D);
    w.xhat1 = k.take<float>(R * D);
    w.rstd1 = k.take<float>(R);
    w.x1 = k.take<float>(R * D);
    w.h = k.take<float>(R * F);
    w.ff = k.take<float>(R * D);
    w.xhat2 = k.take<float>(R * D);
    w.rstd2 = k.take<float>(R);
    w.bytes = k.bytes();
    return w;
}
struct ELScratch {
    float *d1, *dff, *dh, *dsa, *da, *dqkv, *red, *sk;      // sk: slabs of the skinny weight-gradient kernel (d_model <= 64)
    float* eh;                                               // slabs of the fused attention half (enc_head.hip)
    size_t bytes;
};
ELScratch carve_el_scratch(size_t R, size_t D, size_t F, void* base, size_t eh_floats = 0) {
    Carver k(base);
    ELScratch s;
    s.eh = eh_floats ? k.take<float>(eh_floats) : nullptr;
    s.d1 = k.take<float>(R * D);
    s.dff = k.take<float>(R * D);
    s.dh = k.take<float>(R * F);
    s.dsa = k.take<float>(R * D);
    s.da = k.take<float>(R * D);
    s.dqkv = k.take<float>(R * 3 * D);
    s.red = k.take<float>(colsum_scratch_floats(D, 2));
    s.sk = D <= 64 ? k.take<float>(skinny_tn_scratch_floats(3 * (int)D, (int)D, (int)R)) : nullptr;
    s.bytes = k.bytes();
    return s;
}
inline DropCfg el_drop(const immtsf_encoder_layer_cfg* c, float p) {
    DropCfg d;
    d.seed = c->seed;
    d.p = (c->training && p > 0.f) ? p : 0.f;
    d.inv_keep = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    d.seed_dev = c->seed_step_dev;
    return d;
}
inline size_t el_eh_floats(const immtsf_encoder_layer_cfg* c) {
    return enc_head32_ok(c->Bs, c->S, c->D, c->H) ? enc_head32_slab_floats(c->Bs, c->S, c->H) : 0;
}
inline bool bad_el(const immtsf_encoder_layer_cfg* c) {
    return !c || c->Bs <= 0 || c->S <= 0 || c->D <= 0 || c->H <= 0 || c->F <= 0 || (c->D % c->H) || (c->D & 3) || c->D > 1024 ||
           c->precision < 0 || c->precision > 1 || c->p_attn < 0.f || c->p_attn >= 1.f || c->p_drop < 0.f || c->p_drop >= 1.f;
}

// ---- the feed-forward half shared by the two block entry points:  out = LN(x1 + drop(drop(act(x1 W1^T + b1)) W2^T + b2))
// act 1 = ReLU, 2 = GELU(erf).  Saved: h (with its dropout applied), for GELU also the pre-activation z; LN statistics.
struct FFNDims { int R, D, F, act, prec; float eps; uint64_t site_h, site_out; };
// GELU feed-forward at PatchTST-like sizes in bf16 mode (round 3): both GEMM operands as bf16 images in HBM -> the LDS-DMA kernel
// (gemm2.hip) with the activation / dropout / GELU' epilogues, instead of fp32 activations converted while they are staged
// (gemm.hip: 4608 x 2048 x 512 at 150 - 190 TFLOP/s).  x1 and dff are cast once each (5 us); h and dh exist only as bf16.
// IMMTSF_FFN_HF=0: the fp32-activation path, for A/B runs.
struct FFNHf { unsigned short *x16, *h16, *w1h, *w2h, *dff16, *dh16; };
inline bool ffn_hf(const FFNDims& f) {
    constexpr bool on = true;
    return on && f.prec == 1 && f.act == 2 && f.R >= 1024 && (f.D % 16) == 0 && (f.F % 16) == 0;
}
int ffn_forward(const FFNDims& f, const DropCfg& dd, const DropCfg& none, const float* x1, const float* w1, const float* b1,
                const float* w2, const float* b2, const float* ln_w, const float* ln_b, float* h, float* z, float* ff, float* xhat,
                float* rstd, float* out, hipStream_t s, const FFNHf* hf = nullptr) {
    if (hf) {
        Mat W1, W2;
        CHECK(weight_mat(true, w1, (size_t)f.F * f.D, hf->w1h, s, &W1));
        CHECK(weight_mat(true, w2, (size_t)f.D * f.F, hf->w2h, s, &W2));
        CHECK(launch_f32_to_bf16(x1, hf->x16, (size_t)f.R * f.D, s));
        {   // h16 = bf16(dropout(gelu(z))), z = x1 W1^T + b1 kept in fp32 for the backward's GELU'
            GemmArgs g = gemm_args(f.R, f.F, f.D, f.D, f.D, f.F);
            set_problem2(g, 0, cmat(x1, hf->x16), W1, mat(nullptr, hf->h16), b1);
            g.p[0].Cpre = z;
            g.act = f.act;
            g.epi_drop = dd; g.epi_site = f.site_h;
            CHECK(immtsf_launch_gemm(GEMM_NT, f.prec, g, s));
        }
        {
            GemmArgs g = gemm_args(f.R, f.D, f.F, f.F, f.F, f.D);
            set_problem2(g, 0, mat(nullptr, hf->h16), W2, mat(ff), b2);
            CHECK(immtsf_launch_gemm(GEMM_NT, f.prec, g, s));
        }
        return launch_layernorm_fwd(ff, f.R, f.D, ln_w, ln_b, f.eps, xhat, rstd, out, none, 0, s, nullptr, x1, dd, f.site_out);
    }
    if (ffn32_ok(f.R, f.D, f.F, f.act, f.prec)) {       // many rows, d_model 32: h never exists (ffn32.hip); its buffer holds the mask words
        if (ffn32_saved_bytes(f.R, f.F) > (size_t)f.R * f.F * sizeof(float)) return IMMTSF_EWORKSPACE;
        CHECK(ffn32_forward(f.R, f.F, dd, f.site_h, x1, w1, b1, w2, b2, h, ff, s));
        return launch_layernorm_fwd(ff, f.R, f.D, ln_w, ln_b, f.eps, xhat, rstd, out, none, 0, s, nullptr, x1, dd, f.site_out);
    }
    {   // h = dropout(act(x1 W1^T + b1)): activation and dropout in the epilogue (z = the pre-activation, GELU only)
        GemmArgs g = gemm_args(f.R, f.F, f.D, f.D, f.D, f.F);
        set_problem(g, 0, x1, w1, h, b1);
        g.p[0].Cpre = f.act == 2 ? z : nullptr;
        g.act = f.act;
        g.epi_drop = dd; g.epi_site = f.site_h;
        CHECK(immtsf_launch_gemm(GEMM_NT, f.prec, g, s));
    }
    {
        GemmArgs g = gemm_args(f.R, f.D, f.F, f.F, f.F, f.D);
        set_problem(g, 0, h, w2, ff, b2);
        CHECK(immtsf_launch_gemm(GEMM_NT, f.prec, g, s));
    }
    return launch_layernorm_fwd(ff, f.R, f.D, ln_w, ln_b, f.eps, xhat, rstd, out, none, 0, s, nullptr, x1, dd, f.site_out);
}
// dout -> d1 (gradient wrt x1, complete: residual share + through the two GEMMs) and the six parameter gradients
int ffn_backward(const FFNDims& f, const DropCfg& dd, const DropCfg& none, int pz, const float* x1, const float* w1, const float* b1,
                 const float* w2, const float* ln_w, const float* h, const float* z, const float* xhat, const float* rstd, const float* dout, float* d1,
                 float* dff, float* dh, float* red, float* gw1, float* gb1, float* gw2, float* gb2, float* gln_w, float* gln_b,
                 hipStream_t s, const FFNHf* hf = nullptr, int phase = 0) {
    // phase (the bf16-in-HBM path only): 0 = everything; 1 = the data path (LayerNorm backward, its two parameter sums, dh, d1) -- what
    // the layers in front wait for; 2 = the two weight-gradient products alone, from the bf16 images the phase-1 call left in the
    // workspace / scratch, on any stream ordered behind it (immtsf.train.FlagStep: the parameter branch)
    if (phase && !(hf && !ffn32_ok(f.R, f.D, f.F, f.act, f.prec))) return IMMTSF_EUNSUPPORTED;
    if (phase == 2) {
        const Mat DFF = cmat(dff, hf->dff16), H = mat(nullptr, hf->h16), DH = mat(nullptr, hf->dh16), X = cmat(x1, hf->x16);
        GemmArgs wg[2];
        wg[0] = gemm_args(f.D, f.F, f.R, f.D, f.F, f.F);
        set_problem2(wg[0], 0, DFF, H, mat(gw2), nullptr, gb2);
        wg[0].c_prezeroed = pz;
        wg[1] = gemm_args(f.F, f.D, f.R, f.F, f.D, f.D);
        set_problem2(wg[1], 0, DH, X, mat(gw1), nullptr, gb1);
        wg[1].c_prezeroed = pz;
        return immtsf_launch_gemm_tn_list(f.prec, wg, 2, s);
    }
    auto wgrad = [&](const float* dy, const float* xin, int N, int K, float* dW, float* db) {
        GemmArgs g = gemm_args(N, K, f.R, N, K, K);
        set_problem(g, 0, dy, xin, dW, nullptr, db);
        g.c_prezeroed = pz;
        return immtsf_launch_gemm(GEMM_TN, f.prec, g, s);
    };
    // LayerNorm: d1 = gradient of (x1 + drop(ff)) -- the residual's share; dff = d1 * dropout mask.  No output dropout: the
    // kernel leaves dz untouched, so the caller's dout is read in place
    float* g2 = const_cast<float*>(dout);
    CHECK(launch_layernorm_bwd(g2, f.R, f.D, ln_w, xhat, rstd, d1, none, 0, s, dff, dd, f.site_out));
    if (ffn32_ok(f.R, f.D, f.F, f.act, f.prec)) {       // the forward took the fused path: h holds weight images + mask words
        if (ffn32_scratch_bytes(f.R, f.F) > (size_t)f.R * f.F * sizeof(float)) return IMMTSF_EWORKSPACE;
        CHECK(ffn32_backward(f.R, f.F, dd, x1, b1, dff, h, dh, d1, gw1, gb1, gw2, s));
        if (colsum_small_pair_ok(f.R, f.D))      // few rows: LayerNorm's two parameter gradients and the second bias gradient in one launch
            return launch_colsum_small_pair(g2, xhat, dff, f.R, f.D, f.D, gln_w, gln_b, gb2, s, pz);
        CHECK(launch_colsum2(g2, xhat, f.R, f.D, f.D, gln_w, gln_b, red, s, true));   // (red: colsum_scratch_floats(D, 2) at both carve sites)
        return launch_colsum(dff, nullptr, f.R, nullptr, f.D, f.D, gb2, 0, red, s, true);
    }
    CHECK(launch_colsum2(g2, xhat, f.R, f.D, f.D, gln_w, gln_b, red, s, true));
    if (hf) {
        Mat W1, W2;
        CHECK(weight_mat(true, w1, (size_t)f.F * f.D, hf->w1h, s, &W1));
        CHECK(weight_mat(true, w2, (size_t)f.D * f.F, hf->w2h, s, &W2));
        CHECK(launch_f32_to_bf16(dff, hf->dff16, (size_t)f.R * f.D, s));
        const Mat DFF = cmat(dff, hf->dff16), H = mat(nullptr, hf->h16), DH = mat(nullptr, hf->dh16), X = cmat(x1, hf->x16);
        {   // dh16 = bf16((dff W2) x gelu'(z) x the feed-forward dropout)
            GemmArgs g = gemm_args(f.R, f.F, f.D, f.D, f.F, f.F);
            set_problem2(g, 0, DFF, W2, DH, nullptr);
            g.relu_ref = z; g.ld_ref = f.F; g.ref_kind = 2;
            g.epi_drop = dd; g.epi_site = f.site_h;
            CHECK(immtsf_launch_gemm(GEMM_NN, f.prec, g, s));
        }
        if (phase == 0) {   // dW2 = dff^T h, db2
            GemmArgs g = gemm_args(f.D, f.F, f.R, f.D, f.F, f.F);
            set_problem2(g, 0, DFF, H, mat(gw2), nullptr, gb2);
            g.c_prezeroed = pz;
            CHECK(immtsf_launch_gemm(GEMM_TN, f.prec, g, s));
        }
        {   // d1 += dh W1
            GemmArgs g = gemm_args(f.R, f.D, f.F, f.F, f.D, f.D);
            set_problem2(g, 0, DH, W1, mat(d1), nullptr);
            g.accumulate = 1;
            CHECK(immtsf_launch_gemm(GEMM_NN, f.prec, g, s));
        }
        if (phase == 1) return IMMTSF_OK;
        GemmArgs g = gemm_args(f.F, f.D, f.R, f.F, f.D, f.D);      // dW1 = dh^T x1, db1
        set_problem2(g, 0, DH, X, mat(gw1), nullptr, gb1);
        g.c_prezeroed = pz;
        return immtsf_launch_gemm(GEMM_TN, f.prec, g, s);
    }
    {   // linear2: dh = (dff W2) x act'(.) x the feed-forward dropout
        GemmArgs g = gemm_args(f.R, f.F, f.D, f.D, f.F, f.F);
        set_problem(g, 0, dff, w2, dh, nullptr);
        if (f.act == 2) {      // GELU: the factor comes from the saved pre-activation, the dropout mask is regenerated
            g.relu_ref = z; g.ld_ref = f.F; g.ref_kind = 2;
            g.epi_drop = dd; g.epi_site = f.site_h;
        } else {               // ReLU: h == 0 <=> negative pre-activation or dropped: one mask, scaled by 1 / keep
            g.relu_ref = h; g.ld_ref = f.F;
            g.alpha = dd.inv_keep;
        }
        CHECK(immtsf_launch_gemm(GEMM_NN, f.prec, g, s));
    }
    CHECK(wgrad(dff, h, f.D, f.F, gw2, gb2));
    {   // linear1: d1 += dh W1
        GemmArgs g = gemm_args(f.R, f.D, f.F, f.F, f.D, f.D);
        set_problem(g, 0, dh, w1, d1, nullptr);
        g.accumulate = 1;
        CHECK(immtsf_launch_gemm(GEMM_NN, f.prec, g, s));
    }
    return wgrad(dh, x1, f.F, f.D, gw1, gb1);
}

struct FFNWs { float *h, *z, *ff, *xhat, *rstd; unsigned short *x16, *h16, *w1h, *w2h; size_t bytes; };
FFNWs carve_ffn(size_t R, size_t D, size_t F, int act, bool hf, void* base) {
    Carver k(base);
    FFNWs w;
    w.h = hf ? nullptr : k.take<float>(R * F);
    w.z = act == 2 ? k.take<float>(R * F) : nullptr;
    w.ff = k.take<float>(R * D);
    w.xhat = k.take<float>(R * D);
    w.rstd = k.take<float>(R);
    w.x16 = hf ? k.take<unsigned short>(R * D) : nullptr;
    w.h16 = hf ? k.take<unsigned short>(R * F) : nullptr;
    w.w1h = hf ? k.take<unsigned short>(F * D) : nullptr;
    w.w2h = hf ? k.take<unsigned short>(D * F) : nullptr;
    w.bytes = k.bytes();
    return w;
}
struct FFNScratch { float *dff, *dh, *red; unsigned short *dff16, *dh16; size_t bytes; };
FFNScratch carve_ffn_scratch(size_t R, size_t D, size_t F, bool hf, void* base) {
    Carver k(base);
    FFNScratch s;
    s.dff = k.take<float>(R * D);
    s.dh = hf ? nullptr : k.take<float>(R * F);
    s.red = k.take<float>(colsum_scratch_floats(D, 2));
    s.dff16 = hf ? k.take<unsigned short>(R * D) : nullptr;
    s.dh16 = hf ? k.take<unsigned short>(R * F) : nullptr;
    s.bytes = k.bytes();
    return s;
}
inline FFNDims ffn_dims(const immtsf_ffn_block_cfg* c) {
    return FFNDims{c->R, c->D, c->F, c->act, c->precision, c->eps, c->site_base, c->site_base + 1};
}
inline DropCfg mk_drop3(int training, float p, uint64_t seed, const uint64_t* seed_dev) {
    DropCfg d;
    d.seed = seed;
    d.p = (training && p > 0.f) ? p : 0.f;
    d.inv_keep = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    d.seed_dev = seed_dev;
    return d;
}
inline bool bad_ffn(const immtsf_ffn_block_cfg* c) {
    return !c || c->R <= 0 || c->D <= 0 || c->F <= 0 || (c->D & 3) || c->D > 1024 || (c->act != 1 && c->act != 2) || c->precision < 0 ||
           c->precision > 1 || c->p_drop < 0.f || c->p_drop >= 1.f;
}

}  // namespace

extern "C" {

size_t immtsf_encoder_layer_workspace_bytes(const immtsf_encoder_layer_cfg* c) {
    return bad_el(c) ? 0 : carve_el((size_t)c->Bs * c->S, c->D, c->F, nullptr).bytes;
}
size_t immtsf_encoder_layer_scratch_bytes(const immtsf_encoder_layer_cfg* c) {
    return bad_el(c) ? 0 : carve_el_scratch((size_t)c->Bs * c->S, c->D, c->F, nullptr, el_eh_floats(c)).bytes;
}

int immtsf_encoder_layer_forward(const immtsf_encoder_layer_cfg* c, const immtsf_encoder_layer_params* p, const float* x, float* out,
                                 void* workspace, size_t workspace_bytes, immtsf_stream_t stream) {
    if (bad_el(c) || !p || !x || !out || !workspace) return IMMTSF_EINVAL;
    const int R = c->Bs * c->S, D = c->D, F = c->F, E = D / c->H, prec = c->precision;
    ELWs w = carve_el(R, D, F, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DropCfg da = el_drop(c, c->p_attn), dd = el_drop(c, c->p_drop), none = el_drop(c, 0.f);
    if (enc_head32_ok(c->Bs, c->S, D, c->H)) {       // d_model 32: projections, attention and LayerNorm1 as one kernel (enc_head.hip)
        CHECK(launch_enc_head32_fwd(x, c->Bs, c->S, c->H, p->in_w, p->in_b, p->out_w, p->out_b, p->ln1_w, p->ln1_b, c->eps, da, dd,
                                    c->site_base + 0, w.x1, w.xhat1, w.rstd1, s));
    } else {
        {   // packed in-projection
            GemmArgs g = gemm_args(R, 3 * D, D, D, D, 3 * D);
            set_problem(g, 0, x, p->in_w, w.qkv, p->in_b);
            CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
        }
        CHECK(launch_attn_short_fwd(w.qkv, c->Bs, c->S, c->H, E, 1.0f / sqrtf((float)E), 0, da, c->site_base + 0, w.a, s));
        {
            GemmArgs g = gemm_args(R, D, D, D, D, D);
            set_problem(g, 0, w.a, p->out_w, w.sa, p->out_b);
            CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
        }
        CHECK(launch_layernorm_fwd(w.sa, R, D, p->ln1_w, p->ln1_b, c->eps, w.xhat1, w.rstd1, w.x1, none, 0, s, nullptr, x, dd, c->site_base + 1));
    }
    const FFNDims f{R, D, F, 1, prec, c->eps, c->site_base + 2, c->site_base + 3};
    return ffn_forward(f, dd, none, w.x1, p->w1, p->b1, p->w2, p->b2, p->ln2_w, p->ln2_b, w.h, nullptr, w.ff, w.xhat2, w.rstd2, out, s);
}

/* dout (R, D) -> dx (R, D) and the parameter gradients in `gr` (same layout as the parameters; every buffer is overwritten
 * unless grads_prezeroed, in which case split-K weight gradients add into the zeros they were given) */
int immtsf_encoder_layer_backward(const immtsf_encoder_layer_cfg* c, const immtsf_encoder_layer_params* p, const float* x, const float* dout,
                                  float* dx, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                  const immtsf_encoder_layer_params* gr, immtsf_stream_t stream) {
    if (bad_el(c) || !p || !gr || !x || !dout || !dx || !workspace || !scratch) return IMMTSF_EINVAL;
    const int R = c->Bs * c->S, D = c->D, F = c->F, E = D / c->H, prec = c->precision;
    ELWs w = carve_el(R, D, F, workspace);
    ELScratch sc = carve_el_scratch(R, D, F, scratch, el_eh_floats(c));
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DropCfg da = el_drop(c, c->p_attn), dd = el_drop(c, c->p_drop), none = el_drop(c, 0.f);
    const int pz = c->grads_prezeroed ? 1 : 0;
    auto wgrad = [&](const float* dy, const float* xin, int N, int K, float* dW, float* db) {      // dW (N,K) = dy^T xin ; db = colsum dy
        // many rows, tiny output (d_model 32: 96 x 32 and 32 x 32 over 65 k rows at 4096 windows): the streaming kernel of skinny_tn.hip
        if (prec == 1 && sc.sk && skinny_tn_ok(N, K, R, N, K, dy, xin, dW)) return launch_skinny_tn(dy, N, N, xin, K, K, R, dW, db, sc.sk, s);
        GemmArgs g = gemm_args(N, K, R, N, K, K);
        set_problem(g, 0, dy, xin, dW, nullptr, db);
        g.c_prezeroed = pz;
        return immtsf_launch_gemm(GEMM_TN, prec, g, s);
    };
    {
        const FFNDims f{R, D, F, 1, prec, c->eps, c->site_base + 2, c->site_base + 3};
        CHECK(ffn_backward(f, dd, none, pz, w.x1, p->w1, p->b1, p->w2, p->ln2_w, w.h, nullptr, w.xhat2, w.rstd2, dout, sc.d1, sc.dff, sc.dh, sc.red,
                           gr->w1, gr->b1, gr->w2, gr->b2, gr->ln2_w, gr->ln2_b, s));
    }
    if (sc.eh)        // the attention half's backward as one kernel + the slab sum
        return launch_enc_head32_bwd(x, sc.d1, w.xhat1, w.rstd1, c->Bs, c->S, c->H, p->in_w, p->in_b, p->out_w, p->out_b, p->ln1_w, c->eps, da, dd,
                                     c->site_base + 0, dx, gr->in_w, gr->in_b, gr->out_w, gr->out_b, gr->ln1_w, gr->ln1_b, sc.eh, s);
    // LayerNorm1: dx = gradient of (x + drop(sa)) -- the input's residual share; dsa = dx * dropout mask
    CHECK(launch_layernorm_bwd(sc.d1, R, D, p->ln1_w, w.xhat1, w.rstd1, dx, none, 0, s, sc.dsa, dd, c->site_base + 1));
    CHECK(launch_colsum2(sc.d1, w.xhat1, R, D, D, gr->ln1_w, gr->ln1_b, sc.red, s, true));
    {   // out_proj
        GemmArgs g = gemm_args(R, D, D, D, D, D);
        set_problem(g, 0, sc.dsa, p->out_w, sc.da, nullptr);
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    CHECK(wgrad(sc.dsa, w.a, D, D, gr->out_w, gr->out_b));
    CHECK(launch_attn_short_bwd(w.qkv, sc.da, c->Bs, c->S, c->H, E, 1.0f / sqrtf((float)E), 0, da, c->site_base + 0, sc.dqkv, s));
    {   // in-projection: dx += dqkv W_in
        GemmArgs g = gemm_args(R, D, 3 * D, 3 * D, D, D);
        set_problem(g, 0, sc.dqkv, p->in_w, dx, nullptr);
        g.accumulate = 1;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    return wgrad(sc.dqkv, x, 3 * D, D, gr->in_w, gr->in_b);
}

/* ---- LayerNorm(x + Dropout(branch)): the residual joint of every post-norm block (reference layers/Transformer_EncDec.py:52,58) */
int immtsf_residual_layernorm_forward(const float* x, const float* branch, int32_t rows, int32_t d, const float* gamma, const float* beta,
                                      float eps, int32_t training, float p_drop, uint64_t seed, uint64_t site, const uint64_t* seed_step_dev,
                                      float* xhat, float* rstd, float* out, immtsf_stream_t stream) {
    if (!x || !branch || !gamma || !beta || !xhat || !rstd || !out || rows < 0 || d <= 0 || p_drop < 0.f || p_drop >= 1.f) return IMMTSF_EINVAL;
    const DropCfg dd = mk_drop3(training, p_drop, seed, seed_step_dev), none = mk_drop3(0, 0.f, 0, nullptr);
    return launch_layernorm_fwd(branch, rows, d, gamma, beta, eps, xhat, rstd, out, none, 0, static_cast<hipStream_t>(stream), nullptr, x, dd, site);
}

/* dout -> dx (= the gradient of the normalised sum) and dbranch = dx * dropout mask; dgamma, dbeta written; scratch >= 64 (d + 8)
 * floats.  dout is only read. */
int immtsf_residual_layernorm_backward(const float* dout, int32_t rows, int32_t d, const float* gamma, const float* xhat, const float* rstd,
                                       int32_t training, float p_drop, uint64_t seed, uint64_t site, const uint64_t* seed_step_dev, float* dx,
                                       float* dbranch, float* dgamma, float* dbeta, float* scratch, immtsf_stream_t stream) {
    if (!dout || !gamma || !xhat || !rstd || !dx || !dbranch || !dgamma || !dbeta || !scratch || rows < 0 || d <= 0) return IMMTSF_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DropCfg dd = mk_drop3(training, p_drop, seed, seed_step_dev), none = mk_drop3(0, 0.f, 0, nullptr);
    float* g = const_cast<float*>(dout);
    CHECK(launch_layernorm_bwd(g, rows, d, gamma, xhat, rstd, dx, none, 0, s, dbranch, dd, site));
    return launch_colsum2(g, xhat, rows, d, d, dgamma, dbeta, scratch, s);
}

size_t immtsf_ffn_block_workspace_bytes(const immtsf_ffn_block_cfg* c) {
    return bad_ffn(c) ? 0 : carve_ffn(c->R, c->D, c->F, c->act, ffn_hf(ffn_dims(c)), nullptr).bytes;
}
size_t immtsf_ffn_block_scratch_bytes(const immtsf_ffn_block_cfg* c) {
    return bad_ffn(c) ? 0 : carve_ffn_scratch(c->R, c->D, c->F, ffn_hf(ffn_dims(c)), nullptr).bytes;
}

int immtsf_ffn_block_forward(const immtsf_ffn_block_cfg* c, const immtsf_ffn_block_params* p, const float* x, float* out, void* workspace,
                             size_t workspace_bytes, immtsf_stream_t stream) {
    if (bad_ffn(c) || !p || !x || !out || !workspace) return IMMTSF_EINVAL;
    const FFNDims f = ffn_dims(c);
    const bool hf = ffn_hf(f);
    FFNWs w = carve_ffn(c->R, c->D, c->F, c->act, hf, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    const DropCfg dd = mk_drop3(c->training, c->p_drop, c->seed, c->seed_step_dev), none = mk_drop3(0, 0.f, 0, nullptr);
    const FFNHf im{w.x16, w.h16, w.w1h, w.w2h, nullptr, nullptr};
    return ffn_forward(f, dd, none, x, p->w1, p->b1, p->w2, p->b2, p->ln_w, p->ln_b, w.h, w.z, w.ff, w.xhat, w.rstd, out,
                       static_cast<hipStream_t>(stream), hf ? &im : nullptr);
}

int immtsf_ffn_block_backward(const immtsf_ffn_block_cfg* c, const immtsf_ffn_block_params* p, const float* x, const float* dout, float* dx,
                              void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                              const immtsf_ffn_block_params* gr, immtsf_stream_t stream) {
    if (bad_ffn(c) || !p || !gr || !x || !dout || !dx || !workspace || !scratch) return IMMTSF_EINVAL;
    const FFNDims f = ffn_dims(c);
    const bool hf = ffn_hf(f);
    FFNWs w = carve_ffn(c->R, c->D, c->F, c->act, hf, workspace);
    FFNScratch sc = carve_ffn_scratch(c->R, c->D, c->F, hf, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    const DropCfg dd = mk_drop3(c->training, c->p_drop, c->seed, c->seed_step_dev), none = mk_drop3(0, 0.f, 0, nullptr);
    const FFNHf im{w.x16, w.h16, w.w1h, w.w2h, sc.dff16, sc.dh16};
    // grads_prezeroed: bit 0 = the gradient buffers are zero already; bits 1-2 = phase (2: data path only, 4: weight gradients only)
    const int phase = (c->grads_prezeroed >> 1) & 3;
    if (phase == 3) return IMMTSF_EINVAL;
    return ffn_backward(f, dd, none, c->grads_prezeroed & 1, x, p->w1, p->b1, p->w2, p->ln_w, w.h, w.z, w.xhat, w.rstd, dout, dx, sc.dff, sc.dh,
                        sc.red, gr->w1, gr->b1, gr->w2, gr->b2, gr->ln_w, gr->ln_b, static_cast<hipStream_t>(stream), hf ? &im : nullptr, phase);
}

}  // extern "C"
