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
#include "rowops.hpp"

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
    float *d1, *dff, *dh, *dsa, *da, *dqkv, *red;
    size_t bytes;
};
ELScratch carve_el_scratch(size_t R, size_t D, size_t F, void* base) {
    Carver k(base);
    ELScratch s;
    s.d1 = k.take<float>(R * D);
    s.dff = k.take<float>(R * D);
    s.dh = k.take<float>(R * F);
    s.dsa = k.take<float>(R * D);
    s.da = k.take<float>(R * D);
    s.dqkv = k.take<float>(R * 3 * D);
    s.red = k.take<float>(64 * (D + 8));
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
inline bool bad_el(const immtsf_encoder_layer_cfg* c) {
    return !c || c->Bs <= 0 || c->S <= 0 || c->D <= 0 || c->H <= 0 || c->F <= 0 || (c->D % c->H) || (c->D & 3) || c->D > 1024 ||
           c->precision < 0 || c->precision > 1 || c->p_attn < 0.f || c->p_attn >= 1.f || c->p_drop < 0.f || c->p_drop >= 1.f;
}

}  // namespace

extern "C" {

size_t immtsf_encoder_layer_workspace_bytes(const immtsf_encoder_layer_cfg* c) {
    return bad_el(c) ? 0 : carve_el((size_t)c->Bs * c->S, c->D, c->F, nullptr).bytes;
}
size_t immtsf_encoder_layer_scratch_bytes(const immtsf_encoder_layer_cfg* c) {
    return bad_el(c) ? 0 : carve_el_scratch((size_t)c->Bs * c->S, c->D, c->F, nullptr).bytes;
}

int immtsf_encoder_layer_forward(const immtsf_encoder_layer_cfg* c, const immtsf_encoder_layer_params* p, const float* x, float* out,
                                 void* workspace, size_t workspace_bytes, immtsf_stream_t stream) {
    if (bad_el(c) || !p || !x || !out || !workspace) return IMMTSF_EINVAL;
    const int R = c->Bs * c->S, D = c->D, F = c->F, E = D / c->H, prec = c->precision;
    ELWs w = carve_el(R, D, F, workspace);
    if (workspace_bytes < w.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DropCfg da = el_drop(c, c->p_attn), dd = el_drop(c, c->p_drop), none = el_drop(c, 0.f);
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
    {   // h = dropout(relu(x1 W1^T + b1)): activation and dropout in the epilogue
        GemmArgs g = gemm_args(R, F, D, D, D, F);
        set_problem(g, 0, w.x1, p->w1, w.h, p->b1);
        g.act = 1;
        g.epi_drop = dd; g.epi_site = c->site_base + 2;
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    {
        GemmArgs g = gemm_args(R, D, F, F, F, D);
        set_problem(g, 0, w.h, p->w2, w.ff, p->b2);
        CHECK(immtsf_launch_gemm(GEMM_NT, prec, g, s));
    }
    return launch_layernorm_fwd(w.ff, R, D, p->ln2_w, p->ln2_b, c->eps, w.xhat2, w.rstd2, out, none, 0, s, nullptr, w.x1, dd, c->site_base + 3);
}

/* dout (R, D) -> dx (R, D) and the parameter gradients in `gr` (same layout as the parameters; every buffer is overwritten
 * unless grads_prezeroed, in which case split-K weight gradients add into the zeros they were given) */
int immtsf_encoder_layer_backward(const immtsf_encoder_layer_cfg* c, const immtsf_encoder_layer_params* p, const float* x, const float* dout,
                                  float* dx, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                                  const immtsf_encoder_layer_params* gr, immtsf_stream_t stream) {
    if (bad_el(c) || !p || !gr || !x || !dout || !dx || !workspace || !scratch) return IMMTSF_EINVAL;
    const int R = c->Bs * c->S, D = c->D, F = c->F, E = D / c->H, prec = c->precision;
    ELWs w = carve_el(R, D, F, workspace);
    ELScratch sc = carve_el_scratch(R, D, F, scratch);
    if (workspace_bytes < w.bytes || scratch_bytes < sc.bytes) return IMMTSF_EWORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DropCfg da = el_drop(c, c->p_attn), dd = el_drop(c, c->p_drop), none = el_drop(c, 0.f);
    const int pz = c->grads_prezeroed ? 1 : 0;
    auto wgrad = [&](const float* dy, const float* xin, int N, int K, float* dW, float* db) {      // dW (N,K) = dy^T xin ; db = colsum dy
        GemmArgs g = gemm_args(N, K, R, N, K, K);
        set_problem(g, 0, dy, xin, dW, nullptr, db);
        g.c_prezeroed = pz;
        return immtsf_launch_gemm(GEMM_TN, prec, g, s);
    };
    // LayerNorm2: d1 = gradient of (x1 + drop(ff)) -- the residual's share; dff = d1 * dropout mask
    // (no output dropout: the kernel leaves dz untouched, so the caller's dout is read in place)
    float* g2 = const_cast<float*>(dout);
    CHECK(launch_layernorm_bwd(g2, R, D, p->ln2_w, w.xhat2, w.rstd2, sc.d1, none, 0, s, sc.dff, dd, c->site_base + 3));
    CHECK(launch_colsum2(g2, w.xhat2, R, D, D, gr->ln2_w, gr->ln2_b, sc.red, s));
    {   // linear2: dh = (dff W2) masked by h != 0 (relu' and the feed-forward dropout in one) and scaled by 1/keep
        GemmArgs g = gemm_args(R, F, D, D, F, F);
        set_problem(g, 0, sc.dff, p->w2, sc.dh, nullptr);
        g.relu_ref = w.h; g.ld_ref = F;
        g.alpha = dd.inv_keep;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    CHECK(wgrad(sc.dff, w.h, D, F, gr->w2, gr->b2));
    {   // linear1: d1 += dh W1
        GemmArgs g = gemm_args(R, D, F, F, D, D);
        set_problem(g, 0, sc.dh, p->w1, sc.d1, nullptr);
        g.accumulate = 1;
        CHECK(immtsf_launch_gemm(GEMM_NN, prec, g, s));
    }
    CHECK(wgrad(sc.dh, w.x1, F, D, gr->w1, gr->b1));
    // LayerNorm1: dx = gradient of (x + drop(sa)) -- the input's residual share; dsa = dx * dropout mask
    CHECK(launch_layernorm_bwd(sc.d1, R, D, p->ln1_w, w.xhat1, w.rstd1, dx, none, 0, s, sc.dsa, dd, c->site_base + 1));
    CHECK(launch_colsum2(sc.d1, w.xhat1, R, D, D, gr->ln1_w, gr->ln1_b, sc.red, s));
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

}  // extern "C"
