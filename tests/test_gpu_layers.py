"""GPU parity of the layers/ mirror against golden vectors captured from the real reference:
FullAttention, AttentionLayer, Encoder/EncoderLayer (gelu, d_ff), PatchEmbedding, DataEmbedding.  fp32 mode,
1e-4 outputs / 2e-4 gradients relative to max."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _z(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _t(a, dev):
    return torch.from_numpy(np.asarray(a)).to(dev)


def _rel(a, b, floor=1e-3):
    b = b.detach().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b))
    a, b = a.detach().double().cpu(), b.double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / max(float(b.abs().max()), floor))


def test_full_attention():
    dev = _dev()
    from layers.SelfAttention_Family import FullAttention
    z = _z("layer_full_attention")
    q, k, v = [_t(z[n], dev).requires_grad_(True) for n in ("q", "k", "v")]
    fa = FullAttention(False, attention_dropout=0.0)
    o, attn = fa(q, k, v, None)
    assert attn is None and _rel(o, z["out"]) < 1e-4
    (o * _t(z["upstream"], dev)).sum().backward()
    assert _rel(q.grad, z["gq"]) < 2e-4 and _rel(k.grad, z["gk"]) < 2e-4 and _rel(v.grad, z["gv"]) < 2e-4


def test_full_attention_causal_and_dropout_vs_torch():
    dev = _dev()
    from immtsf import config, ops
    from layers.SelfAttention_Family import FullAttention
    g = torch.Generator().manual_seed(0)
    B, L, H, E = 3, 37, 2, 24
    q, k, v = [torch.randn(B, L, H, E, generator=g).to(dev).requires_grad_(True) for _ in range(3)]
    fa = FullAttention(True, attention_dropout=0.0).to(dev)
    o, _ = fa(q, k, v, None)
    s = torch.einsum("blhe,bshe->bhls", q, k) / E ** 0.5
    s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool, device=dev), 1), float("-inf"))
    ref = torch.einsum("bhls,bshd->blhd", torch.softmax(s, -1), v)
    assert _rel(o, ref.detach().cpu()) < 1e-4
    # dropout: export the mask the kernel used and replay it in torch
    config.manual_seed(5)
    fd = FullAttention(False, attention_dropout=0.3).to(dev).train()
    o2, _ = fd(q, k, v, None)
    config.manual_seed(5)
    seed = config.next_seed()
    keep = ops.dropout_keep_mask(seed, fd.site, B * H * L * L, 0.3, dev).view(B, H, L, L).float()
    a = torch.softmax(torch.einsum("blhe,bshe->bhls", q, k) / E ** 0.5, -1) * keep / 0.7
    ref2 = torch.einsum("bhls,bshd->blhd", a, v)
    assert _rel(o2, ref2.detach().cpu()) < 1e-4
    up = torch.randn(o2.shape, generator=g).to(dev)
    gq = torch.autograd.grad((o2 * up).sum(), q, retain_graph=True)[0]
    gq_ref = torch.autograd.grad((ref2 * up).sum(), q)[0]
    assert _rel(gq, gq_ref.cpu()) < 2e-4


@pytest.mark.parametrize("B,L,S,H,E,D", [(3, 10, 10, 2, 256, 256), (2, 32, 32, 1, 64, 32), (5, 7, 12, 3, 24, 40), (1, 1, 1, 1, 4, 4)])
def test_full_attention_short_wide_one_kernel_vs_torch(B, L, S, H, E, D):
    """csrc/attn_mid.hip (FullAttention over <= 32 positions, heads up to 256 wide: one kernel per direction) against torch: plain,
    causal, and with dropout under the exported Philox mask; every gradient; and against the batched-GEMM path it replaces"""
    dev = _dev()
    from immtsf import config, ops
    from layers.SelfAttention_Family import FullAttention
    g = torch.Generator().manual_seed(L * 100 + E)
    q = torch.randn(B, L, H, E, generator=g).to(dev).requires_grad_(True)
    k = torch.randn(B, S, H, E, generator=g).to(dev).requires_grad_(True)
    v = torch.randn(B, S, H, D, generator=g).to(dev).requires_grad_(True)
    up = torch.randn(B, L, H, D, generator=g).to(dev)
    for causal in ((False, True) if L == S else (False,)):
        fa = FullAttention(causal, attention_dropout=0.0).to(dev)
        o, _ = fa(q, k, v, None)
        s_ = torch.einsum("blhe,bshe->bhls", q, k) / E ** 0.5
        if causal:
            s_ = s_.masked_fill(torch.triu(torch.ones(L, S, dtype=torch.bool, device=dev), 1), float("-inf"))
        ref = torch.einsum("bhls,bshd->blhd", torch.softmax(s_, -1), v)
        assert _rel(o, ref.detach().cpu()) < 1e-5
        got = torch.autograd.grad((o * up).sum(), (q, k, v))
        want = torch.autograd.grad((ref * up).sum(), (q, k, v))
        for a, b in zip(got, want):
            assert _rel(a, b.cpu()) < 1e-4
    config.manual_seed(7)
    fd = FullAttention(False, attention_dropout=0.3).to(dev).train()
    o2, _ = fd(q, k, v, None)
    config.manual_seed(7)
    seed = config.next_seed()
    keep = ops.dropout_keep_mask(seed, fd.site, B * H * L * S, 0.3, dev).view(B, H, L, S).float()
    a_ = torch.softmax(torch.einsum("blhe,bshe->bhls", q, k) / E ** 0.5, -1) * keep / 0.7
    ref2 = torch.einsum("bhls,bshd->blhd", a_, v)
    assert _rel(o2, ref2.detach().cpu()) < 1e-5
    got = torch.autograd.grad((o2 * up).sum(), (q, k, v))
    want = torch.autograd.grad((ref2 * up).sum(), (q, k, v))
    for a, b in zip(got, want):
        assert _rel(a, b.cpu()) < 1e-4
    # the batched-GEMM path with the same seed draws the same mask
    try:
        config.attn_mid = False
        config.manual_seed(7)
        o3, _ = fd(q, k, v, None)
    finally:
        config.attn_mid = True
    assert _rel(o3, o2.detach().cpu()) < 1e-4


def _load(mod, z):
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
    missing = mod.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing


def test_attention_layer():
    dev = _dev()
    from layers.SelfAttention_Family import AttentionLayer, FullAttention
    z = _z("layer_attention_layer")
    al = AttentionLayer(FullAttention(False, attention_dropout=0.0), 8, int(z["H"])).to(dev)
    _load(al, z)
    x = _t(z["x"], dev).requires_grad_(True)
    o, _ = al(x, x, x, None)
    assert _rel(o, z["out"]) < 1e-4
    (o * _t(z["upstream"], dev)).sum().backward()
    assert _rel(x.grad, z["gx"]) < 2e-4
    for k, p in al.named_parameters():
        assert _rel(p.grad, z["g." + k]) < 2e-4, k


def test_encoder_stack():
    dev = _dev()
    from layers.SelfAttention_Family import AttentionLayer, FullAttention
    from layers.Transformer_EncDec import Encoder, EncoderLayer
    z = _z("layer_encoder")
    H = int(z["H"])
    enc = Encoder([EncoderLayer(AttentionLayer(FullAttention(False, attention_dropout=0.0), 8, H), 8, 16, dropout=0.0,
                                activation="gelu") for _ in range(2)], norm_layer=torch.nn.LayerNorm(8)).to(dev)
    _load(enc, z)
    x = _t(z["x"], dev).requires_grad_(True)
    o, attns = enc(x)
    assert len(attns) == 2 and _rel(o, z["out"]) < 1e-4
    (o * _t(z["upstream"], dev)).sum().backward()
    assert _rel(x.grad, z["gx"]) < 3e-4
    for k, p in enc.named_parameters():
        assert _rel(p.grad, z["g." + k]) < 3e-4, k


def test_patch_embedding():
    dev = _dev()
    from layers.Embed import PatchEmbedding
    z = _z("layer_patch_embedding")
    pe = PatchEmbedding(8, int(z["patch_len"]), int(z["stride"]), int(z["stride"]), 0.0).to(dev)
    pe.value_embedding.weight.data = _t(z["w"], dev)
    x = _t(z["x"], dev).requires_grad_(True)
    o, n_vars = pe(x)
    assert n_vars == int(z["n_vars"]) and _rel(o, z["out"]) < 1e-4
    (o * _t(z["upstream"], dev)).sum().backward()
    assert _rel(x.grad, z["gx"]) < 2e-4 and _rel(pe.value_embedding.weight.grad, z["gw"]) < 2e-4


def test_data_embedding():
    dev = _dev()
    from layers.Embed import DataEmbedding
    z = _z("layer_data_embedding")
    de = DataEmbedding(5, 8, dropout=0.0).to(dev)
    de.value_embedding.tokenConv.weight.data = _t(z["w"], dev)
    x = _t(z["x"], dev).requires_grad_(True)
    o = de(x, None)
    assert _rel(o, z["out"]) < 1e-4
    (o * _t(z["upstream"], dev)).sum().backward()
    assert _rel(x.grad, z["gx"]) < 2e-4 and _rel(de.value_embedding.tokenConv.weight.grad, z["gw"]) < 2e-4


@pytest.mark.parametrize("rows,dims", [(16384, (42, 32, 32, 1)), (512, (64, 32)), (1000, (19, 31, 7))])
def test_mlp_chain_vs_torch(rows, dims):
    """Linear (ReLU Linear)* on the HIP GEMM (ReLU mask in the next layer's dgrad epilogue, bias gradient as a ones
    column of the wgrad GEMM) against the same nn.Sequential in eager torch: fp32, 1e-4 / 2e-4."""
    dev = _dev()
    from immtsf import config, ops
    config.precision = "fp32"
    torch.manual_seed(rows)
    mods = []
    for i in range(len(dims) - 1):
        mods += [torch.nn.Linear(dims[i], dims[i + 1]), torch.nn.ReLU()]
    seq = torch.nn.Sequential(*mods[:-1]).to(dev)
    lins = [m for m in seq if isinstance(m, torch.nn.Linear)]
    x = torch.randn(rows // 8, 8, dims[0], device=dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    up = torch.randn(rows // 8, 8, dims[-1], device=dev)
    ref = seq(xa)
    (ref * up).sum().backward()
    ref_g = [p.grad.clone() for p in seq.parameters()]
    seq.zero_grad()
    out = ops.mlp(xb, [m.weight for m in lins], [m.bias for m in lins])
    (out * up).sum().backward()
    assert _rel(out, ref.detach().cpu()) < 1e-4
    assert _rel(xb.grad, xa.grad.cpu()) < 2e-4
    for p, g in zip(seq.parameters(), ref_g):
        assert _rel(p.grad, g.cpu()) < 2e-4
    # single Linear with the fused ReLU epilogue
    seq.zero_grad()
    xc = x.clone().requires_grad_(True)
    y = ops.linear(xc, lins[0].weight, lins[0].bias, relu=True)
    yr = torch.relu(torch.nn.functional.linear(x, lins[0].weight.detach(), lins[0].bias.detach()))
    assert _rel(y, yr.cpu()) < 1e-4
    y.sum().backward()
    assert _rel(lins[0].bias.grad, (yr > 0).float().reshape(-1, dims[1]).sum(0).cpu()) < 2e-4


def test_time2vec_vs_torch():
    dev = _dev()
    from immtsf import ops
    torch.manual_seed(3)
    for rows, d in [((64, 1, 32), 10), ((7,), 1), ((2048, 3), 33)]:
        lin0, lin = torch.nn.Linear(1, 1).to(dev), (torch.nn.Linear(1, d - 1).to(dev) if d > 1 else None)
        t = torch.rand(*rows, device=dev) * 3
        parts = [lin0(t.unsqueeze(-1))] + ([torch.sin(lin(t.unsqueeze(-1)))] if lin is not None else [])
        ref = torch.cat(parts, -1)
        up = torch.randn_like(ref)
        ps = list(lin0.parameters()) + (list(lin.parameters()) if lin is not None else [])
        gref = torch.autograd.grad((ref * up).sum(), ps)
        out = ops.time2vec(t, lin0.weight, lin0.bias, lin.weight if lin is not None else None,
                           lin.bias if lin is not None else None)
        g = torch.autograd.grad((out * up).sum(), ps)
        assert out.shape == ref.shape and _rel(out, ref.detach().cpu()) < 1e-5
        for a, b in zip(g, gref):
            assert _rel(a, b.cpu()) < 2e-4
    with pytest.raises(RuntimeError):
        ops.time2vec(t.requires_grad_(True), lin0.weight, lin0.bias, None, None)


def test_packed_qkv_attention_equals_sliced():
    """full_attention_qkv (strided reads of a packed in-projection, packed gradient) == full_attention on the slices"""
    dev = _dev()
    from immtsf import config, ops
    config.precision = "fp32"
    torch.manual_seed(4)
    for B, L, H, E in [(512, 2, 1, 32), (5, 19, 3, 8), (7, 8, 2, 24), (3, 1, 1, 64), (3, 2, 1, 5)]:      # L <= 8, E % 4 == 0, E <= 64: the one-thread-per-row kernel
        qkv = torch.randn(B, L, 3, H, E, device=dev)
        a, b = qkv.clone().requires_grad_(True), qkv.clone().requires_grad_(True)
        up = torch.randn(B, L, H, E, device=dev)
        o1 = ops.full_attention(a[:, :, 0], a[:, :, 1], a[:, :, 2], E ** -0.5)
        o2 = ops.full_attention_qkv(b, E ** -0.5)
        (o1 * up).sum().backward()
        (o2 * up).sum().backward()
        assert _rel(o2, o1.detach().cpu()) < 1e-6 and _rel(b.grad, a.grad.cpu()) < 1e-6


def test_short_attention_causal_and_dropout_equal_the_gemm_path():
    """L <= ATTN_SHORT_MAX routes full_attention_qkv to the one-launch kernel: causal masking and train-mode dropout (same
    Philox site / index stream) give what the batched-GEMM + row-softmax formulation gives, forward and backward."""
    dev = _dev()
    from immtsf import config, ops
    config.precision = "fp32"
    torch.manual_seed(9)
    for causal, p in [(True, 0.0), (False, 0.3), (True, 0.25)]:
        B, L, H, E = 33, 6, 2, 16
        qkv = torch.randn(B, L, 3, H, E, device=dev)
        a, b = qkv.clone().requires_grad_(True), qkv.clone().requires_grad_(True)
        up = torch.randn(B, L, H, E, device=dev)
        o1 = ops.FullAttentionQKVFn.apply(a, E ** -0.5, p, True, 77, 21, causal, 0)
        o2 = ops.full_attention_qkv(b, E ** -0.5, p, True, 77, 21, causal)
        (o1 * up).sum().backward()
        (o2 * up).sum().backward()
        assert _rel(o2, o1.detach().cpu()) < 1e-5 and _rel(b.grad, a.grad.cpu()) < 1e-5, (causal, p)


# ---- PatchTST-size layers (d_model 512, 2 heads, d_ff 2048, 10 patches, 48 rows) and TimeLLM's ReprogrammingLayer against
# fixtures generated from the real reference (tests/golden/make_golden.py:gen_layers_big; weights regenerated from
# tests/golden/seeded.py), fp32 mode at 1e-4 / 2e-4 and bf16 mode at 3e-2 / 4e-2 (relative L2 for bf16) ------------------
def _big(tag):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_golden import _big_case
    return _big_case(tag)


def _build_big(tag, dev):
    from layers.Embed import PatchEmbedding
    from layers.SelfAttention_Family import AttentionLayer, FullAttention
    from layers.Transformer_EncDec import EncoderLayer
    seeded, x, sd, seed = _big(tag)
    D, H, DFF = 512, 2, 2048
    if tag == "attention_layer":
        m = AttentionLayer(FullAttention(False, attention_dropout=0.0), D, H)
        run = lambda xx: m(xx, xx, xx, None)[0]      # noqa: E731
    elif tag == "encoder_layer":
        m = EncoderLayer(AttentionLayer(FullAttention(False, attention_dropout=0.0), D, H), D, DFF, dropout=0.0, activation="gelu")
        run = lambda xx: m(xx)[0]      # noqa: E731
    elif tag == "patch_embedding":
        m = PatchEmbedding(D, 18, 9, 9, 0.0)
        run = lambda xx: m(xx)[0]      # noqa: E731
    else:
        from models.TimeLLM import ReprogrammingLayer
        m = ReprogrammingLayer(16, 8, d_llm=768, attention_dropout=0.0)
        src = torch.from_numpy(seeded.rand((1000, 768), 541)).to(dev)
        run = lambda xx: m(xx, src, src)      # noqa: E731
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m.to(dev).train(), run, seeded, x, seed


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("tag", ["attention_layer", "encoder_layer", "patch_embedding", "reprogramming"])
def test_layers_patchtst_size_vs_reference_golden(tag, precision):
    dev = _dev()
    from immtsf import config
    z = _z("layer_big_" + tag)
    config.precision = precision
    try:
        m, run, seeded, x, seed = _build_big(tag, dev)
        xx = torch.from_numpy(x).to(dev).requires_grad_(True)
        o = run(xx)
        up = torch.from_numpy(seeded.rand(tuple(o.shape), seed + 1000)).to(dev)
        (o * up).sum().backward()
        torch.cuda.synchronize()
    finally:
        config.precision = "fp32"
    out, gx = o.detach().cpu().numpy().astype(np.float64), xx.grad.cpu().numpy().astype(np.float64)
    grads = {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}
    if precision == "fp32":
        tol_o, tol_g = 1e-4, 2e-4
        assert np.abs(out[:8] - z["out8"]).max() <= tol_o * np.abs(z["out8"]).max()
        assert np.abs(gx[:8] - z["gx8"]).max() <= tol_g * np.abs(z["gx8"]).max()
    else:
        tol_o, tol_g = 3e-2, 4e-2
        assert np.linalg.norm(out[:8] - z["out8"]) <= tol_o * np.linalg.norm(z["out8"])
        assert np.linalg.norm(gx[:8] - z["gx8"]) <= tol_g * np.linalg.norm(z["gx8"])
    assert abs(np.linalg.norm(out) / float(z["out_norm"]) - 1.0) < tol_o
    assert abs(np.linalg.norm(gx) / float(z["gx_norm"]) - 1.0) < tol_g
    gmax = max(float(z["probe." + k][-1]) for k in grads)
    for i, k in enumerate(sorted(grads)):
        want = z["probe." + k]
        got = seeded.probes(grads[k], seed + 2000 + i)
        # a random projection of a gradient with relative error eps deviates by ~ eps * norm
        assert np.abs(got - want).max() <= 4 * tol_g * max(want[-1], 1e-2 * gmax), (tag, precision, k, got, want)


def test_patch_and_token_embedding_dropout_and_input_gradient():
    """the one-kernel embeddings: the input gradient (not needed by any configured backbone, but part of the op) equals
    autograd through the eager formulation, and with dropout on the kept fraction matches p and the output is the
    dropout-free output times the exported mask / (1 - p)"""
    dev = _dev()
    from immtsf import config, ops
    from oracle import layers_ref as L
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 4, 50, generator=g).to(dev).requires_grad_(True)
    W = (torch.randn(96, 12, generator=g) * 0.3).to(dev).requires_grad_(True)
    pe = L.sinusoid(200, 96, dev)
    o = ops.patch_embed(x, W, pe, 12, 5, 5)
    ref = L.patch_embedding(W.detach().clone().requires_grad_(True), x.detach().clone().requires_grad_(True), 12, 5, 5)
    up = torch.randn(o.shape, generator=g).to(dev)
    (o * up).sum().backward()
    xr, wr = x.detach().clone().requires_grad_(True), W.detach().clone().requires_grad_(True)
    (L.patch_embedding(wr, xr, 12, 5, 5) * up).sum().backward()
    assert _rel(o, ref.detach().cpu()) < 1e-5 and _rel(x.grad, xr.grad.cpu()) < 1e-4 and _rel(W.grad, wr.grad.cpu()) < 1e-4
    xt = torch.randn(3, 20, 7, generator=g).to(dev).requires_grad_(True)
    Wt = (torch.randn(32, 7, 3, generator=g) * 0.3).to(dev).requires_grad_(True)
    pe2 = L.sinusoid(64, 32, dev)
    o = ops.token_embed(xt, Wt, pe2)
    up = torch.randn(o.shape, generator=g).to(dev)
    (o * up).sum().backward()
    xr, wr = xt.detach().clone().requires_grad_(True), Wt.detach().clone().requires_grad_(True)
    r = L.data_embedding(wr, xr)
    (r * up).sum().backward()
    assert _rel(o, r.detach().cpu()) < 1e-5 and _rel(xt.grad, xr.grad.cpu()) < 1e-4 and _rel(Wt.grad, wr.grad.cpu()) < 1e-4
    config.manual_seed(9)
    od = ops.patch_embed(x.detach(), W.detach(), pe, 12, 5, 5, p_drop=0.25, training=True)
    o0 = ops.patch_embed(x.detach(), W.detach(), pe, 12, 5, 5)
    kept = (od != 0)
    frac = float(kept.float().mean())
    assert abs(frac - 0.75) < 0.02, frac
    assert _rel(od[kept], (o0[kept] / 0.75).cpu()) < 1e-5


# ------------------------------------------------------------------------------------------ fused encoder layer block
def _l2err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-3 * max(1.0, b.numel() ** 0.5)))


@pytest.mark.parametrize("S,D,H,F,precision", [(2, 32, 1, 2048, "fp32"), (5, 48, 4, 96, "fp32"), (2, 32, 1, 2048, "bf16")])
def test_encoder_layer_block_vs_torch_layer(S, D, H, F, precision):
    """immtsf_encoder_layer_forward/backward against torch's own nn.TransformerEncoderLayer evaluated in float64 on the CPU
    (dropout 0): tPatchGNN's shape (S = 2 patches, d_model 32, dim_feedforward 2048) and an odd multi-head one."""
    dev = _dev()
    from immtsf import config, ops
    torch.manual_seed(3)
    ref = torch.nn.TransformerEncoderLayer(d_model=D, nhead=H, dim_feedforward=F, dropout=0.0, batch_first=True).double()
    lyr = torch.nn.TransformerEncoderLayer(d_model=D, nhead=H, dim_feedforward=F, dropout=0.0, batch_first=True).to(dev)
    lyr.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    x = torch.randn(37, S, D)
    up = torch.randn(37, S, D)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    (yr * up.double()).sum().backward()
    config.precision = precision
    try:
        xg = x.to(dev).requires_grad_(True)
        y = ops.encoder_layer(lyr, xg, True, ops.SITE_LAYER_BASE + 64)
        (y * up.to(dev)).sum().backward()
    finally:
        config.precision = "fp32"
    err = _rel if precision == "fp32" else _l2err
    tol_o, tol_g = (1e-4, 3e-4) if precision == "fp32" else (3e-2, 4e-2)
    errs = {"out": err(y, yr.detach().float()), "dx": err(xg.grad, xr.grad.float())}
    for (k, p), (_, q) in zip(lyr.named_parameters(), ref.named_parameters()):
        errs["g." + k] = err(p.grad, q.grad.float())
    assert errs.pop("out") <= tol_o
    bad = {k: v for k, v in errs.items() if not v <= tol_g}
    assert not bad, bad


def test_encoder_layer_block_many_rows_bf16():
    """tPatchGNN's layer at many windows (5003 sequences of 2 patches = 10 006 rows, d_model 32, bf16): the paths that only exist
    there -- the feed-forward without its intermediate (csrc/ffn32.hip) and the streaming weight gradients of the attention
    projections (csrc/skinny_tn.hip) -- against torch's layer in float64, all outputs and parameter gradients at the bf16 bars."""
    dev = _dev()
    from immtsf import config, ops
    S, D, H, F, Bs = 2, 32, 1, 2048, 5003
    torch.manual_seed(8)
    ref = torch.nn.TransformerEncoderLayer(d_model=D, nhead=H, dim_feedforward=F, dropout=0.0, batch_first=True).double()
    lyr = torch.nn.TransformerEncoderLayer(d_model=D, nhead=H, dim_feedforward=F, dropout=0.0, batch_first=True).to(dev)
    lyr.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    x = torch.randn(Bs, S, D)
    up = torch.randn(Bs, S, D)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    (yr * up.double()).sum().backward()
    config.precision = "bf16"
    try:
        xg = x.to(dev).requires_grad_(True)
        y = ops.encoder_layer(lyr, xg, True, ops.SITE_LAYER_BASE + 64)
        (y * up.to(dev)).sum().backward()
    finally:
        config.precision = "fp32"
    errs = {"out": _l2err(y, yr.detach().float()), "dx": _l2err(xg.grad, xr.grad.float())}
    for (k, p), (_, q) in zip(lyr.named_parameters(), ref.named_parameters()):
        errs["g." + k] = _l2err(p.grad, q.grad.float())
    assert errs.pop("out") <= 3e-2
    # (5e-2: the first feed-forward weight's gradient goes through two ReLU masks of bf16 pre-activations -- the ones within rounding
    # of zero flip against the float64 reference -- it sits at 4.0e-2 with or without the two many-row kernels)
    bad = {k: v for k, v in errs.items() if not v <= 5e-2}
    assert not bad, bad


def test_encoder_layer_block_dropout_masks():
    """training mode, p = 0.3 on all four sites: the block must equal a torch composition that uses the exported Philox
    keep-masks (attention weights, dropout1, feed-forward dropout, dropout2), forward and backward, fp32 1e-4 / 3e-4."""
    dev = _dev()
    from immtsf import config, ops
    Bs, S, D, H, F, p = 29, 3, 32, 2, 64, 0.3
    torch.manual_seed(5)
    config.manual_seed(99)
    lyr = torch.nn.TransformerEncoderLayer(d_model=D, nhead=H, dim_feedforward=F, dropout=p, batch_first=True).to(dev).train()
    x = torch.randn(Bs, S, D, device=dev, requires_grad=True)
    up = torch.randn(Bs, S, D, device=dev)
    base = ops.SITE_LAYER_BASE + 64
    cnt0 = config._counter
    y = ops.encoder_layer(lyr, x, True, base)
    (y * up).sum().backward()
    got = {"dx": x.grad.clone(), **{k: q.grad.clone() for k, q in lyr.named_parameters()}}
    # the seed the call drew: replay config.next_seed() from the same counter
    config._counter = cnt0
    seed = config.next_seed()
    keep = lambda site, shape: ops.dropout_keep_mask(seed, site, int(np.prod(shape)), p, dev).view(*shape).double() / (1 - p)  # noqa: E731
    m_att, m1, mf, m2 = keep(base, (Bs, H, S, S)), keep(base + 1, (Bs * S, D)), keep(base + 2, (Bs * S, F)), keep(base + 3, (Bs * S, D))
    P = {k: q.detach().double().requires_grad_(True) for k, q in lyr.named_parameters()}
    xd = x.detach().double().requires_grad_(True)
    E = D // H
    qkv = (xd.reshape(-1, D) @ P["self_attn.in_proj_weight"].T + P["self_attn.in_proj_bias"]).view(Bs, S, 3, H, E)
    q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)
    A = torch.softmax(q @ k.transpose(-1, -2) / E ** 0.5, -1) * m_att
    a = (A @ v).transpose(1, 2).reshape(Bs * S, D)
    sa = a @ P["self_attn.out_proj.weight"].T + P["self_attn.out_proj.bias"]
    ln = torch.nn.functional.layer_norm
    x1 = ln(xd.reshape(-1, D) + sa * m1, (D,), P["norm1.weight"], P["norm1.bias"], lyr.norm1.eps)
    h = torch.relu(x1 @ P["linear1.weight"].T + P["linear1.bias"]) * mf
    ff = h @ P["linear2.weight"].T + P["linear2.bias"]
    out = ln(x1 + ff * m2, (D,), P["norm2.weight"], P["norm2.bias"], lyr.norm2.eps).view(Bs, S, D)
    (out * up.double()).sum().backward()
    assert _rel(y, out.detach().float()) <= 1e-4
    errs = {"dx": _rel(got["dx"], xd.grad.float())}
    for k2, q2 in P.items():
        errs[k2] = _rel(got[k2], q2.grad.float())
    bad = {k2: v2 for k2, v2 in errs.items() if not v2 <= 3e-4}
    assert not bad, bad


@pytest.mark.parametrize("R,p", [(6007, 0.0), (6007, 0.3), (2048, 0.3), (40000, 0.1), (1024, 0.1), (1000, 0.0), (300, 0.3), (4096, 0.2)])
def test_ffn_block_many_rows_fused_vs_torch(R, p):
    """The many-rows feed-forward of tPatchGNN's encoder layer (csrc/ffn32.hip: d_model 32, dim_feedforward 2048, ReLU, bf16,
    no (R x F) intermediate; reference models/tPatchGNN.py:118-121) against the float64 torch composition that uses the
    exported Philox keep-masks -- forward, data gradient and all parameter gradients at the bf16 bars; ragged row counts
    (not a multiple of the 16-row tiles / 128-row sub-blocks) and the three rows-per-wave variants."""
    dev = _dev()
    from immtsf import config, ops
    D, F = 32, 2048
    torch.manual_seed(17)
    config.manual_seed(23)
    conv1, conv2 = torch.nn.Conv1d(D, F, 1).to(dev), torch.nn.Conv1d(F, D, 1).to(dev)
    n2 = torch.nn.LayerNorm(D).to(dev)
    with torch.no_grad():
        n2.weight.uniform_(0.5, 1.5)
        n2.bias.uniform_(-0.3, 0.3)
        conv1.bias.uniform_(-0.2, 0.2)
    x = torch.randn(1, R, D, device=dev, requires_grad=True)
    up = torch.randn(1, R, D, device=dev)
    base = ops.SITE_LAYER_BASE + 192
    c0 = config._counter
    config.precision = "bf16"
    try:
        y = ops.ffn_block(x, conv1, conv2, n2, "relu", p, True, base)
        (y * up).sum().backward()
    finally:
        config.precision = "fp32"
    config._counter = c0
    seed = config.next_seed() if p > 0 else 0
    keep = lambda site, shape: (ops.dropout_keep_mask(seed, site, int(np.prod(shape)), p, dev).view(*shape).double() / (1 - p)  # noqa: E731
                                if p > 0 else torch.ones(*shape, dtype=torch.float64, device=dev))
    mh, mo = keep(base, (R, F)), keep(base + 1, (R, D))
    P = {k: q.detach().double().requires_grad_(True) for k, q in (("w1", conv1.weight), ("b1", conv1.bias), ("w2", conv2.weight),
                                                                  ("b2", conv2.bias), ("g2", n2.weight), ("be2", n2.bias))}
    xd = x.detach().double().reshape(R, D).requires_grad_(True)
    h = torch.relu(xd @ P["w1"].squeeze(-1).T + P["b1"]) * mh
    ff = h @ P["w2"].squeeze(-1).T + P["b2"]
    out = torch.nn.functional.layer_norm(xd + ff * mo, (D,), P["g2"], P["be2"], n2.eps)
    (out * up.double().reshape(R, D)).sum().backward()
    assert _l2err(y.reshape(R, D), out.detach().float()) <= 3e-2
    errs = {"dx": _l2err(x.grad.reshape(R, D), xd.grad.float()),
            "w1": _l2err(conv1.weight.grad, P["w1"].grad.float()), "b1": _l2err(conv1.bias.grad, P["b1"].grad.float()),
            "w2": _l2err(conv2.weight.grad, P["w2"].grad.float()), "b2": _l2err(conv2.bias.grad, P["b2"].grad.float()),
            "g2": _l2err(n2.weight.grad, P["g2"].grad.float()), "be2": _l2err(n2.bias.grad, P["be2"].grad.float())}
    bad = {k: v for k, v in errs.items() if not v <= 4e-2}
    assert not bad, bad


@pytest.mark.parametrize("act,p", [("gelu", 0.0), ("gelu", 0.3), ("relu", 0.3)])
def test_ffn_block_and_residual_layernorm_vs_torch(act, p):
    """ops.residual_layer_norm + ops.ffn_block (the two joints of layers.Transformer_EncDec.EncoderLayer) against a float64
    torch composition that uses the exported Philox keep-masks: GELU/ReLU in the GEMM epilogue, the activation derivative
    and the regenerated dropout mask in the data-gradient epilogue; fp32 1e-4 / 3e-4."""
    dev = _dev()
    from immtsf import config, ops
    R, D, F = 150, 48, 112
    torch.manual_seed(11)
    config.manual_seed(5)
    conv1, conv2 = torch.nn.Conv1d(D, F, 1).to(dev), torch.nn.Conv1d(F, D, 1).to(dev)
    n1, n2 = torch.nn.LayerNorm(D).to(dev), torch.nn.LayerNorm(D).to(dev)
    with torch.no_grad():
        for n in (n1, n2):
            n.weight.uniform_(0.5, 1.5)
            n.bias.uniform_(-0.3, 0.3)
    x = torch.randn(6, 25, D, device=dev, requires_grad=True)
    br = torch.randn(6, 25, D, device=dev, requires_grad=True)
    up = torch.randn(6, 25, D, device=dev)
    base = ops.SITE_LAYER_BASE + 128
    c0 = config._counter
    x1 = ops.residual_layer_norm(x, br, n1, p, True, base)
    y = ops.ffn_block(x1, conv1, conv2, n2, act, p, True, base + 1)
    (y * up).sum().backward()
    config._counter = c0
    s1 = config.next_seed() if p > 0 else 0
    s2 = config.next_seed() if p > 0 else 0
    keep = lambda seed, site, shape: (ops.dropout_keep_mask(seed, site, int(np.prod(shape)), p, dev).view(*shape).double() / (1 - p)  # noqa: E731
                                      if p > 0 else torch.ones(*shape, dtype=torch.float64, device=dev))
    m1, mh, mo = keep(s1, base, (R, D)), keep(s2, base + 1, (R, F)), keep(s2, base + 2, (R, D))
    P = {k: q.detach().double().requires_grad_(True) for k, q in (("w1", conv1.weight), ("b1", conv1.bias), ("w2", conv2.weight),
                                                                  ("b2", conv2.bias), ("g1", n1.weight), ("be1", n1.bias),
                                                                  ("g2", n2.weight), ("be2", n2.bias))}
    xd, bd = x.detach().double().reshape(R, D).requires_grad_(True), br.detach().double().reshape(R, D).requires_grad_(True)
    ln = torch.nn.functional.layer_norm
    a1 = ln(xd + bd * m1, (D,), P["g1"], P["be1"], n1.eps)
    pre = a1 @ P["w1"].squeeze(-1).T + P["b1"]
    h = (torch.nn.functional.gelu(pre) if act == "gelu" else torch.relu(pre)) * mh
    ff = h @ P["w2"].squeeze(-1).T + P["b2"]
    out = ln(a1 + ff * mo, (D,), P["g2"], P["be2"], n2.eps)
    (out * up.double().reshape(R, D)).sum().backward()
    assert _rel(y.reshape(R, D), out.detach().float()) <= 1e-4
    errs = {"dx": _rel(x.grad.reshape(R, D), xd.grad.float()), "dbranch": _rel(br.grad.reshape(R, D), bd.grad.float()),
            "w1": _rel(conv1.weight.grad, P["w1"].grad.float()), "b1": _rel(conv1.bias.grad, P["b1"].grad.float()),
            "w2": _rel(conv2.weight.grad, P["w2"].grad.float()), "b2": _rel(conv2.bias.grad, P["b2"].grad.float()),
            "g1": _rel(n1.weight.grad, P["g1"].grad.float()), "be1": _rel(n1.bias.grad, P["be1"].grad.float()),
            "g2": _rel(n2.weight.grad, P["g2"].grad.float()), "be2": _rel(n2.bias.grad, P["be2"].grad.float())}
    bad = {k: v for k, v in errs.items() if not v <= 3e-4}
    assert not bad, bad


@pytest.mark.parametrize("B,H,W,Cin,Cout,n,precision", [(3, 5, 4, 6, 8, 3, "fp32"), (64, 8, 8, 16, 32, 6, "fp32"), (64, 8, 8, 16, 32, 6, "bf16")])
def test_inception_block_as_one_merged_convolution(B, H, W, Cin, Cout, n, precision):
    """ops.inception_merge + ops.conv2d_same_cl (GELU) against layers.Conv_Blocks.Inception_Block_V1 evaluated by torch in
    float64 on the CPU (the mean of n same-padded convolutions), forward and every gradient; the second case is TimesNet's
    cfg4 shape (d_model 16 -> d_ff 32, six kernels up to 11 x 11, 64 windows x 64 positions)."""
    dev = _dev()
    from immtsf import config, ops
    from layers.Conv_Blocks import Inception_Block_V1
    torch.manual_seed(2)
    blk = Inception_Block_V1(Cin, Cout, num_kernels=n).to(dev)
    with torch.no_grad():
        for k in blk.kernels:
            k.bias.uniform_(-0.2, 0.2)
    ref = Inception_Block_V1(Cin, Cout, num_kernels=n).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in blk.state_dict().items()})
    x = torch.randn(B, H, W, Cin)
    up = torch.randn(B, H, W, Cout)
    xr = x.double().requires_grad_(True)
    yr = torch.nn.functional.gelu(ref(xr.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
    (yr * up.double()).sum().backward()
    config.precision = precision
    try:
        xg = x.to(dev).requires_grad_(True)
        Weff, beff, KS = ops.inception_merge(blk)
        y = ops.conv2d_same_cl(xg, Weff, beff, KS, act="gelu")
        (y * up.to(dev)).sum().backward()
    finally:
        config.precision = "fp32"
    err = _rel if precision == "fp32" else _l2err
    tol_o, tol_g = (1e-4, 3e-4) if precision == "fp32" else (3e-2, 4e-2)
    assert err(y, yr.detach().float()) <= tol_o
    errs = {"dx": err(xg.grad, xr.grad.float())}
    for (k, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        errs[k] = err(p.grad, q.grad.float())
    bad = {k: v for k, v in errs.items() if not v <= tol_g}
    assert not bad, bad
