"""GPU parity of the time-aware patch encoder and the tPatchGNN drop-in: fused HIP TE+TTCN kernel
vs. the golden vectors of the real reference and vs. the eager formulation at the benchmark dimensions.
Tolerance: 1e-4 outputs / 2e-4 gradients relative to max (fp32 both sides)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b, floor=1e-3):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / max(float(b.abs().max()), floor))


def _oracle_twin(m, args, dev):
    """oracle/tpatchgnn_ref.py (the plain-torch restatement, pinned against the reference's goldens in
    tests/test_oracle_golden.py) with the product module's weights, on the same device: the eager comparator"""
    from oracle.tpatchgnn_ref import TPatchGNNRef
    ref = TPatchGNNRef(args).to(dev)
    ref.load_state_dict(m.state_dict(), strict=True)
    return ref


def _golden_model(dev, encoder):
    from models.tPatchGNN import tPatchGNN
    z = np.load(os.path.join(GOLDEN, "model_tpatchgnn.npz"))
    args = types.SimpleNamespace(device=str(dev), hid_dim=8, C=3, npatch=2, nlayer=1, te_dim=4, n_heads=1, tf_layer=1,
                                 node_dim=4, hop=1, outlayer="Linear")
    m = tPatchGNN(args).to(dev)
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}, strict=True)
    return m.eval(), z


def test_ttcn_vs_reference_golden():
    dev = _dev()
    m, z = _golden_model(dev, "hip")
    X, tt, mask = [torch.from_numpy(z[k]).to(dev) for k in ("X", "tt", "mask")]
    B, M, L, N = X.shape
    flat = lambda t: t.permute(0, 3, 1, 2).reshape(B * N * M, L)    # noqa: E731
    h = m._encode_patches(flat(X), flat(tt), flat(mask))[:, :-1]
    assert _rel(h, torch.from_numpy(z["ttcn_out"])) < 1e-4
    (h * torch.from_numpy(z["ttcn_upstream"]).to(dev)).sum().backward()
    # the gradient of Filter_Generators.4.bias is a pure cancellation residue (sum_l softmax'(l) = 0 per column), so
    # every gradient is judged against the largest gradient magnitude of the block, not its own ~1e-6 noise floor
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("g_ttcn."))
    errs = {}
    for k, p in m.named_parameters():
        if f"g_ttcn.{k}" in z.files:
            errs[k] = _rel(p.grad, torch.from_numpy(z[f"g_ttcn.{k}"]), floor=1e-2 * gmax)
    assert len(errs) == 11, sorted(errs)
    bad = {k: v for k, v in errs.items() if v > 2e-4}
    assert not bad, bad


def test_tpatchgnn_forecasting_vs_reference_golden():
    dev = _dev()
    m, z = _golden_model(dev, "hip")
    out = m.forecasting(*[torch.from_numpy(z[k]).to(dev) for k in ("tp", "X", "tt", "mask")])
    assert out.shape == z["out"].shape
    assert _rel(out, torch.from_numpy(z["out"])) < 1e-4
    (out * torch.from_numpy(z["upstream"]).to(dev)).sum().backward()
    bad = {}
    for k, p in m.named_parameters():
        e = _rel(p.grad, torch.from_numpy(z["g." + k]))
        if e > 3e-4:
            bad[k] = e
    assert not bad, bad


@pytest.mark.parametrize("L", [32, 45])
def test_ttcn_benchmark_dims_vs_eager(L):
    """P = 64*8*2 patches, te_dim 10, hid_dim 32 (BASELINE cfg2); L = 45 exercises the 32-observation LDS chunking;
    30 % of the patches are empty."""
    dev = _dev()
    from models.tPatchGNN import tPatchGNN
    args = types.SimpleNamespace(device=str(dev), hid_dim=32, C=8, npatch=2, nlayer=1, te_dim=10, n_heads=1, tf_layer=1,
                                 node_dim=10, hop=1, outlayer="Linear")
    torch.manual_seed(0)
    m = tPatchGNN(args).to(dev)
    ref = _oracle_twin(m, args, dev)
    g = torch.Generator().manual_seed(1)
    P = 64 * 8 * 2
    cnt = torch.randint(1, L + 1, (P, 1), generator=g)
    cnt[torch.rand(P, 1, generator=g) < 0.3] = 0
    mask = (torch.arange(L).view(1, -1) < cnt).float().to(dev)
    x = torch.randn(P, L, generator=g).to(dev) * mask
    tt = torch.rand(P, L, generator=g).to(dev) * mask
    up = torch.randn(P, 32, generator=g).to(dev)
    h = m._encode_patches(x, tt, mask)
    (h * up).sum().backward()
    g_hip = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    h2 = ref.encode_patches(x, tt, mask)
    (h2 * up).sum().backward()
    assert _rel(h, h2) < 1e-4
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    bad = {k: _rel(g_hip[k], p.grad, floor=1e-2 * gmax) for k, p in ref.named_parameters() if p.grad is not None}
    bad = {k: v for k, v in bad.items() if v > 3e-4}
    assert not bad, bad


@pytest.mark.parametrize("name", ["PatchTST", "DLinear", "TimesNet"])
def test_backbone_wrappers_vs_reference_golden(name):
    """a15: Model(args).forecasting(tp_to_predict, observed_data, observed_tp, observed_mask) -> (B, Lp, C) for the
    other configured backbones, history shorter than input_len (padding path), B < args.batch_size."""
    dev = _dev()
    import importlib
    z = np.load(os.path.join(GOLDEN, f"model_{name.lower()}.npz"))
    cfg = types.SimpleNamespace(input_len=8, pred_len=6, d_model=8, d_ff=16, n_heads=2, e_layers=1, dropout=0.0, factor=5,
                                activation="gelu", enc_in=3, c_out=3, batch_size=4, device=str(dev), moving_avg=5, top_k=2,
                                num_kernels=2, embed="fixed", freq="h")
    m = getattr(importlib.import_module(f"models.{name}"), name)(cfg).to(dev)
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}, strict=True)
    m.train()
    out = m.forecasting(*[torch.from_numpy(z[k]).to(dev) for k in ("tpp", "data", "tp", "mask")])
    assert out.shape == z["out"].shape
    assert _rel(out, torch.from_numpy(z["out"])) < 1e-4
    (out * torch.from_numpy(z["upstream"]).to(dev)).sum().backward()
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("g."))
    bad = {}
    for k, p in m.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        e = _rel(g, torch.from_numpy(z["g." + k]), floor=1e-2 * gmax)
        if e > 3e-4:
            bad[k] = e
    assert not bad, bad


def test_timesnet_bf16_period_images_vs_fp32():
    """TimesNet at the cfg4 dimensions (d_model 16, d_ff 32, top_k 5, 2 layers, 64 windows): the bf16 mode's period images -- im2col
    written as bf16, the products on the bf16-in-HBM kernels with the device-side row count (csrc/conv.hip conv2d_period_*) -- against
    the exact-fp32 mode of the same module (which the reference golden pins): forecast and every parameter gradient inside the bf16
    band; and forecasting() reads nothing on the host (the period selection stays on the device): it runs under hipGraph capture."""
    dev = _dev()
    from immtsf import config
    from models.TimesNet import TimesNet
    cfg = types.SimpleNamespace(input_len=32, pred_len=32, d_model=16, d_ff=32, n_heads=2, e_layers=2, dropout=0.0, factor=5,
                                activation="gelu", enc_in=8, c_out=8, batch_size=64, device=str(dev), moving_avg=5, top_k=5,
                                num_kernels=6, embed="fixed", freq="h")
    torch.manual_seed(0)
    m = TimesNet(cfg).to(dev).train()
    g = torch.Generator().manual_seed(4)
    data = torch.randn(64, 32, 8, generator=g).to(dev)
    mask = (torch.rand(64, 32, 8, generator=g) < 0.7).float().to(dev)
    tp = torch.sort(torch.rand(64, 32, generator=g), 1).values.to(dev)
    tpp = torch.sort(torch.rand(64, 32, generator=g), 1).values.to(dev)
    up = torch.randn(64, 32, 8, generator=g).to(dev)
    res = {}
    try:
        for prec in ("fp32", "bf16"):
            config.precision = prec
            m.zero_grad()
            out = m.forecasting(tpp, data * mask, tp, mask)
            (out * up).sum().backward()
            res[prec] = (out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        # no host read inside forecasting(): capturable
        config.precision = "bf16"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            m.forecasting(tpp, data * mask, tp, mask)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr), torch.no_grad():
            o2 = m.forecasting(tpp, data * mask, tp, mask)
        gr.replay()
        torch.cuda.synchronize()
        assert _rel(o2, res["bf16"][0]) < 1e-5
    finally:
        config.precision = "fp32"
    assert _rel(res["bf16"][0], res["fp32"][0]) < 3e-2
    gmax = max(float(v.abs().max()) for v in res["fp32"][1].values())
    bad = {k: _rel(res["bf16"][1][k], v, floor=1e-2 * gmax) for k, v in res["fp32"][1].items()}
    bad = {k: v for k, v in bad.items() if v > 5e-2}
    assert not bad, bad


@pytest.mark.parametrize("Cin,Cout,n,B,total,periods", [
    (32, 16, 6, 5, 64, [2, 3, 7, 64, 33]),       # cfg4's second convolution; images of 4 and 5 row tiles
    (16, 32, 6, 3, 64, [5, 9, 13, 17, 31]),      # cfg4's first (two taps per k-step); 5 and 6 row tiles
    (8, 40, 2, 4, 50, [4, 7, 11]),               # four taps per k-step, three column tiles of four
    (24, 40, 3, 2, 60, [1, 60, 8]),              # channel counts that are not powers of two; one-column and one-row images
    (64, 64, 4, 2, 64, [6, 64, 50, 63]),         # the widest case: four column tiles, 8 row tiles
])
def test_inception_periods_implicit_vs_conv2d(Cin, Cout, n, B, total, periods):
    """csrc/conv.hip conv_period_mfma_kernel (ops.inception_periods: an Inception_Block_V1 over the k period images of a TimesBlock, no
    im2col image) against torch's conv2d on the same images, layers/Conv_Blocks.py:8-31 + models/TimesNet.py:50-66: output (+ GELU),
    data gradient and the n kernels' gradients; bf16 operands, 3e-2 of the largest value."""
    dev = _dev()
    from immtsf import config, ops
    torch.manual_seed(Cin * 7 + Cout)
    block = types.SimpleNamespace(kernels=torch.nn.ModuleList([torch.nn.Conv2d(Cin, Cout, 2 * i + 1, padding=i) for i in range(n)]).to(dev))
    k, Lmax = len(periods), 2 * total
    lens = [-(-total // p) * p for p in periods]
    period = torch.tensor(periods, dtype=torch.int32, device=dev)
    rows = torch.tensor([B * l for l in lens], dtype=torch.int32, device=dev)
    assert ops.inception_periods_ok(block, Lmax, "bf16")
    x = torch.randn(k, Lmax * B, Cin, device=dev, requires_grad=True)
    up = torch.randn(k, Lmax * B, Cout, device=dev)
    y = ops.inception_periods(x, period, rows, block, B, Lmax, act="gelu", precision="bf16")
    valid = torch.zeros(k, Lmax * B, 1, device=dev)
    for j, l in enumerate(lens):
        valid[j, :l * B] = 1
    (y * up * valid).sum().backward()
    got = (y.detach() * valid, x.grad.clone() * valid, [c.weight.grad.clone() for c in block.kernels], [c.bias.grad.clone() for c in block.kernels])
    x.grad = None
    for c in block.kernels:
        c.weight.grad = c.bias.grad = None
    ref = torch.zeros(k, Lmax * B, Cout, device=dev)
    outs = []
    for j, (p, l) in enumerate(zip(periods, lens)):
        img = x[j, :l * B].view(l, B, Cin).permute(1, 2, 0).reshape(B, Cin, l // p, p)
        o = torch.nn.functional.gelu(torch.stack([c(img) for c in block.kernels], -1).mean(-1))
        outs.append(o.reshape(B, Cout, l).permute(2, 0, 1).reshape(l * B, Cout))
    loss = sum((o * up[j, :o.shape[0]]).sum() for j, o in enumerate(outs))
    loss.backward()
    for j, o in enumerate(outs):
        ref[j, :o.shape[0]] = o.detach()
    assert _rel(got[0], ref) < 3e-2
    assert _rel(got[1], x.grad * valid) < 3e-2
    gmax = max(float(c.weight.grad.abs().max()) for c in block.kernels)
    for i, c in enumerate(block.kernels):
        assert _rel(got[2][i], c.weight.grad, floor=1e-2 * gmax) < 3e-2, i
        assert _rel(got[3][i], c.bias.grad) < 3e-2, i


@pytest.mark.parametrize("B,N,M,D,nd,hop", [(64, 8, 2, 32, 10, 1), (3, 5, 3, 16, 4, 2), (2, 41, 2, 64, 10, 1), (1, 1, 1, 8, 2, 3)])
def test_fused_graph_stage_vs_eager(B, N, M, D, nd, hop):
    """the one-workgroup-per-cell adaptive-graph kernel (forward, and the recomputing backward with atomically
    accumulated parameter gradients) against the eager formulation in the same module; fp32, 1e-4 / 2e-4."""
    dev = _dev()
    from models.tPatchGNN import tPatchGNN
    torch.manual_seed(B * 100 + N)
    args = types.SimpleNamespace(device=str(dev), hid_dim=D, C=N, npatch=M, nlayer=1, te_dim=4, n_heads=1, tf_layer=1,
                                 node_dim=nd, hop=hop, outlayer="Linear")
    m = tPatchGNN(args).to(dev)
    ref = _oracle_twin(m, args, dev)
    rps = dict(ref.named_parameters())
    names = ["nodevec1", "nodevec2", "nodevec_gate1.0.0.weight", "nodevec_gate1.0.0.bias", "nodevec_gate2.0.0.weight",
             "nodevec_gate2.0.0.bias", "nodevec_linear1.0.weight", "nodevec_linear1.0.bias", "nodevec_linear2.0.weight",
             "nodevec_linear2.0.bias", "gconv.0.mlp.mlp.weight", "gconv.0.mlp.mlp.bias"]
    ps = dict(m.named_parameters())
    x = torch.randn(B, N, M, D, device=dev)
    up = torch.randn(B, N, M, D, device=dev)
    res = {}
    for mode in ("torch", "hip"):
        xi = x.clone().requires_grad_(True)
        out = m._graph_stage(0, xi) if mode == "hip" else ref.graph_stage(0, xi)
        (out * up).sum().backward()
        res[mode] = (out.detach(), xi.grad.clone(), {n: (ps if mode == "hip" else rps)[n].grad.clone() for n in names})
    assert _rel(res["hip"][0], res["torch"][0]) < 1e-4
    assert _rel(res["hip"][1], res["torch"][1]) < 2e-4
    gmax = max(float(g.abs().max()) for g in res["torch"][2].values())
    for n in names:
        assert _rel(res["hip"][2][n], res["torch"][2][n], floor=1e-3 * gmax) < 2e-4, n


def test_graph_stage_backward_from_saved_cells_equals_the_recomputing_one():
    """immtsf_tpatchgnn_gcn_backward_saved (the cells' intermediates left by immtsf_tpatchgnn_gcn_forward_saved: what a training forward of
    the module uses) against immtsf_tpatchgnn_gcn_backward (recomputes every cell from x): the same kernels behind the recompute, so dx
    and all twelve parameter gradients agree to round-off of the atomics' order; the forwards' outputs bit for bit."""
    dev = _dev()
    import ctypes as C
    from immtsf import _lib
    from immtsf.ops import _struct, ptr, stream_ptr
    lib = _lib.load()
    torch.manual_seed(5)
    B, N, M, D, nd, order = 7, 8, 2, 32, 10, 1
    shapes = [(N, nd), (nd, N), (1, D + nd), (1,), (1, D + nd), (1,), (nd, D), (nd,), (nd, D), (nd,), (D, (order + 1) * D), (D,)]
    params = [0.3 * torch.randn(*sh, device=dev) for sh in shapes]
    x, dout = torch.randn(B, N, M, D, device=dev), torch.randn(B, N, M, D, device=dev)
    ps = _struct(_lib.GCNParams, params)
    out_a, out_b = torch.empty_like(x), torch.empty_like(x)
    saved = torch.empty(lib.immtsf_tpatchgnn_gcn_saved_floats(B, N, M, D, nd, order), dtype=torch.float32, device=dev)
    assert saved.numel() == B * M * ((order + 2) * N * D + 4 * N * nd + 2 * N * N + 4 * N)
    _lib.check(lib.immtsf_tpatchgnn_gcn_forward(B, N, M, D, nd, order, ptr(x), C.byref(ps), ptr(out_a), stream_ptr()), "gcn_forward")
    _lib.check(lib.immtsf_tpatchgnn_gcn_forward_saved(B, N, M, D, nd, order, ptr(x), C.byref(ps), ptr(out_b), ptr(saved), stream_ptr()), "gcn_forward_saved")
    assert torch.equal(out_a, out_b)
    res = []
    for use_saved in (False, True):
        grads = [torch.zeros_like(t) for t in params]
        gs = _struct(_lib.GCNParams, grads)
        dx = torch.empty_like(x)
        if use_saved:
            _lib.check(lib.immtsf_tpatchgnn_gcn_backward_saved(B, N, M, D, nd, order, ptr(saved), C.byref(ps), ptr(dout), ptr(dx), C.byref(gs), stream_ptr()),
                       "gcn_backward_saved")
        else:
            _lib.check(lib.immtsf_tpatchgnn_gcn_backward(B, N, M, D, nd, order, ptr(x), C.byref(ps), ptr(dout), ptr(dx), C.byref(gs), stream_ptr()),
                       "gcn_backward")
        torch.cuda.synchronize()
        res.append([dx] + grads)
    for a, b in zip(*res):
        assert _rel(b, a, floor=1e-6) < 1e-5


def test_out_of_limit_shapes_take_the_tested_unfused_paths():
    """Shapes outside the fused kernels' limits -- more variables than one graph cell's LDS image holds (adaptive-graph stage), more
    than 8 patches per variable (transformer block) -- run the module's UNFUSED formulations: `_graph_stage`'s stock-torch branch and
    `_encoder_layer_hip` (per-op HIP GEMM / attention / LayerNorm).  Neither is reached by a BASELINE configuration; both are the
    product's behaviour for such inputs, so both are compared with the oracle here (eval mode: no dropout)."""
    dev = _dev()
    from immtsf.ops import encoder_layer_supported, gcn_adaptive_supported
    from models.tPatchGNN import tPatchGNN
    torch.manual_seed(11)
    B, N, M, D, nd = 2, 300, 12, 32, 10
    assert not gcn_adaptive_supported(N, D, nd, 1) and not encoder_layer_supported(M, D, 1)
    args = types.SimpleNamespace(device=str(dev), hid_dim=D, C=N, npatch=M, nlayer=1, te_dim=4, n_heads=1, tf_layer=1,
                                 node_dim=nd, hop=1, outlayer="Linear")
    m = tPatchGNN(args).to(dev).eval()
    ref = _oracle_twin(m, args, dev).eval()
    x = torch.randn(B, N, M, D, device=dev)
    up = torch.randn(B, N, M, D, device=dev)
    res = {}
    for mode, mod in (("hip", m), ("torch", ref)):
        xi = x.clone().requires_grad_(True)
        out = mod._graph_stage(0, xi) if mode == "hip" else mod.graph_stage(0, xi)
        (out * up).sum().backward()
        res[mode] = (out.detach(), xi.grad.clone())
    assert _rel(res["hip"][0], res["torch"][0]) < 1e-4 and _rel(res["hip"][1], res["torch"][1]) < 2e-4
    # the transformer over 12 patches: product `_transformer` (falls to `_encoder_layer_hip`) vs the oracle's stock encoder
    seq = torch.randn(B * 7, M, D, device=dev)
    ups = torch.randn(B * 7, M, D, device=dev)
    got, want = [], []
    for store, fn in ((got, lambda t: m._transformer(0, t)), (want, lambda t: ref.transformer_encoder[0](t))):
        si = seq.clone().requires_grad_(True)
        o = fn(si)
        (o * ups).sum().backward()
        store += [o.detach(), si.grad.clone()]
    assert _rel(got[0], want[0]) < 1e-4 and _rel(got[1], want[1]) < 3e-4


@pytest.mark.parametrize("L", [32, 45, 7])
def test_ttcn_on_chip_equals_streaming_bf16(L):
    """bf16 mode: the fused kernel pair (filter tile produced, normalised and pooled on chip; backward recomputes it)
    against the streaming formulation (filter tensor through HBM) -- same rounding points, so they must agree far
    inside the bf16 band -- and both against the fp32 eager formulation at the bf16 tolerance."""
    dev = _dev()
    from immtsf import _lib, config
    from models.tPatchGNN import tPatchGNN
    lib = _lib.load()
    args = types.SimpleNamespace(device=str(dev), hid_dim=32, C=8, npatch=2, nlayer=1, te_dim=10, n_heads=1, tf_layer=1,
                                 node_dim=10, hop=1, outlayer="Linear")
    torch.manual_seed(0)
    m = tPatchGNN(args).to(dev)
    ref = _oracle_twin(m, args, dev)
    g = torch.Generator().manual_seed(L)
    P = 64 * 8 * 2
    cnt = torch.randint(1, L + 1, (P, 1), generator=g)
    cnt[torch.rand(P, 1, generator=g) < 0.3] = 0
    mask = (torch.arange(L).view(1, -1) < cnt).float().to(dev)
    x = torch.randn(P, L, generator=g).to(dev) * mask
    tt = torch.rand(P, L, generator=g).to(dev) * mask
    up = torch.randn(P, 32, generator=g).to(dev)
    res = {}
    try:
        for mode, cfgbits, prec, enc in (("fused", 0, "bf16", "hip"), ("stream", 0x8000, "bf16", "hip"), ("eager", 0, "fp32", "torch")):
            lib.immtsf_debug_gemm_config(cfgbits, 0)
            config.precision = prec
            mod = m if enc == "hip" else ref
            mod.zero_grad()
            h = m._encode_patches(x, tt, mask) if enc == "hip" else ref.encode_patches(x, tt, mask)
            (h * up).sum().backward()
            res[mode] = (h.detach().clone(), {k: p.grad.clone() for k, p in mod.named_parameters() if p.grad is not None})
    finally:
        lib.immtsf_debug_gemm_config(0, 0)
        config.precision = "fp32"

    gnorm = max(float(g.double().norm()) for g in res["eager"][1].values())

    def l2(a, b):      # relative L2; gradients that are zero by construction (the filter bias: a per-column shift of a
        return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-3 * gnorm))   # softmax) count as noise
    assert l2(res["fused"][0], res["stream"][0]) < 2e-3
    assert l2(res["fused"][0], res["eager"][0]) < 3e-2
    assert res["fused"][1].keys() == res["stream"][1].keys() == res["eager"][1].keys()
    for k in res["eager"][1]:
        assert l2(res["fused"][1][k], res["stream"][1][k]) < 5e-3, k
        # vs fp32: inside the bf16 band, or no worse than the streaming bf16 path on cancellation-prone gradients
        assert l2(res["fused"][1][k], res["eager"][1][k]) < max(4e-2, 1.5 * l2(res["stream"][1][k], res["eager"][1][k])), k


@pytest.mark.parametrize("B,N,Lp,D,E,H", [(64, 8, 32, 32, 10, 32), (3, 5, 7, 32, 4, 32), (2, 41, 70, 24, 10, 32),
                                          (2, 3, 260, 32, 10, 32), (4, 6, 20, 48, 6, 32)])
def test_fused_decoder_vs_eager(B, N, Lp, D, E, H):
    """models/tPatchGNN.py:283-291: decoder(cat[h repeated over Lp ; te repeated over N]) as one kernel per direction
    equals the eager nn.Sequential on the materialised (B, N, Lp, D+E) tensor -- outputs, data gradients and every
    parameter gradient (fp32 both sides; shapes include several row chunks per window, Lp > 256 and D != H)."""
    dev = _dev()
    from immtsf.ops import tpatch_decoder, tpatch_decoder_supported
    torch.manual_seed(B * 1000 + Lp)
    dec = torch.nn.Sequential(torch.nn.Linear(D + E, H), torch.nn.ReLU(inplace=True), torch.nn.Linear(H, H),
                              torch.nn.ReLU(inplace=True), torch.nn.Linear(H, 1)).to(dev)
    assert tpatch_decoder_supported(dec, N, Lp, D, E)
    h = torch.randn(B, N, D, device=dev, requires_grad=True)
    te = torch.randn(B, Lp, E, device=dev, requires_grad=True)
    up = torch.randn(B, Lp, N, device=dev)
    out = tpatch_decoder(dec, h, te)
    (out * up).sum().backward()
    got = [h.grad.clone(), te.grad.clone()] + [p.grad.clone() for p in dec.parameters()]
    h.grad = te.grad = None
    dec.zero_grad()
    x = torch.cat([h.unsqueeze(2).expand(B, N, Lp, D), te.unsqueeze(1).expand(B, N, Lp, E)], dim=-1)
    ref = dec(x.double().float()).squeeze(-1).permute(0, 2, 1)
    (ref * up).sum().backward()
    want = [h.grad, te.grad] + [p.grad for p in dec.parameters()]
    assert out.shape == (B, Lp, N) and _rel(out, ref) < 1e-5
    for i, (a, b) in enumerate(zip(got, want)):
        assert _rel(a, b) < 2e-5, i


@pytest.mark.parametrize("B,N,Lp,D,E", [(64, 8, 32, 32, 10), (3, 5, 7, 32, 4), (2, 41, 70, 24, 10), (300, 8, 32, 32, 10), (5, 3, 260, 32, 10)])
def test_fused_decoder_bf16_mfma_vs_eager(B, N, Lp, D, E):
    """bf16 mode: the decoder's second layer and its gradients on MFMA tiles (csrc/decoder.hip dec_fwd/bwd_mfma_kernel; persistent
    workgroups that walk several windows at B = 300) against the eager fp32 nn.Sequential at the bf16 bars -- ragged tile
    counts (rows not a multiple of 16 / 32), several variables per tile (Lp < 16) and Lp > 256."""
    dev = _dev()
    from immtsf.ops import tpatch_decoder, tpatch_decoder_supported
    H = 32
    torch.manual_seed(B * 77 + Lp)
    dec = torch.nn.Sequential(torch.nn.Linear(D + E, H), torch.nn.ReLU(inplace=True), torch.nn.Linear(H, H),
                              torch.nn.ReLU(inplace=True), torch.nn.Linear(H, 1)).to(dev)
    assert tpatch_decoder_supported(dec, N, Lp, D, E)
    h = torch.randn(B, N, D, device=dev, requires_grad=True)
    te = torch.randn(B, Lp, E, device=dev, requires_grad=True)
    up = torch.randn(B, Lp, N, device=dev)
    out = tpatch_decoder(dec, h, te, precision="bf16")
    (out * up).sum().backward()
    got = [h.grad.clone(), te.grad.clone()] + [p.grad.clone() for p in dec.parameters()]
    h.grad = te.grad = None
    dec.zero_grad()
    x = torch.cat([h.unsqueeze(2).expand(B, N, Lp, D), te.unsqueeze(1).expand(B, N, Lp, E)], dim=-1)
    l2 = lambda a, b: float((a.detach().double() - b.detach().double()).norm() / b.detach().double().norm().clamp_min(1e-12))      # noqa: E731

    def run(bf16_operands):
        h.grad = te.grad = None
        dec.zero_grad()
        h1 = torch.relu(dec[0](x))
        if bf16_operands:      # what the kernel computes: second-layer operands rounded to bf16, fp32 accumulation
            z2 = h1.bfloat16().float() @ dec[2].weight.bfloat16().float().T + dec[2].bias
        else:
            z2 = dec[2](h1)
        ref = dec[4](torch.relu(z2)).squeeze(-1).permute(0, 2, 1)
        (ref * up).sum().backward()
        return ref, [h.grad.clone(), te.grad.clone()] + [p.grad.clone() for p in dec.parameters()]

    # against the same arithmetic in torch (the ReLU masks agree): tight; against plain fp32 (masks of pre-activations within
    # bf16 rounding of zero flip, each flip is an O(1) error of that element's gradient): the bf16-mode bar of the step tests
    ref_e, want_e = run(True)
    assert l2(out, ref_e) < 2e-3
    errs = {i: l2(a, b) for i, (a, b) in enumerate(zip(got, want_e))}
    assert all(e < 1.5e-2 for e in errs.values()), errs
    ref_f, want_f = run(False)
    assert l2(out, ref_f) < 2e-2
    errs = {i: l2(a, b) for i, (a, b) in enumerate(zip(got, want_f))}
    assert all(e < 1e-1 for e in errs.values()), errs

@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("B,N,Lp,D,E", [(64, 8, 32, 32, 10), (3, 5, 7, 32, 4), (300, 8, 32, 32, 10), (2, 3, 260, 32, 10), (4, 6, 20, 48, 2)])
def test_decoder_with_learnable_te_inside_equals_the_two_ops(precision, B, N, Lp, D, E):
    """tpatch_decoder_te (LearnableTE of the prediction times built while a window is staged, its gradient reduced to the four
    parameter gradients in the backward kernel; models/tPatchGNN.py:176-180 + :283-291) against time2vec + tpatch_decoder, the same
    kernels with te as a tensor: output, d h, the decoder's gradients identical in arithmetic; the time-embedding gradients are the
    same sums in another order.  Gradients ACCUMULATE into what the buffers hold (the patch encoder shares them)."""
    dev = _dev()
    from immtsf.ops import time2vec, tpatch_decoder, tpatch_decoder_te
    H = 32
    torch.manual_seed(B * 31 + Lp + E)
    dec = torch.nn.Sequential(torch.nn.Linear(D + E, H), torch.nn.ReLU(inplace=True), torch.nn.Linear(H, H),
                              torch.nn.ReLU(inplace=True), torch.nn.Linear(H, 1)).to(dev)
    te_s, te_p = torch.nn.Linear(1, 1).to(dev), torch.nn.Linear(1, E - 1).to(dev)
    tw = (te_s.weight, te_s.bias, te_p.weight, te_p.bias)
    h = torch.randn(B, N, D, device=dev, requires_grad=True)
    t = torch.rand(B, Lp, device=dev) * 3.0
    up = torch.randn(B, Lp, N, device=dev)
    params = list(dec.parameters()) + list(tw)

    def grads(out):
        for q in params + [h]:
            q.grad = None
        (out * up).sum().backward()
        return [h.grad.clone()] + [q.grad.clone() for q in params]

    ref = tpatch_decoder(dec, h, time2vec(t, *tw), precision=precision)
    want = grads(ref)
    out = tpatch_decoder_te(dec, h, t, *tw, precision=precision)
    got = grads(out)
    assert _rel(out, ref) < 1e-6
    for i, (a, b) in enumerate(zip(got, want)):
        # (the decoder's gradients are atomic sums -- their order differs from launch to launch: 1e-5; the time embedding's are scalar
        # sums of cancelling terms in another order: 1e-4)
        assert _rel(a, b) < (1e-5 if i < 7 else 1e-4), (i, _rel(a, b))


def test_fused_decoder_unsupported_shapes_fall_back():
    dev = _dev()
    from immtsf.ops import tpatch_decoder_supported
    mk = lambda *mods: torch.nn.Sequential(*mods).to(dev)      # noqa: E731
    L, R = torch.nn.Linear, torch.nn.ReLU
    assert not tpatch_decoder_supported(mk(L(42, 64), R(), L(64, 64), R(), L(64, 1)), 8, 32, 32, 10)      # H != 32
    assert not tpatch_decoder_supported(mk(L(42, 32), R(), L(32, 1)), 8, 32, 32, 10)                      # other depth
    assert not tpatch_decoder_supported(mk(L(42, 32), R(), L(32, 32), R(), L(32, 1)), 2000, 2000, 32, 10)  # LDS


@pytest.mark.parametrize("shape", [(64, 2, 32, 8), (3, 5, 7, 11), (1, 1, 1, 1)])
def test_patch_flatten3_equals_three_permute_copies(shape):
    """immtsf_patch_flatten3: (B, M, L, N) values / time stamps / mask -> the encoder's (B*N*M, L) patch rows, bit-exact with
    the reference's permute(0, 3, 1, 2).reshape (models/tPatchGNN.py:271-275)"""
    dev = _dev()
    from immtsf.ops import patch_flatten3
    B, M, L, N = shape
    g = torch.Generator().manual_seed(5)
    x, tt = torch.randn(B, M, L, N, generator=g).to(dev), torch.rand(B, M, L, N, generator=g).to(dev)
    mk = (torch.rand(B, M, L, N, generator=g) < 0.5).float().to(dev)
    fx, ft, fm = patch_flatten3(x, tt, mk)
    for got, src in ((fx, x), (ft, tt), (fm, mk)):
        assert torch.equal(got, src.permute(0, 3, 1, 2).reshape(B * N * M, L))


@pytest.mark.parametrize("B,L,C", [(3, 5, 2), (64, 32, 6), (257, 96, 8)])
def test_instance_norm_equals_the_expression(B, L, C):
    """immtsf_instance_norm (one launch) against the normalisation as the reference writes it (models/PatchTST.py:104-109,
    models/TimesNet.py:113-117: x - mean over time, / sqrt(biased variance + 1e-5)); a tensor that wants a gradient keeps the expression"""
    dev = _dev()
    from models._common import plain_instance_norm
    torch.manual_seed(B + L)
    x = torch.randn(B, L, C, device=dev) * 3.0 + 1.5
    xn, mu, sd = plain_instance_norm(x)
    mu_r = x.mean(1, keepdim=True)
    xc = x - mu_r
    sd_r = torch.sqrt(torch.var(xc, dim=1, keepdim=True, unbiased=False) + 1e-5)
    assert mu.shape == mu_r.shape and sd.shape == sd_r.shape
    assert float((mu - mu_r).abs().max()) <= 1e-5 and float((sd - sd_r).abs().max()) <= 1e-5 * float(sd_r.abs().max())
    assert float((xn - xc / sd_r).abs().max()) <= 1e-5
    y = x.clone().requires_grad_(True)
    yn, _, _ = plain_instance_norm(y)
    yn.sum().backward()
    assert y.grad is not None and torch.isfinite(y.grad).all()
