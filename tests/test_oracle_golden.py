"""Pin the CPU oracle (oracle/fusion_ref.py) against golden vectors captured from the real reference.

Tolerance: 2e-5 relative-to-max on outputs, 1e-4 on gradients (fp32 both sides, different op order
inside nn.MultiheadAttention / nn.GRU vs. the oracle's explicit formulation).  Index/bool tensors:
bit-exact."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import fusion_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _close(a, b, tol, what="", floor=1e-3):
    """max|a-b| <= tol * max(max|b|, floor).  The floor matters for gradients that are exactly zero in
    exact arithmetic (e.g. d/dW_q of MMF_XAttn_Add when E_txt is identical over T, so the softmax is
    uniform): both sides then hold ~1e-9 rounding noise."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(np.abs(b).max(), floor)
    err = np.abs(a - b).max() / scale
    assert err <= tol, f"{what}: rel-to-max err {err:.3e} > {tol}"


def _grads(params, out, upstream):
    ps = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    return ps


CASES = ["tiny_h1", "tiny_h2", "noproj_h2", "mid_h4"]


@pytest.mark.parametrize("case", CASES)
def test_ragged_index_bit_exact(case):
    z = _load(f"ttf_t2v_{case}")
    mask, lengths, offsets, rowmap = R.ragged_index(_t(z["notes"]))
    assert np.array_equal(mask.numpy(), z["note_mask"])
    assert np.array_equal(offsets.numpy(), z["offsets"])
    assert np.array_equal(lengths.numpy(), z["lengths"])
    N = z["notes"].shape[1]
    exp = [b * N + n for b in range(len(z["lengths"])) for n in range(N) if z["note_mask"][b, n]]
    assert np.array_equal(rowmap.numpy(), np.asarray(exp, np.int32))


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("block", ["ttf_t2v", "ttf_rec"])
def test_ttf_blocks(case, block):
    z = _load(f"{block}_{case}")
    p = {k: v.requires_grad_(True) for k, v in R.params_from_npz(z).items()}
    notes, tau, t_hat = _t(z["notes"]), _t(z["tau"]), _t(z["t_hat"])
    H = int(z["H"])
    for expand in ([True, False] if block == "ttf_t2v" else [True]):
        if block == "ttf_t2v":
            E, M = R.ttf_t2v_xattn(p, notes, tau, t_hat, H, expand_T=expand)
        else:
            E, M = R.ttf_recavg(p, notes, tau, t_hat)
        _close(E.detach(), z["out_eval.0"], 2e-5, "E_txt eval")
        _close(E.detach(), z["out_train.0"], 2e-5, "E_txt train(p=0)")
        assert np.array_equal(M.numpy(), z["out_eval.1"])
        for v in p.values():
            v.grad = None
        (E * _t(z["upstream"])).sum().backward()
        for k, v in p.items():
            g = v.grad if v.grad is not None else torch.zeros_like(v)
            _close(g, z[f"g.{k}"], 1e-4, f"grad {k} expand={expand}")


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("block", ["mmf_xattn", "mmf_gr"])
def test_mmf_blocks(case, block):
    z = _load(f"{block}_{case}")
    p = {k: v.requires_grad_(True) for k, v in R.params_from_npz(z).items()}
    Y = _t(z["Y_ts"]).requires_grad_(True)
    E = _t(z["E_txt"]).requires_grad_(True)
    M = _t(z["M_txt"])
    if block == "mmf_xattn":
        out = R.mmf_xattn_add(p, Y, E, M, int(z["H"]), float(z["kappa"]))
    else:
        out = R.mmf_gr_add(p, Y, E, M)
    _close(out.detach(), z["out_eval.0"], 2e-5, "Y eval")
    _close(out.detach(), z["out_train.0"], 2e-5, "Y train(p=0)")
    (out * _t(z["upstream"])).sum().backward()
    gtol = 1e-4
    for k, v in p.items():
        _close(v.grad, z[f"g.{k}"], gtol, f"grad {k}")
    _close(Y.grad, z["gin.0"], gtol, "grad Y_ts")
    _close(E.grad, z["gin.1"], gtol, "grad E_txt")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "fusion_*.npz"))),
                         ids=lambda p: os.path.basename(p)[7:-4])
def test_fusion_model(path):
    z = np.load(path)
    name = os.path.basename(path)[len("fusion_"):-4]
    ttf = "TTF_T2V_XAttn" if name.startswith("TTF_T2V_XAttn") else "TTF_RecAvg"
    rest = name[len(ttf) + 1:]
    mmf = "MMF_XAttn_Add" if rest.startswith("MMF_XAttn_Add") else "MMF_GR_Add"
    p = {k: v.requires_grad_(True) for k, v in R.params_from_npz(z).items()}
    Y = _t(z["Y_ts"]).requires_grad_(True)
    out = R.fusion_forward(ttf, mmf, p, _t(z["notes"]), _t(z["tau"]), _t(z["t_hat"]), Y,
                           H=int(z["H"]), kappa=float(z["kappa"]))
    _close(out.detach(), z["out_eval.0"], 2e-5, "Y_out eval")
    if "zeronote" in name:
        # forward-only fixture: the reference's own backward is NaN here (recorded), ours must be finite
        assert bool(z["ref_grad_has_nan"]) or ttf == "TTF_RecAvg"
        out.sum().backward()
        for k, v in p.items():
            if v.grad is not None:
                assert torch.isfinite(v.grad).all(), k
        # quirk: a no-text window returns Y_ts/(1+kappa) under MMF_XAttn_Add, Y_ts under MMF_GR_Add
        b0 = int(np.where(z["lengths"] == 0)[0][0])
        if mmf == "MMF_XAttn_Add":
            _close(out[b0].detach(), z["Y_ts"][b0] / (1 + float(z["kappa"])), 1e-6, "no-text quirk")
        else:
            _close(out[b0].detach(), z["Y_ts"][b0], 1e-6, "no-text passthrough")
        return
    _close(out.detach(), z["out_train.0"], 2e-5, "Y_out train(p=0)")
    (out * _t(z["upstream"])).sum().backward()
    gtol = 1e-4
    for k, v in p.items():
        g = v.grad if v.grad is not None else torch.zeros_like(v)
        _close(g, z[f"g.{k}"], gtol, f"grad {k}")
    _close(Y.grad, z["gin.3"], gtol, "grad Y_ts")


def test_masked_mse():
    z = _load("loss_mse")
    pred = _t(z["pred"]).requires_grad_(True)
    loss = R.masked_mse(_t(z["truth"]), pred, _t(z["mask"]))
    _close(loss.detach(), z["loss"], 1e-6, "loss")
    loss.backward()
    _close(pred.grad, z["dpred"], 1e-6, "dpred")
    es, mc = R.masked_err_sums(_t(z["truth"]), pred.detach(), _t(z["mask"]))
    _close(es, z["err_sum"], 1e-6)
    assert np.array_equal(mc.numpy(), z["mask_count"])
    _close(R.masked_mse(_t(z["truth"]), pred.detach(), _t(z["mask"]), "MAE"), z["mae"], 1e-6)


def test_t_hat_shape_error():
    z = _load("ttf_t2v_tiny_h1")
    p = R.params_from_npz(z)
    with pytest.raises(ValueError):
        R.ttf_t2v_xattn(p, _t(z["notes"]), _t(z["tau"]), _t(z["t_hat"])[:2], int(z["H"]))
    with pytest.raises(ValueError):
        R.ttf_recavg(R.params_from_npz(_load("ttf_rec_tiny_h1")), _t(z["notes"]), _t(z["tau"]), _t(z["t_hat"])[:2])


def test_nan_guard():
    z = _load("ttf_t2v_tiny_h1")
    notes = _t(z["notes"]).clone()
    notes[0, 0, 0] = float("nan")
    with pytest.raises(ValueError):
        R.ttf_t2v_xattn(R.params_from_npz(z), notes, _t(z["tau"]), _t(z["t_hat"]), int(z["H"]))


def test_tpatchgnn_oracle_vs_reference_golden():
    """oracle/tpatchgnn_ref.py (the eager restatement the GPU tests and bench.py's cpu_baseline use) against the fixture
    generated from the real reference's models/tPatchGNN.py: TE + TTCN in isolation and the whole `forecasting`, outputs
    and the gradient of every parameter (tests/golden/make_golden.py:gen_tpatchgnn)."""
    import types

    from oracle.tpatchgnn_ref import TPatchGNNRef
    z = _load("model_tpatchgnn")
    args = types.SimpleNamespace(device="cpu", hid_dim=8, C=3, npatch=2, nlayer=1, te_dim=4, n_heads=1, tf_layer=1, node_dim=4,
                                 hop=1, outlayer="Linear")
    m = TPatchGNNRef(args)
    m.load_state_dict({k[2:]: _t(z[k]) for k in z.files if k.startswith("p.")}, strict=True)
    m.eval()
    X, tt, mask, tp = (_t(z[k]) for k in ("X", "tt", "mask", "tp"))
    B, M, L, N = X.shape
    flat = lambda t: t.permute(0, 3, 1, 2).reshape(B * N * M, L)    # noqa: E731
    h = m.encode_patches(flat(X), flat(tt), flat(mask))[:, :-1]
    _close(h.detach(), z["ttcn_out"], 2e-5, "ttcn_out")
    (h * _t(z["ttcn_upstream"])).sum().backward()
    gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("g_ttcn."))
    n = 0
    for k, p in m.named_parameters():
        if f"g_ttcn.{k}" in z.files:
            _close(p.grad, z[f"g_ttcn.{k}"], 1e-4, "g_ttcn." + k, floor=1e-2 * gmax)
            n += 1
    assert n == 11
    m.zero_grad()
    out = m.forecasting(tp, X, tt, mask)
    _close(out.detach(), z["out"], 2e-5, "forecasting")
    (out * _t(z["upstream"])).sum().backward()
    for k, p in m.named_parameters():
        _close(p.grad, z["g." + k], 1e-4, "g." + k)


# ---- oracle/layers_ref.py against the reference's layer fixtures ----------------------------------------------------
def _layer_params(z):
    return {k[2:]: _t(z[k]).clone().requires_grad_(True) for k in z.files if k.startswith("p.")}


def test_layers_oracle_toy_goldens():
    from oracle import layers_ref as L
    z = _load("layer_full_attention")
    q, k, v = (_t(z[n]).clone().requires_grad_(True) for n in ("q", "k", "v"))
    o = L.full_attention(q, k, v)
    _close(o.detach(), z["out"], 2e-5, "full_attention")
    (o * _t(z["upstream"])).sum().backward()
    for t, n in ((q, "gq"), (k, "gk"), (v, "gv")):
        _close(t.grad, z[n], 1e-4, n)
    z = _load("layer_attention_layer")
    p = _layer_params(z)
    x = _t(z["x"]).clone().requires_grad_(True)
    o = L.attention_layer(p, "", x, x, x, int(z["H"]))
    _close(o.detach(), z["out"], 2e-5, "attention_layer")
    (o * _t(z["upstream"])).sum().backward()
    _close(x.grad, z["gx"], 1e-4, "attention_layer gx")
    for kname, t in p.items():
        _close(t.grad, z["g." + kname], 1e-4, kname)
    z = _load("layer_encoder")
    p = _layer_params(z)
    x = _t(z["x"]).clone().requires_grad_(True)
    o = L.encoder(p, x, int(z["H"]), 2)
    _close(o.detach(), z["out"], 2e-5, "encoder")
    (o * _t(z["upstream"])).sum().backward()
    _close(x.grad, z["gx"], 1e-4, "encoder gx")
    for kname, t in p.items():
        _close(t.grad, z["g." + kname], 1e-4, kname)
    z = _load("layer_patch_embedding")
    w = _t(z["w"]).clone().requires_grad_(True)
    x = _t(z["x"]).clone().requires_grad_(True)
    o = L.patch_embedding(w, x, int(z["patch_len"]), int(z["stride"]), int(z["stride"]))
    _close(o.detach(), z["out"], 2e-5, "patch_embedding")
    (o * _t(z["upstream"])).sum().backward()
    _close(x.grad, z["gx"], 1e-4, "patch gx")
    _close(w.grad, z["gw"], 1e-4, "patch gw")
    z = _load("layer_data_embedding")
    w = _t(z["w"]).clone().requires_grad_(True)
    x = _t(z["x"]).clone().requires_grad_(True)
    o = L.data_embedding(w, x)
    _close(o.detach(), z["out"], 2e-5, "data_embedding")
    (o * _t(z["upstream"])).sum().backward()
    _close(x.grad, z["gx"], 1e-4, "data gx")
    _close(w.grad, z["gw"], 1e-4, "data gw")


def _big_case(tag):
    """inputs / weights of a PatchTST-size layer case, regenerated from tests/golden/seeded.py exactly as make_golden.py did"""
    import sys
    sys.path.insert(0, GOLDEN)
    import seeded
    D, H, DFF = 512, 2, 2048
    att = {f"{n}_projection.{w}": ((D, D) if w == "weight" else (D,)) for n in ("query", "key", "value", "out") for w in ("weight", "bias")}
    if tag == "attention_layer":
        return seeded, seeded.rand((48, 10, D), 501), seeded.state_like(att, 510), 510
    if tag == "encoder_layer":
        shapes = {"attention." + k: v for k, v in att.items()}
        shapes.update({"conv1.weight": (DFF, D, 1), "conv1.bias": (DFF,), "conv2.weight": (D, DFF, 1), "conv2.bias": (D,),
                       "norm1.weight": (D,), "norm1.bias": (D,), "norm2.weight": (D,), "norm2.bias": (D,)})
        return seeded, seeded.rand((48, 10, D), 501), seeded.state_like(shapes, 520), 520
    if tag == "patch_embedding":
        return seeded, seeded.rand((8, 6, 96), 530), seeded.state_like({"value_embedding.weight": (D, 18)}, 531), 531
    if tag == "reprogramming":
        # d_keys = d_model // n_heads = 2: the projections are 16 wide (models/TimeLLM.py:36-41)
        shapes = {"query_projection.weight": (16, 16), "query_projection.bias": (16,), "key_projection.weight": (16, 768),
                  "key_projection.bias": (16,), "value_projection.weight": (16, 768), "value_projection.bias": (16,),
                  "out_projection.weight": (768, 16), "out_projection.bias": (768,)}
        return seeded, seeded.rand((48, 5, 16), 540), seeded.state_like(shapes, 542), 542
    raise KeyError(tag)


def _check_big(tag, out, gx, grads, z, seeded, seed, tol_o, tol_g):
    _close(out[:8], z["out8"], tol_o, tag + " out8")
    assert abs(float(np.linalg.norm(np.asarray(out, np.float64))) / float(z["out_norm"]) - 1.0) < tol_o, tag + " out norm"
    _close(gx[:8], z["gx8"], tol_g, tag + " gx8", floor=1e-3 * float(np.abs(z["gx8"]).max()))
    assert abs(float(np.linalg.norm(np.asarray(gx, np.float64))) / float(z["gx_norm"]) - 1.0) < tol_g, tag + " gx norm"
    # gradients that are zero in exact arithmetic (the key bias: a softmax shift) are rounding noise on both sides: judge
    # every fingerprint against the largest gradient norm of the module, not its own
    gmax = max(float(z["probe." + k][-1]) for k in grads)
    for i, k in enumerate(sorted(grads)):
        want = z["probe." + k]
        got = seeded.probes(grads[k], seed + 2000 + i)
        assert np.abs(got - want).max() <= tol_g * max(want[-1], 1e-2 * gmax) * 4, (tag, k, got, want)     # |projection| ~ norm


@pytest.mark.parametrize("tag", ["attention_layer", "encoder_layer", "patch_embedding", "reprogramming"])
def test_layers_oracle_patchtst_size(tag):
    from oracle import layers_ref as L
    seeded, x, sd, seed = _big_case(tag)
    z = _load("layer_big_" + tag)
    p = {k: _t(v).clone().requires_grad_(True) for k, v in sd.items()}
    xx = _t(x).clone().requires_grad_(True)
    if tag == "attention_layer":
        o = L.attention_layer(p, "", xx, xx, xx, 2)
    elif tag == "encoder_layer":
        o = L.encoder_layer(p, "", xx, 2, "gelu")
    elif tag == "patch_embedding":
        o = L.patch_embedding(p["value_embedding.weight"], xx, 18, 9, 9)
    else:
        src = _t(seeded.rand((1000, 768), 541))
        o = L.reprogramming_layer(p, xx, src, src, 8)
    up = _t(seeded.rand(tuple(o.shape), seed + 1000))
    (o * up).sum().backward()
    _check_big(tag, o.detach().numpy(), xx.grad.numpy(), {k: v.grad.numpy() for k, v in p.items()}, z, seeded, seed, 5e-5, 2e-4)
