"""GPU parity tests (run on the MI355X box with `-m gpu`): the HIP fusion path, called through the drop-in modules
(i.e. through the C ABI), against (a) the committed golden vectors of the real reference and (b) the CPU oracle
on seeded synthetic batches up to the benchmark shape.

Tolerances (relative to the tensor's max magnitude, floor 1e-3 for gradients that are ~0):
  fp32 mode  outputs 1e-4 (north_star), gradients 2e-4 (reduction order differs: MFMA k-chains vs. MKL)
  bf16 mode  relative L2 error 3e-2 outputs / 4e-2 gradients (operands rounded to 8 significant bits, fp32
             accumulation); 1.5e-1 for the two scalar gradients that are sums of cancelling terms
Index/bool tensors: bit-exact.
"""
import glob
import os
import sys
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


def _relerr(a, b, floor=1e-3):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / max(float(b.abs().max()), floor))


def _l2err(a, b, floor=1e-3):
    """relative L2 error (bf16 mode: a max-norm over a cancelling sum, e.g. a scalar gradient, is not meaningful)"""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).norm() / max(float(b.norm()), floor * max(1.0, b.numel() ** 0.5)))


def _check(errs, tol):
    bad = {k: v for k, v in errs.items() if not (v <= tol)}
    assert not bad, f"tolerance {tol}: " + ", ".join(f"{k}={v:.2e}" for k, v in sorted(bad.items(), key=lambda kv: -kv[1])[:12])


def _args(ttf, mmf, llm, d_txt, H, C, dropout=0.0, kappa=0.5, sigma=1.3):
    return types.SimpleNamespace(TTF_module=ttf, MMF_module=mmf, llm_model_fusion=llm, llm_layers_fusion=6,
                                 max_length=1024, device="cuda", use_text_embeddings=True, recency_sigma=sigma,
                                 n_heads_fusion=H, dropout=dropout, d_txt=d_txt, C=C, kappa=kappa)


def _setup_toys():
    from fusions.load_llm import register_d_model
    register_d_model("TOY16", 16)
    register_d_model("TOY48", 48)


def _split_name(path):
    name = os.path.basename(path)[len("fusion_"):-4]
    ttf = "TTF_T2V_XAttn" if name.startswith("TTF_T2V_XAttn") else "TTF_RecAvg"
    rest = name[len(ttf) + 1:]
    mmf = "MMF_XAttn_Add" if rest.startswith("MMF_XAttn_Add") else "MMF_GR_Add"
    case = rest[len(mmf) + 1:]
    return ttf, mmf, case


@pytest.fixture(params=["fold", "chain"])
def t2v_form(request):
    """run the test with TTF_T2V_XAttn forced into its folded form (wherever that form's limits hold; csrc/t2v_fold.hip) and as the
    reference's GEMM chain: the library's own choice ("auto") is one of the two, by batch size"""
    from immtsf import config
    old, config.t2v_form = config.t2v_form, request.param
    yield request.param
    config.t2v_form = old


# ------------------------------------------------------------------------------------------ golden vectors
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "fusion_*.npz"))),
                         ids=lambda p: os.path.basename(p)[7:-4])
def test_fusion_model_vs_reference_golden(path, t2v_form):
    dev = _dev()
    _setup_toys()
    from fusions.FusionModel import FusionModel
    from immtsf import config
    config.precision = "fp32"
    z = np.load(path)
    ttf, mmf, case = _split_name(path)
    llm = "TOY48" if int(z["d_m"]) == 48 else "TOY16"
    d_txt = None if int(z["d_txt"]) < 0 else int(z["d_txt"])
    C = z["Y_ts"].shape[2]
    m = FusionModel(_args(ttf, mmf, llm, d_txt, int(z["H"]), C, kappa=float(z["kappa"]))).to(dev)
    sd = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p.")}
    m.load_state_dict(sd, strict=True)
    notes, tau, t_hat = [torch.from_numpy(z[k]).to(dev) for k in ("notes", "tau", "t_hat")]
    Y = torch.from_numpy(z["Y_ts"]).to(dev).requires_grad_(True)
    m.eval()
    with torch.no_grad():
        out = m(notes, tau, t_hat, Y)
    errs = {"out_eval": _relerr(out, torch.from_numpy(z["out_eval.0"]))}
    if case == "zeronote":
        m.train()
        out = m(notes, tau, t_hat, Y)
        out.sum().backward()     # the reference yields NaN grads here; ours must be finite
        for k, p in m.named_parameters():
            assert torch.isfinite(p.grad).all(), k
        _check(errs, 1e-4)
        return
    m.train()
    out = m(notes, tau, t_hat, Y)
    errs["out_train"] = _relerr(out, torch.from_numpy(z["out_train.0"]))
    (out * torch.from_numpy(z["upstream"]).to(dev)).sum().backward()
    gerrs = {"gY": _relerr(Y.grad, torch.from_numpy(z["gin.3"]))}
    for k, p in m.named_parameters():
        gerrs["g." + k] = _relerr(p.grad, torch.from_numpy(z["g." + k]))
    _check(errs, 1e-4)
    _check(gerrs, 2e-4)


@pytest.mark.parametrize("case", ["tiny_h1", "tiny_h2", "noproj_h2", "mid_h4"])
def test_ragged_index_bit_exact(case):
    dev = _dev()
    from immtsf import _lib
    lib = _lib.load()
    z = np.load(os.path.join(GOLDEN, f"ttf_t2v_{case}.npz"))
    notes = torch.from_numpy(z["notes"]).to(dev)
    B, N, d_m = notes.shape
    mask = torch.zeros(B * N, dtype=torch.uint8, device=dev)
    lengths = torch.zeros(B, dtype=torch.int32, device=dev)
    offsets = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    rowmap = torch.full((B * N,), -1, dtype=torch.int32, device=dev)
    seg = torch.full((B * N,), -1, dtype=torch.int32, device=dev)
    mtxt = torch.zeros(B, dtype=torch.uint8, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.immtsf_ragged_index(_lib.ptr(notes), B, N, d_m, _lib.ptr(mask), _lib.ptr(lengths), _lib.ptr(offsets),
                                       _lib.ptr(rowmap), _lib.ptr(seg), _lib.ptr(mtxt), _lib.ptr(flag), _lib.stream_ptr()),
               "ragged_index")
    torch.cuda.synchronize()
    assert np.array_equal(mask.cpu().numpy().reshape(B, N).astype(bool), z["note_mask"])
    assert np.array_equal(offsets.cpu().numpy(), z["offsets"])
    assert np.array_equal(lengths.cpu().numpy(), z["lengths"])
    tot = int(z["offsets"][-1])
    exp = [b * N + n for b in range(B) for n in range(N) if z["note_mask"][b, n]]
    assert np.array_equal(rowmap.cpu().numpy()[:tot], np.asarray(exp, np.int32))
    assert np.array_equal(seg.cpu().numpy()[:tot], np.asarray([e // N for e in exp], np.int32))
    assert int(flag.item()) == 0


# ------------------------------------------------------------------------------------------ oracle, synthetic
def _synthetic(seed, B, N, T, C, d_m, dev, min_notes=1, scatter_masks=False, lengths=None):
    g = torch.Generator().manual_seed(seed)
    notes = torch.randn(B, N, d_m, generator=g)
    drawn = torch.randint(min_notes, N + 1, (B,), generator=g)
    lengths = drawn if lengths is None else torch.tensor(lengths)
    tau = torch.zeros(B, N)
    for b in range(B):
        L = int(lengths[b])
        notes[b, L:] = 0
        tau[b, :L] = torch.sort(torch.rand(L, generator=g) * 24.0).values
    if scatter_masks:     # masked notes need not be a suffix: knock out one interior note of window 0
        notes[0, 0] = 0
    t_hat = torch.sort(torch.rand(B, T, generator=g), dim=1).values
    Y = torch.randn(B, T, C, generator=g)
    up = torch.randn(B, T, C, generator=g)
    return notes, tau, t_hat, Y, up


def _run_pair(ttf, mmf, B, N, T, C, d_m, d_txt, H, precision, seed=0, llm="GPT2", min_notes=1, scatter=False,
              err=None, lengths=None):
    err = err or _relerr
    dev = _dev()
    from fusions.FusionModel import FusionModel
    from fusions.load_llm import register_d_model
    from immtsf import config
    from oracle import fusion_ref as R
    register_d_model("SYN", d_m)
    config.precision = precision
    torch.manual_seed(seed)
    m = FusionModel(_args(ttf, mmf, "SYN", d_txt, H, C)).to(dev)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "layer_norm" in k:
                p.uniform_(0.5, 1.5) if k.endswith("weight") else p.uniform_(-0.3, 0.3)
    notes, tau, t_hat, Y, up = _synthetic(seed + 1, B, N, T, C, d_m, dev, min_notes, scatter, lengths)
    m.train()
    Yg = Y.to(dev).requires_grad_(True)
    out = m(notes.to(dev), tau.to(dev), t_hat.to(dev), Yg)
    (out * up.to(dev)).sum().backward()
    # oracle on the CPU with the same weights
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    Yc = Y.clone().requires_grad_(True)
    ref = R.fusion_forward(ttf, mmf, p, notes, tau, t_hat, Yc, H=H, kappa=0.5, expand_T=False)
    (ref * up).sum().backward()
    errs = {"out": err(out, ref)}
    gerrs = {"gY": err(Yg.grad, Yc.grad)}
    for k, prm in m.named_parameters():
        g = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
        gerrs["g." + k] = err(prm.grad, g)
    config.precision = "fp32"
    return errs, gerrs


PAIRS = [("TTF_T2V_XAttn", "MMF_XAttn_Add"), ("TTF_T2V_XAttn", "MMF_GR_Add"), ("TTF_RecAvg", "MMF_XAttn_Add"),
         ("TTF_RecAvg", "MMF_GR_Add")]


@pytest.mark.parametrize("ttf,mmf", PAIRS)
def test_pairs_fp32_small_odd_shapes(ttf, mmf, t2v_form):
    # dims that are not multiples of any tile: exercises every edge path of the GEMM and the row kernels
    errs, gerrs = _run_pair(ttf, mmf, B=5, N=7, T=9, C=5, d_m=52, d_txt=36, H=3, precision="fp32", scatter=True)
    _check(errs, 1e-4)
    _check(gerrs, 2e-4)


@pytest.mark.parametrize("C,d_txt", [(20, 12), (3, 10)])
def test_xattn_add_shapes_outside_the_fused_head_and_vector_rows(C, d_txt):
    """C > 16 keeps MMF_XAttn_Add's output head on the GEMM + ln_blend kernels; d_txt % 4 != 0 keeps LayerNorm and the
    head on their scalar-row kernels: both fallbacks still match the oracle (fp32)."""
    errs, gerrs = _run_pair("TTF_T2V_XAttn", "MMF_XAttn_Add", B=4, N=5, T=6, C=C, d_m=16, d_txt=d_txt, H=2, precision="fp32")
    _check(errs, 1e-4)
    _check(gerrs, 2e-4)


@pytest.mark.parametrize("ttf,mmf", PAIRS)
@pytest.mark.parametrize("d_m,d_txt", [(40, 24), (44, 48)])
def test_pairs_bf16_shapes_outside_the_bf16_dataflow(ttf, mmf, d_m, d_txt):
    """bf16 mode at widths the bf16-in-memory GEMM path does not take (d_txt % 16 != 0, or d_m % 8 != 0): the blocks fall back
    to fp32 activations in memory + the round-1 kernel (operands rounded while staged) and must still be within the bf16 bars."""
    errs, gerrs = _run_pair(ttf, mmf, B=9, N=11, T=7, C=5, d_m=d_m, d_txt=d_txt, H=2, precision="bf16", err=_l2err)
    _check(errs, 3e-2)
    small = {k: v for k, v in gerrs.items() if "time2vec.linear" in k or "log_recency_sigma" in k}
    _check({k: v for k, v in gerrs.items() if k not in small}, 4e-2)
    _check(small, 2.5e-1)


@pytest.mark.parametrize("ttf,mmf", PAIRS)
def test_pairs_fp32_benchmark_shape(ttf, mmf, t2v_form):
    # BASELINE config 2 shape: B=64, N<=32, T=32, C=8, d_m=d=768, H=1
    errs, gerrs = _run_pair(ttf, mmf, B=64, N=32, T=32, C=8, d_m=768, d_txt=768, H=1, precision="fp32")
    _check(errs, 1e-4)
    _check(gerrs, 2e-4)


@pytest.mark.parametrize("ttf,mmf", PAIRS)
def test_pairs_bf16_benchmark_shape(ttf, mmf, t2v_form):
    errs, gerrs = _run_pair(ttf, mmf, B=64, N=32, T=32, C=8, d_m=768, d_txt=768, H=1, precision="bf16", err=_l2err)
    _check(errs, 3e-2)
    # scalar / near-cancelling gradients (time2vec.linear, log_recency_sigma) get a wider band in bf16: they are sums of
    # ~36 k terms of both signs, so which operands happen to be rounded decides the third digit (0.05 .. 0.18 observed
    # across kernel versions that all pass the fp32 test above at 2e-4)
    small = {k: v for k, v in gerrs.items() if "time2vec.linear" in k or "log_recency_sigma" in k}
    _check({k: v for k, v in gerrs.items() if k not in small}, 4e-2)
    _check(small, 2.5e-1)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_many_windows_paths_vs_oracle(precision, t2v_form):
    """320 windows (B*T = 10 240 rows, ~5 k packed note rows, d_m 256 -> d 768): the many-row paths against the oracle -- the
    three-launch index builder (B > 256), the 16-byte column sums (M >= 4096) and, in bf16 mode, the persistent GEMM of
    csrc/gemm3.hip for the projections over B*T rows (240 row tiles of 128 x 256) and its split-K form with the in-kernel bias
    gradient for their weight gradients (K = 10 240 >= 8192)."""
    bf = precision == "bf16"
    errs, gerrs = _run_pair("TTF_T2V_XAttn", "MMF_XAttn_Add", B=320, N=32, T=32, C=8, d_m=256, d_txt=768, H=2, precision=precision,
                            err=_l2err if bf else None)
    _check(errs, 3e-2 if bf else 1e-4)
    small = {k: v for k, v in gerrs.items() if "time2vec.linear" in k} if bf else {}
    _check({k: v for k, v in gerrs.items() if k not in small}, 4e-2 if bf else 3e-4)
    _check(small, 2.5e-1)


def test_llama_dims_multihead_fp32(t2v_form):
    # config 3/5 flavour: d_m=4096 -> d=768, H=4, fewer windows so the CPU oracle stays quick
    errs, gerrs = _run_pair("TTF_T2V_XAttn", "MMF_XAttn_Add", B=6, N=40, T=16, C=6, d_m=4096, d_txt=768, H=4,
                            precision="fp32")
    _check(errs, 1e-4)
    _check(gerrs, 2e-4)


# ------------------------------------------------------------------------------------------ dropout
def _dropout_parity(ttf, mmf, precision="fp32", B=6, N=9, T=7, C=4, d_m=40, d=32, H=2, pd=0.25, err=None, tol=(1e-4, 2e-4), exact_eval=True):
    """train mode, p=0.25: export the Philox keep-masks the kernels used and feed them to the oracle."""
    dev = _dev()
    from fusions.FusionModel import FusionModel
    from fusions.load_llm import register_d_model
    from immtsf import config, ops
    from oracle import fusion_ref as R
    _relerr = err or globals()["_relerr"]
    register_d_model("SYN", d_m)
    config.precision = precision
    config.manual_seed(1234)
    torch.manual_seed(3)
    a = _args(ttf, mmf, "SYN", d, H, C, dropout=pd)
    m = FusionModel(a).to(dev)
    notes, tau, t_hat, Y, up = _synthetic(5, B, N, T, C, d_m, dev)
    m.train()
    Yg = Y.to(dev).requires_grad_(True)
    out = m(notes.to(dev), tau.to(dev), t_hat.to(dev), Yg)
    (out * up.to(dev)).sum().backward()

    def keep(seed, site, shape):
        n = int(np.prod(shape))
        return ops.dropout_keep_mask(seed, site, n, pd, dev).cpu().view(*shape).float()

    drop = {"ttf": {}, "mmf": {}}
    if ttf == "TTF_T2V_XAttn":
        drop["ttf"]["attn"] = keep(m.ttf.last_seed, 1, (B, T, H, N))
        drop["ttf"]["out"] = keep(m.ttf.last_seed, 2, (B, T, d))
    else:
        drop["ttf"]["out"] = keep(m.ttf.last_seed, 3, (B, T, d))
    if mmf == "MMF_XAttn_Add":
        drop["mmf"]["attn"] = keep(m.mmf.last_seed, 4, (B, H, T, T))
        drop["mmf"]["out"] = keep(m.mmf.last_seed, 5, (B, T, C))
    else:
        drop["mmf"]["out"] = keep(m.mmf.last_seed, 6, (B, T, C))
    for grp in drop.values():           # the masks must look like Bernoulli(1-p): 4.5 sigma band per mask
        for k, v in grp.items():
            sigma = (pd * (1 - pd) / v.numel()) ** 0.5
            assert abs(float(v.mean()) - (1 - pd)) < 4.5 * sigma + 1e-3, (k, float(v.mean()), v.numel())
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    Yc = Y.clone().requires_grad_(True)
    ref = R.fusion_forward(ttf, mmf, p, notes, tau, t_hat, Yc, H=H, kappa=0.5, drop=drop, p_drop=pd, expand_T=True)
    (ref * up).sum().backward()
    errs = {"out": _relerr(out, ref)}
    gerrs = {"gY": _relerr(Yg.grad, Yc.grad)}
    for k, prm in m.named_parameters():
        g = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
        gerrs["g." + k] = _relerr(prm.grad, g)
    _check(errs, tol[0])
    small = {k: v for k, v in gerrs.items() if precision == "bf16" and ("time2vec.linear" in k or "log_recency_sigma" in k)}
    _check({k: v for k, v in gerrs.items() if k not in small}, tol[1])
    _check(small, 2.5e-1)
    # eval mode ignores dropout and is deterministic
    m.eval()
    with torch.no_grad():
        o1 = m(notes.to(dev), tau.to(dev), t_hat.to(dev), Y.to(dev))
        o2 = m(notes.to(dev), tau.to(dev), t_hat.to(dev), Y.to(dev))
    if exact_eval:
        assert torch.equal(o1, o2)
    else:       # a GEMM that splits its reduction over workgroups adds the partial sums in arrival order
        _check({"eval-repeat": _relerr(o1, o2)}, 1e-5)
    ref_eval = R.fusion_forward(ttf, mmf, {k: v.detach() for k, v in p.items()}, notes, tau, t_hat, Y, H=H, kappa=0.5)
    _check({"eval": _relerr(o1, ref_eval)}, tol[0])


@pytest.mark.parametrize("ttf,mmf", PAIRS)
def test_dropout_with_exported_masks(ttf, mmf, t2v_form):
    _dropout_parity(ttf, mmf)


@pytest.mark.parametrize("T,H,d", [(17, 2, 64), (7, 2, 96), (32, 1, 1024)])
def test_xattn_add_tile_attention_bf16_with_dropout(T, H, d):
    """bf16 mode runs MMF_XAttn_Add's T x T attention as one MFMA tile kernel per direction (attn.hip: xattn_tile_*): head widths
    on the LDS-staged score path (32), on the direct one with a half k-step tail (48) and over four column chunks (1024); T
    odd / below / at the 32-row tile; attention dropout through the same exported Philox masks as the GEMM path."""
    _dropout_parity("TTF_T2V_XAttn", "MMF_XAttn_Add", precision="bf16", T=T, H=H, d=d, d_m=64, err=_l2err, tol=(3e-2, 4e-2),
                    exact_eval=d < 1024)


def test_dropout_mask_statistics():
    """a million-element keep mask per site: mean within 5 sigma of 1-p, sites/seeds decorrelated, p=0 keeps all"""
    dev = _dev()
    from immtsf import ops
    n, pd = 1 << 20, 0.1
    sig = (pd * (1 - pd) / n) ** 0.5
    m1 = ops.dropout_keep_mask(42, 1, n, pd, dev).float()
    m2 = ops.dropout_keep_mask(42, 2, n, pd, dev).float()
    m3 = ops.dropout_keep_mask(43, 1, n, pd, dev).float()
    for m in (m1, m2, m3):
        assert abs(float(m.mean()) - (1 - pd)) < 5 * sig
    for a, b in ((m1, m2), (m1, m3)):
        corr = float(((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std()))
        assert abs(corr) < 5e-3
    assert torch.equal(m1, ops.dropout_keep_mask(42, 1, n, pd, dev).float())
    assert float(ops.dropout_keep_mask(42, 1, 4096, 0.0, dev).float().mean()) == 1.0
    # neighbouring elements are not correlated either (lag-1 autocorrelation)
    c = float(((m1[1:] - m1.mean()) * (m1[:-1] - m1.mean())).mean() / m1.var())
    assert abs(c) < 5e-3


# ------------------------------------------------------------------------------------------ error behaviour, properties
def test_nan_and_shape_errors():
    dev = _dev()
    _setup_toys()
    from fusions.FusionModel import FusionModel
    from immtsf import config
    config.nan_check = "sync"
    m = FusionModel(_args("TTF_T2V_XAttn", "MMF_XAttn_Add", "TOY16", 8, 2, 3)).to(dev)
    notes, tau, t_hat, Y, _ = _synthetic(0, 4, 5, 6, 3, 16, dev)
    bad = notes.clone()
    bad[1, 0, 3] = float("nan")
    with pytest.raises(ValueError, match="NaN"):
        m(bad.to(dev), tau.to(dev), t_hat.to(dev), Y.to(dev))
    Yb = Y.clone()
    Yb[0, 0, 0] = float("nan")
    with pytest.raises(ValueError, match="Y_ts contains NaN"):
        m(notes.to(dev), tau.to(dev), t_hat.to(dev), Yb.to(dev))
    with pytest.raises(ValueError, match="Expected t_hat shape"):
        m(notes.to(dev), tau.to(dev), t_hat[:2].to(dev), Y.to(dev))
    out1 = m(notes.to(dev), tau.to(dev), t_hat.to(dev), Y.to(dev))       # still usable afterwards
    out2 = m(notes.to(dev), tau.to(dev), t_hat[0].to(dev), Y.to(dev))    # 1-D t_hat broadcasts
    assert out1.shape == out2.shape == (4, 6, 3)
    config.nan_check = "deferred"
    m(bad.to(dev), tau.to(dev), t_hat.to(dev), Y.to(dev))                 # no raise in forward ...
    with pytest.raises(ValueError, match="NaN"):
        m.check_nan()                                                      # ... but the flag was set
    config.nan_check = "sync"
    with pytest.raises(Exception):
        m(notes, tau, t_hat, Y)      # CPU tensors: no silent fallback


def test_padding_invariance_and_window_independence(t2v_form):
    """size-independent properties at the full benchmark shape: (1) extra zero padding of the note axis changes
    nothing (the ragged pack ignores it); (2) a window's output does not depend on the other windows."""
    dev = _dev()
    from fusions.FusionModel import FusionModel
    from fusions.load_llm import register_d_model
    from immtsf import config
    register_d_model("SYN", 768)
    config.precision = "fp32"
    torch.manual_seed(0)
    m = FusionModel(_args("TTF_T2V_XAttn", "MMF_XAttn_Add", "SYN", 768, 1, 8)).to(dev).eval()
    notes, tau, t_hat, Y, _ = _synthetic(1, 64, 32, 32, 8, 768, dev)
    notes, tau, t_hat, Y = notes.to(dev), tau.to(dev), t_hat.to(dev), Y.to(dev)
    with torch.no_grad():
        base = m(notes, tau, t_hat, Y)
        padded = m(torch.cat([notes, torch.zeros(64, 9, 768, device=dev)], 1),
                   torch.cat([tau, torch.zeros(64, 9, device=dev)], 1), t_hat, Y)
        half = m(notes[:32], tau[:32], t_hat[:32], Y[:32])
    # not bit-identical: the padded upper bound of the packed row count feeds the GEMM tile/split-K heuristics
    assert _relerr(padded, base) < 1e-5
    assert _relerr(half, base[:32]) < 1e-5


def test_masked_mse_matches_oracle():
    dev = _dev()
    from immtsf.ops import masked_mse
    from oracle import fusion_ref as R
    z = np.load(os.path.join(GOLDEN, "loss_mse.npz"))
    pred = torch.from_numpy(z["pred"]).to(dev).requires_grad_(True)
    loss = masked_mse(pred, torch.from_numpy(z["truth"]).to(dev), torch.from_numpy(z["mask"]).to(dev))
    (loss * 3.0).backward()
    assert _relerr(loss, torch.from_numpy(z["loss"])) < 1e-6
    assert _relerr(pred.grad, 3.0 * torch.from_numpy(z["dpred"])) < 1e-5
    g = torch.Generator().manual_seed(0)
    t, p = torch.randn(64, 32, 8, generator=g), torch.randn(64, 32, 8, generator=g)
    mk = (torch.rand(64, 32, 8, generator=g) < 0.7).float()
    pc = p.clone().requires_grad_(True)
    R.masked_mse(t, pc, mk).backward()
    pg = p.to(dev).requires_grad_(True)
    l2 = masked_mse(pg, t.to(dev), mk.to(dev))
    l2.backward()
    assert _relerr(l2, R.masked_mse(t, p, mk)) < 1e-5
    assert _relerr(pg.grad, pc.grad) < 1e-5


def test_masked_mse_single_kernel_vs_staged_and_unit_seed():
    """The single-workgroup kernel (small inputs), the three-stage path (large inputs, or a process group), the multi-workgroup
    kernel for counts known beforehand (C <= 64; called twice: its ticket word must come back to zero) and the oracle agree,
    with local and with global counts; backward_unit() seeds with the cached 1 and skips the multiply."""
    dev = _dev()
    from immtsf import ops
    from oracle import fusion_ref as R
    g = torch.Generator().manual_seed(3)
    for shape in [(64, 32, 8), (7, 5, 3), (300, 64, 9), (40, 3, 70)]:      # the third is above IMMTSF_MSE_SMALL_MAX; the last has C > 64
        t, p = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g)
        mk = (torch.rand(*shape, generator=g) < 0.6).float()
        mk[..., 0] = 0.0                                            # a variable that is never observed
        pc = p.clone().requires_grad_(True)
        ref = R.masked_mse(t, pc, mk)
        ref.backward()
        pg = p.to(dev).requires_grad_(True)
        loss = ops.masked_mse(pg, t.to(dev), mk.to(dev))
        ops.backward_unit(loss)
        assert _relerr(loss, ref) < 1e-5
        assert _relerr(pg.grad, pc.grad) < 1e-5
        # global counts = 2x the local ones: loss and gradient halve
        cnt = 2.0 * mk.reshape(-1, shape[-1]).sum(0).to(dev)
        pg2 = p.to(dev).requires_grad_(True)
        l2 = ops.masked_mse(pg2, t.to(dev), mk.to(dev), None, cnt)
        l2.backward()
        assert _relerr(l2, 0.5 * ref) < 1e-5
        assert _relerr(pg2.grad, 0.5 * pc.grad) < 1e-5
        l3 = ops.masked_mse(p.to(dev), t.to(dev), mk.to(dev), None, cnt)
        assert torch.equal(l3, l2.detach())
    assert ops.is_unit_grad(ops.unit_grad(dev)) and not ops.is_unit_grad(torch.ones((), device=dev))


def test_long_ragged_llama_dims_fp32():
    """BASELINE configs[4] flavour (MIMIC-shaped): long ragged note sequences (N up to 1500 here, kernels are sized for
    4096), LLaMA-width embeddings d_m=4096 -> d_txt=768, T=32, C=8; few windows so the CPU oracle stays quick."""
    errs, gerrs = _run_pair("TTF_T2V_XAttn", "MMF_XAttn_Add", B=3, N=1500, T=32, C=8, d_m=4096, d_txt=768, H=1,
                            precision="fp32", min_notes=700)
    _check(errs, 1e-4)
    _check(gerrs, 3e-4)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_cfg5_longest_windows_vs_oracle(precision):
    """BASELINE configs[4]'s extreme: two windows of N_b = 4096 and 2977 notes (the chunked ragged-attention path: 16 and
    12 chunks of 256 notes, partial sums, cross-chunk softmax / dqs reductions), d_m = 4096 -> d_txt = 768, T = 32, C = 8,
    against the oracle -- north_star tolerances: fp32 1e-4 (gradients 3e-4: 4096-term sums), bf16 3e-2 / 4e-2."""
    errs, gerrs = _run_pair("TTF_T2V_XAttn", "MMF_XAttn_Add", B=2, N=4096, T=32, C=8, d_m=4096, d_txt=768, H=1,
                            precision=precision, lengths=[4096, 2977], err=_relerr if precision == "fp32" else _l2err)
    if precision == "fp32":
        _check(errs, 1e-4)
        _check(gerrs, 3e-4)
    else:
        _check(errs, 3e-2)
        small = {k: v for k, v in gerrs.items() if "time2vec.linear" in k}
        _check({k: v for k, v in gerrs.items() if k not in small}, 4e-2)
        _check(small, 2.5e-1)


def test_chunked_attention_dropout_and_empty_windows():
    """the chunked path (N > 256) with attention-weight dropout on, multi-head, a one-note window, a one-chunk window
    and a window that ends exactly on a chunk boundary: the exported Philox keep-masks fed to the oracle, fp32 1e-4."""
    dev = _dev()
    from fusions.FusionModel import FusionModel
    from fusions.load_llm import register_d_model
    from immtsf import config, ops
    from oracle import fusion_ref as R
    B, N, T, C, d_m, d, H, pd = 5, 700, 9, 4, 40, 32, 2, 0.25
    register_d_model("SYN", d_m)
    config.precision = "fp32"
    config.manual_seed(77)
    torch.manual_seed(5)
    m = FusionModel(_args("TTF_T2V_XAttn", "MMF_XAttn_Add", "SYN", d, H, C, dropout=pd)).to(dev).train()
    notes, tau, t_hat, Y, up = _synthetic(9, B, N, T, C, d_m, dev, lengths=[700, 1, 512, 37, 257])
    Yg = Y.to(dev).requires_grad_(True)
    out = m(notes.to(dev), tau.to(dev), t_hat.to(dev), Yg)
    (out * up.to(dev)).sum().backward()
    keep = lambda seed, site, shape: ops.dropout_keep_mask(seed, site, int(np.prod(shape)), pd, dev).cpu().view(*shape).float()  # noqa: E731
    drop = {"ttf": {"attn": keep(m.ttf.last_seed, 1, (B, T, H, N)), "out": keep(m.ttf.last_seed, 2, (B, T, d))},
            "mmf": {"attn": keep(m.mmf.last_seed, 4, (B, H, T, T)), "out": keep(m.mmf.last_seed, 5, (B, T, C))}}
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    Yc = Y.clone().requires_grad_(True)
    ref = R.fusion_forward("TTF_T2V_XAttn", "MMF_XAttn_Add", p, notes, tau, t_hat, Yc, H=H, kappa=0.5, drop=drop, p_drop=pd,
                           expand_T=True)
    (ref * up).sum().backward()
    gerrs = {"gY": _relerr(Yg.grad, Yc.grad)}
    for k, prm in m.named_parameters():
        g = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
        gerrs["g." + k] = _relerr(prm.grad, g)
    _check({"out": _relerr(out, ref)}, 1e-4)
    _check(gerrs, 2e-4)


def test_cfg3_shape_patchtst_fusion_step():
    """BASELINE configs[2]: PatchTST + TTF_T2V_XAttn + MMF_GR_Add, LLaMA dims, 64 windows: one full training step
    (backbone -> fusion -> masked MSE -> backward) runs on the HIP path, bf16, finite loss and gradients."""
    dev = _dev()
    from fusions.FusionModel import FusionModel
    from fusions.load_llm import register_d_model
    from immtsf import config
    from immtsf.ops import masked_mse
    from models.PatchTST import PatchTST
    register_d_model("SYN", 4096)
    config.precision = "bf16"
    a = _args("TTF_T2V_XAttn", "MMF_GR_Add", "SYN", 768, 1, 6, dropout=0.1)
    a.input_len = a.pred_len = 32
    a.d_model, a.d_ff, a.n_heads, a.e_layers, a.factor, a.activation, a.enc_in, a.batch_size = 512, 2048, 2, 1, 5, "gelu", 6, 64
    torch.manual_seed(0)
    model, fusion = PatchTST(a).to(dev).train(), FusionModel(a).to(dev).train()
    notes, tau, t_hat, Y, up = _synthetic(2, 64, 32, 32, 6, 4096, dev)
    g = torch.Generator().manual_seed(3)
    data = torch.randn(64, 32, 6, generator=g).to(dev)
    mask = (torch.rand(64, 32, 6, generator=g) < 0.7).float().to(dev)
    tp = torch.sort(torch.rand(64, 32, generator=g), 1).values.to(dev)
    pred = model.forecasting(t_hat.to(dev), data * mask, tp, mask)
    out = fusion(notes.to(dev), tau.to(dev), t_hat.to(dev), pred)
    loss = masked_mse(out, Y.to(dev), (up.to(dev) > 0).float())
    loss.backward()
    config.precision = "fp32"
    assert out.shape == (64, 32, 6) and torch.isfinite(loss)
    for k, p in list(model.named_parameters()) + list(fusion.named_parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all(), k


def test_cfg4_shape_timesnet_fusion_step():
    """BASELINE configs[3] (one rank's shard): TimesNet + TTF_RecAvg + MMF_XAttn_Add, GDELT-shaped, 64 windows, C=8,
    input_len = pred_len = 32, TimesNet d_model=16, d_ff=32, top_k=5, e_layers=2 (main.py:852-858): one full training
    step on the HIP path in fp32 against the oracle for the fusion part (the backbone output is fed to both)."""
    dev = _dev()
    from fusions.FusionModel import FusionModel
    from fusions.load_llm import register_d_model
    from immtsf import config
    from immtsf.ops import masked_mse
    from models.TimesNet import TimesNet
    from oracle import fusion_ref as R
    register_d_model("SYN768", 768)
    config.precision = "fp32"
    a = _args("TTF_RecAvg", "MMF_XAttn_Add", "SYN768", 768, 1, 8, dropout=0.0)
    a.input_len = a.pred_len = 32
    a.d_model, a.d_ff, a.top_k, a.e_layers, a.num_kernels, a.enc_in, a.c_out, a.batch_size = 16, 32, 5, 2, 6, 8, 8, 64
    a.embed, a.freq = "fixed", "h"
    torch.manual_seed(0)
    model, fusion = TimesNet(a).to(dev).train(), FusionModel(a).to(dev).train()
    notes, tau, t_hat, Y, up = _synthetic(5, 64, 32, 32, 8, 768, dev)
    g = torch.Generator().manual_seed(4)
    data = torch.randn(64, 32, 8, generator=g).to(dev)
    mask = (torch.rand(64, 32, 8, generator=g) < 0.7).float().to(dev)
    tp = torch.sort(torch.rand(64, 32, generator=g), 1).values.to(dev)
    tmask = (up > 0).float()
    pred = model.forecasting(t_hat.to(dev), data * mask, tp, mask)
    out = fusion(notes.to(dev), tau.to(dev), t_hat.to(dev), pred)
    loss = masked_mse(out, Y.to(dev), tmask.to(dev))
    loss.backward()
    assert out.shape == (64, 32, 8) and torch.isfinite(loss)
    for k, p in fusion.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    got = [k for k, p in model.named_parameters() if p.grad is not None]      # the temporal embedding is unused (x_mark=None)
    assert got and all(torch.isfinite(dict(model.named_parameters())[k].grad).all() for k in got)
    prm = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in fusion.state_dict().items()}
    ref = R.fusion_forward("TTF_RecAvg", "MMF_XAttn_Add", prm, notes, tau, t_hat, pred.detach().cpu(), H=1, kappa=0.5,
                           expand_T=False)
    R.masked_mse(Y, ref, tmask).backward()
    assert _relerr(out, ref) < 1e-4
    _check({k: _relerr(q.grad, prm[k].grad) for k, q in fusion.named_parameters()}, 3e-4)


def test_cfg5_full_size_long_ragged_bf16():
    """BASELINE configs[4] at its full per-GPU size: 64 windows, N_b ~ U{1..4096} notes (Sum N ~ 130 k), d_m = 4096 ->
    d_txt = 768, T = 32, C = 8, bf16.  The oracle cannot run this (its K/V expansion alone is ~26 GB), so: finite loss
    and gradients, the ragged index equals the lengths the batch was built with, and the first 4 windows give the same
    result alone as inside the full batch (window independence at full size, bf16 tolerance)."""
    dev = _dev()
    from fusions.FusionModel import FusionModel
    from fusions.load_llm import register_d_model
    from immtsf import config
    register_d_model("SYN", 4096)
    config.precision = "bf16"
    try:
        B, N, T, C = 64, 4096, 32, 8
        torch.manual_seed(0)
        m = FusionModel(_args("TTF_T2V_XAttn", "MMF_XAttn_Add", "SYN", 768, 1, C)).to(dev).train()
        g = torch.Generator(device=dev).manual_seed(11)
        lengths = torch.randint(1, N + 1, (B,), generator=g, device=dev)
        lengths[0], lengths[1] = N, 1
        keep = torch.arange(N, device=dev).view(1, N) < lengths.view(B, 1)
        notes = torch.randn(B, N, 4096, generator=g, device=dev) * keep.unsqueeze(-1)
        tau = torch.sort(torch.rand(B, N, generator=g, device=dev) * 24.0, 1).values * keep
        t_hat = torch.sort(torch.rand(B, T, generator=g, device=dev), 1).values
        Y = torch.randn(B, T, C, generator=g, device=dev).requires_grad_(True)
        E, M = m.ttf(notes, tau, t_hat)
        assert bool(M.all())
        out = m.mmf(Y, E, M)
        out.square().mean().backward()
        assert torch.isfinite(out).all()
        for k, p in m.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
        with torch.no_grad():
            sub = m(notes[:4].contiguous(), tau[:4].contiguous(), t_hat[:4].contiguous(), Y[:4].detach().contiguous())
        assert _l2err(sub, out[:4]) < 3e-2
    finally:
        config.precision = "fp32"


def test_ragged_index_many_windows_bit_exact():
    """B > 256 takes the three-launch form of the index builder (count / scan / fill over the whole chip): lengths, offsets, row map
    and segment ids bit-exact against numpy, including windows without notes and masked notes that are not a suffix."""
    dev = _dev()
    from immtsf import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    B, N, d_m = 1500, 37, 8
    keep = rng.random((B, N)) < 0.6
    keep[rng.integers(0, B, 40)] = False
    notes = torch.from_numpy((rng.standard_normal((B, N, d_m)).astype(np.float32) + 3.0) * keep[..., None]).to(dev)
    mask = torch.zeros(B * N, dtype=torch.uint8, device=dev)
    lengths = torch.zeros(B, dtype=torch.int32, device=dev)
    offsets = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    rowmap = torch.full((B * N,), -1, dtype=torch.int32, device=dev)
    seg = torch.full((B * N,), -1, dtype=torch.int32, device=dev)
    mtxt = torch.zeros(B, dtype=torch.uint8, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.immtsf_ragged_index(_lib.ptr(notes), B, N, d_m, _lib.ptr(mask), _lib.ptr(lengths), _lib.ptr(offsets),
                                       _lib.ptr(rowmap), _lib.ptr(seg), _lib.ptr(mtxt), _lib.ptr(flag), _lib.stream_ptr()), "ragged_index")
    torch.cuda.synchronize()
    assert np.array_equal(mask.cpu().numpy().reshape(B, N).astype(bool), keep)
    assert np.array_equal(lengths.cpu().numpy(), keep.sum(1).astype(np.int32))
    assert np.array_equal(offsets.cpu().numpy(), np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.int32))
    exp = np.flatnonzero(keep.reshape(-1)).astype(np.int32)
    assert np.array_equal(rowmap.cpu().numpy()[:len(exp)], exp)
    assert np.array_equal(seg.cpu().numpy()[:len(exp)], exp // N)
    assert np.array_equal(mtxt.cpu().numpy().astype(bool), keep.any(1))


def test_timellm_offline_smoke():
    """TimeLLM wrapper with a random-init GPT-2 body (no hub access here): shapes and finite outputs/gradients only --
    full-forward parity with the reference is unpinned."""
    dev = _dev()
    from models.TimeLLM import TimeLLM
    cfg = types.SimpleNamespace(input_len=16, pred_len=8, use_norm=True, d_ff=32, ts_vocab_size=50, input_token_len=8,
                                stride=4, domain_des="synthetic", top_k=3, C=3, llm_model_timellm="GPT2",
                                llm_layers_timellm=2, dropout=0.0, d_model=16, n_heads=2, batch_size=4, device=str(dev),
                                immtsf_offline_llm=True)
    torch.manual_seed(0)
    m = TimeLLM(cfg).to(dev)
    m.word_embeddings = m.llm_model.get_input_embeddings().weight
    g = torch.Generator().manual_seed(1)
    data = torch.randn(3, 12, 3, generator=g).to(dev)
    mask = (torch.rand(3, 12, 3, generator=g) < 0.8).float().to(dev)
    tp = torch.sort(torch.rand(3, 12, generator=g), 1).values.to(dev)
    out = m.forecasting(torch.rand(3, 5, generator=g).to(dev), data * mask, tp, mask)
    assert out.shape == (3, 5, 3) and torch.isfinite(out).all()
    out.square().mean().backward()
    assert torch.isfinite(m.mapping_layer.weight.grad).all()


def test_timellm_forecasting_vs_reference_golden():
    """The TimeLLM wrapper end to end against the REFERENCE's forecasting (models/TimeLLM.py:167-278) run with its hub loader
    patched to a random-init 2-layer GPT-2 + byte tokenizer (tests/golden/make_golden.py gen_timellm): same seeded weights in
    every tensor (tests/golden/seeded.py), output 1e-4, gradient fingerprints of every trainable parameter 1e-3."""
    dev = _dev()
    sys.path.insert(0, GOLDEN)
    import seeded
    from immtsf import config
    from models.TimeLLM import TimeLLM
    config.precision = "fp32"
    z = np.load(os.path.join(GOLDEN, "model_timellm.npz"))
    cfg = types.SimpleNamespace(input_len=16, pred_len=8, use_norm=True, d_ff=32, ts_vocab_size=20, input_token_len=8, stride=4,
                                domain_des="synthetic", top_k=3, C=3, llm_model_timellm="GPT2", llm_layers_timellm=2, dropout=0.0,
                                d_model=16, n_heads=2, batch_size=4, device=str(dev),
                                immtsf_offline_llm=dict(vocab_size=320, n_positions=512, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0))
    m = TimeLLM(cfg)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if v.dtype.is_floating_point}
    assert sorted(shapes) == [str(k) for k in z["keys"]], "state_dict keys differ from the reference's"
    assert m._get_prompt(torch.zeros(1, 16, 3))[0] == str(z["prompt0"])          # the prompt text is byte-identical
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded.state_like(shapes, 6100).items()}, strict=False)
    assert not missing.unexpected_keys
    m = m.to(dev).train()
    m.word_embeddings = m.llm_model.get_input_embeddings().weight
    m.reprogramming_layer.dropout.p = 0.0            # as the fixture: the layer's own default 0.1 is the only dropout left in the path
    t = lambda k: torch.from_numpy(z[k]).to(dev)     # noqa: E731
    # The prompt spells the batch statistics with all float digits (reference :167-194), so a 1-ulp difference between the CPU's and
    # the GPU's normalisation changes the token string.  The statistics are compared as NUMBERS (same text skeleton, values to 1e-5);
    # downstream both sides then consume the reference's recorded strings, so the comparison pins the composition, not the last digit.
    import re
    ref_prompts = [str(q) for q in z["prompts"]]
    seen = []
    real_prompt = m._get_prompt
    m._get_prompt = lambda x: seen.append(real_prompt(x)) or ref_prompts
    out = m.forecasting(t("tpp"), t("data").clone(), t("tp"), t("mask"))
    num, lags = r"-?\d+\.\d+(?:e-?\d+)?", r"Top lags \[([\d, ]+)\]"
    for mine, ref in zip(seen[0], ref_prompts):
        # (the circular autocorrelation is symmetric, corr[k] == corr[L - k], so torch.topk's order inside such a pair is a tie
        # the CPU and the GPU break differently: the lag lists are compared as classes {k, L - k})
        skel = lambda q: re.sub(lags, "Top lags [*]", re.sub(num, "#", q))       # noqa: E731
        assert skel(mine) == skel(ref), (mine, ref)
        a, b = [float(v) for v in re.findall(num, mine)], [float(v) for v in re.findall(num, ref)]
        assert np.allclose(a, b, rtol=1e-5, atol=1e-6), (mine, ref)
        cls = lambda q: sorted(min(int(v), 16 - int(v)) for v in re.search(lags, q).group(1).split(","))      # noqa: E731
        assert cls(mine) == cls(ref), (mine, ref)
    assert _l2err(out, t("out")) < 1e-4
    (out * t("upstream")).sum().backward()
    for i, (k, p_) in enumerate(sorted(m.named_parameters())):
        if p_.requires_grad:
            got = seeded.probes(p_.grad.detach().cpu().numpy(), 6200 + i)
            ref = z[f"gp.{k}"]
            # (floor: the key projection's bias gradient is zero in exact arithmetic -- softmax is shift invariant -- and ~1e-9 noise
            # on both sides)
            assert np.abs(got - ref).max() <= 1e-3 * max(1e-4, np.abs(ref).max()), (k, got, ref)


def test_mmf_monolithic_entry_equals_the_two_halves():
    """immtsf_mmf_xattn_add_forward/backward (one call) and the key/value + query halves the module uses are the same
    computation: outputs and every gradient must agree to fp32 rounding."""
    dev = _dev()
    from fusions.MMF_XAttn_Add import MMF_XAttn_Add
    from immtsf import config
    from immtsf.ops import MMFXAttnAddFn
    config.precision = "fp32"
    torch.manual_seed(2)
    B, T, Cc, d, H = 5, 9, 3, 16, 2
    mmf = MMF_XAttn_Add(d, Cc, d, n_heads_fusion=H, dropout=0.0, kappa=0.7).to(dev).train()
    Y, E = torch.randn(B, T, Cc, device=dev), torch.randn(B, T, d, device=dev)
    M = torch.tensor([1, 1, 0, 1, 1], dtype=torch.bool, device=dev).view(B, 1)
    up = torch.randn(B, T, Cc, device=dev)
    res = []
    for mono in (False, True):
        mmf.zero_grad()
        y, e = Y.clone().requires_grad_(True), E.clone().requires_grad_(True)
        if mono:
            out = MMFXAttnAddFn.apply(y, e, M.view(B).view(torch.uint8), H, 0.7, 0.0, False, 0, 0, *mmf._params())
        else:
            out = mmf(y, e, M)
        (out * up).sum().backward()
        res.append([out.detach(), y.grad, e.grad] + [p.grad.clone() for p in mmf.parameters()])
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= 1e-5 * max(1e-3, float(b.abs().max()))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("B,T,Cc,d,H,pd", [(5, 7, 3, 32, 1, 0.0), (6, 32, 8, 64, 2, 0.2), (3, 13, 5, 48, 4, 0.3), (64, 32, 8, 768, 1, 0.1),
                                            (2, 260, 8, 64, 1, 0.1), (300, 6, 15, 16, 1, 0.0), (1, 1, 1, 4, 1, 0.0), (2, 3, 15, 16, 4, 0.25),
                                            (160, 32, 8, 64, 1, 0.1)])       # (the last: >= 4096 rows, the data gradient as a row kernel -- rank_expand)
def test_xattn_add_low_rank_form_equals_full_rank(B, T, Cc, d, H, pd, precision):
    """MMF_XAttn_Add's low-rank form (csrc/xrank.hip: the text side projected onto the (2C+1) H columns the attention needs, the
    attention + head as one kernel per direction, parameter gradients by the chain rule through the folded factors) against the
    full-rank key/value + query halves (immtsf.config.xattn_rank = False): same Philox sites and indices, so also under dropout;
    outputs, data gradients and every parameter gradient.  (Both are pinned to the reference by the goldens: the module takes the
    low-rank form wherever its limits allow.)"""
    dev = _dev()
    from fusions.MMF_XAttn_Add import MMF_XAttn_Add
    from immtsf import config
    config.precision = precision
    torch.manual_seed(B * 1000 + T)
    mmf = MMF_XAttn_Add(d, Cc, d, n_heads_fusion=H, dropout=pd, kappa=0.7).to(dev).train()
    with torch.no_grad():
        for p_ in mmf.parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    Y, E = torch.randn(B, T, Cc, device=dev), torch.randn(B, T, d, device=dev)
    M = (torch.rand(B, device=dev) > 0.25).view(B, 1)
    M[0] = True
    if B == 2 and T == 3:
        M[:] = False          # no window has text: the block degenerates to Y / (1 + kappa), every parameter gradient is zero
    up = torch.randn(B, T, Cc, device=dev)
    res, seed0 = [], config.next_seed
    try:
        config.next_seed = lambda: 4242
        for rank in (True, False):
            config.xattn_rank = rank
            assert mmf._rank(T) == rank
            mmf.zero_grad()
            y, e = Y.clone().requires_grad_(True), E.clone().requires_grad_(True)
            out = mmf(y, e, M)
            (out * up).sum().backward()
            res.append([("out", out.detach()), ("dY", y.grad), ("dE", e.grad)] + [(k, p_.grad.clone()) for k, p_ in mmf.named_parameters()])
    finally:
        config.next_seed, config.xattn_rank, config.precision = seed0, True, "fp32"
    tol = 2e-4 if precision == "fp32" else 4e-2
    gmax = max(float(b.abs().max()) for k, b in res[1][3:])
    for (k, a), (_, b) in zip(*res):
        assert torch.isfinite(a).all(), k
        # floor: the key projection's bias gradient is zero in exact arithmetic; small gradients are compared on the scale of the block's largest
        den = float(b.norm()) + (1e-3 * gmax * b.numel() ** 0.5 if k not in ("out", "dY", "dE") else 1e-6) + 1e-30
        err = float((a - b).norm()) / den
        assert err <= tol, (k, err)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("B,N,T,d_m,d,H,pd,packed", [(5, 6, 7, 48, 32, 1, 0.0, False), (6, 32, 32, 96, 64, 2, 0.2, False), (3, 17, 13, 64, 64, 4, 0.3, True),
                                                      (64, 32, 32, 768, 768, 1, 0.1, False), (64, 32, 32, 768, 768, 1, 0.1, True),
                                                      (4, 64, 9, 4096, 128, 1, 0.1, False), (7, 5, 32, 80, None, 2, 0.15, False),
                                                      (300, 9, 6, 32, 16, 1, 0.0, True)])
def test_t2v_folded_form_equals_the_chain_as_written(B, N, T, d_m, d, H, pd, packed, precision):
    """TTF_T2V_XAttn's folded form (csrc/t2v_fold.hip: scores as a mat-vec of the raw notes with a folded vector per head, one
    sum-of-notes x (H d) x (d_m + d/2) product in place of input_proj / KV_proj / in-projection / out_proj, parameter gradients by the
    chain rule through the folded factors) against the reference's GEMM chain as written (immtsf.config.t2v_form = "chain"): same Philox
    sites and indices, so also under dropout; E_txt, M_txt and every parameter gradient; padded and packed notes, with and without an
    input projection (d_txt = None), a window without notes.  (Both are pinned to the reference by the goldens: the module takes the
    folded form wherever its limits hold.)  reference: fusions/TTF_T2V_XAttn.py:120-182."""
    dev = _dev()
    import ctypes as C
    from fusions.TTF_T2V_XAttn import TTF_T2V_XAttn
    from fusions.load_llm import register_d_model
    from immtsf import _lib, config
    from immtsf.ops import PackedNotes, make_cfg
    register_d_model(f"FOLD{d_m}", d_m)
    config.precision = precision
    torch.manual_seed(B * 100 + N)
    ttf = TTF_T2V_XAttn(f"FOLD{d_m}", 6, n_heads_fusion=H, dropout=pd, d_txt=d).to(dev).train()
    dd = ttf.d_txt
    with torch.no_grad():
        for p_ in ttf.parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    g = torch.Generator().manual_seed(B + 17 * T)
    lengths = torch.randint(1, N + 1, (B,), generator=g)
    if B > 2:
        lengths[1] = 0            # a window without notes
    keep = torch.arange(N).view(1, -1) < lengths.view(-1, 1)
    notes = (torch.randn(B, N, d_m, generator=g) * keep.unsqueeze(-1)).to(dev)
    tau = (torch.sort(torch.rand(B, N, generator=g) * 24.0, dim=1).values * keep).to(dev)
    t_hat = torch.rand(B, T, generator=g).to(dev)
    up = torch.randn(B, T, dd, generator=g).to(dev)
    src = notes
    if packed:
        rows = torch.arange(B * N, device=dev, dtype=torch.int32).view(B, N)[keep.to(dev)].contiguous()
        src = PackedNotes(notes.reshape(B * N, d_m).contiguous(), rows, lengths.to(dev).to(torch.int32), N)
    cfg = make_cfg(B, N, T, 0, d_m, dd, H, 1 if precision == "bf16" else 0, True, pd, 0.0, 0, None)
    cfg.form = 2
    assert _lib.load().immtsf_ttf_t2v_xattn_folded(C.byref(cfg)) == 1          # these shapes are inside the folded form's limits
    cfg.form = 1
    assert _lib.load().immtsf_ttf_t2v_xattn_folded(C.byref(cfg)) == 0
    res, seed0 = [], config.next_seed
    try:
        config.next_seed = lambda: 9191
        for form in ("fold", "chain"):
            config.t2v_form = form
            ttf.zero_grad()
            E, M = ttf(src, tau, t_hat)
            (E * up).sum().backward()
            res.append([("E", E.detach()), ("M", M.float())] + [(k, p_.grad.clone()) for k, p_ in ttf.named_parameters()])
    finally:
        config.next_seed, config.t2v_form, config.precision = seed0, "auto", "fp32"
    tol = 2e-4 if precision == "fp32" else 4e-2
    gmax = max(float(b.abs().max()) for k, b in res[1][2:])
    for (k, a), (_, b) in zip(*res):
        assert torch.isfinite(a).all(), k
        if k == "M":
            assert torch.equal(a, b)
            continue
        # floor: the key bias gradient is zero in exact arithmetic; small gradients are compared on the scale of the block's largest
        den = float(b.norm()) + (1e-3 * gmax * b.numel() ** 0.5 if k != "E" else 1e-6) + 1e-30
        err = float((a - b).norm()) / den
        assert err <= tol, (k, err)


@pytest.mark.parametrize("B,N,T,d_m,d,pd,packed", [(5, 100, 7, 48, 32, 0.0, False), (4, 300, 32, 96, 64, 0.2, False), (3, 517, 13, 64, 64, 0.3, True),
                                                    (6, 32, 32, 768, 768, 0.1, True), (3, 1100, 32, 4096, 128, 0.1, False),
                                                    (7, 70, 32, 80, None, 0.15, False), (2, 4096, 32, 256, 64, 0.1, True)])
def test_t2v_mix_first_form_equals_the_chain_as_written(B, N, T, d_m, d, pd, packed):
    """TTF_T2V_XAttn's MIX-FIRST form for long windows (csrc/t2v_premix.hip: softmax weights as a bf16 matrix, the window's raw rows
    [embedding ; Time2Vec] mixed per forecast step on the MFMA, ONE B T-row product with the folded W_tot; backward: the weights'
    gradient as note x step products against dx W_tot, the score vector's gradient as one pass over the notes) against the reference's
    GEMM chain as written (config.t2v_form = "chain"), bf16 mode, same Philox sites and indices (so also under dropout): E_txt, M_txt,
    every parameter gradient; padded and packed notes, with and without an input projection, a window without notes, windows of one
    note and of N notes.  reference: fusions/TTF_T2V_XAttn.py:120-182."""
    dev = _dev()
    import ctypes as C
    from fusions.TTF_T2V_XAttn import TTF_T2V_XAttn
    from fusions.load_llm import register_d_model
    from immtsf import _lib, config
    from immtsf.ops import PackedNotes, make_cfg
    register_d_model(f"FOLD{d_m}", d_m)
    config.precision = "bf16"
    torch.manual_seed(B * 100 + N)
    ttf = TTF_T2V_XAttn(f"FOLD{d_m}", 6, n_heads_fusion=1, dropout=pd, d_txt=d).to(dev).train()
    dd = ttf.d_txt
    with torch.no_grad():
        for p_ in ttf.parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    g = torch.Generator().manual_seed(B + 17 * T)
    lengths = torch.randint(1, N + 1, (B,), generator=g)
    lengths[0] = N
    if B > 2:
        lengths[1] = 0            # a window without notes
        lengths[2] = 1
    keep = torch.arange(N).view(1, -1) < lengths.view(-1, 1)
    notes = (torch.randn(B, N, d_m, generator=g) * keep.unsqueeze(-1)).to(dev)
    tau = (torch.sort(torch.rand(B, N, generator=g) * 24.0, dim=1).values * keep).to(dev)
    t_hat = torch.rand(B, T, generator=g).to(dev)
    up = torch.randn(B, T, dd, generator=g).to(dev)
    src = notes
    if packed:
        rows = torch.arange(B * N, device=dev, dtype=torch.int32).view(B, N)[keep.to(dev)].contiguous()
        src = PackedNotes(notes.reshape(B * N, d_m).contiguous(), rows, lengths.to(dev).to(torch.int32), N)
    cfg = make_cfg(B, N, T, 0, d_m, dd, 1, 1, True, pd, 0.0, 0, None)
    cfg.form = 3
    assert _lib.load().immtsf_ttf_t2v_xattn_folded(C.byref(cfg)) == 1
    res, seed0 = [], config.next_seed
    try:
        config.next_seed = lambda: 9191
        for form, prec in (("mix", "bf16"), ("chain", "bf16"), ("chain", "fp32")):
            config.t2v_form, config.precision = form, prec
            ttf.zero_grad()
            E, M = ttf(src, tau, t_hat)
            (E * up).sum().backward()
            res.append([("E", E.detach()), ("M", M.float())] + [(k, p_.grad.clone()) for k, p_ in ttf.named_parameters()])
    finally:
        config.next_seed, config.t2v_form, config.precision = seed0, "auto", "fp32"
    gmax = max(float(b.abs().max()) for k, b in res[1][2:])
    for (k, a), (_, b) in zip(res[0], res[1]):
        assert torch.isfinite(a).all(), k
        if k == "M":
            assert torch.equal(a, b)
            continue
        den = float(b.norm()) + (1e-3 * gmax * b.numel() ** 0.5 if k != "E" else 1e-6) + 1e-30
        err = float((a - b).norm()) / den
        assert err <= 4e-2, (k, err)
    # ... and against the fp32 chain on every parameter's OWN scale (no floor from the block's largest gradient: with thousands of notes
    # per window the projections' gradients are a thousand times smaller than proj_out's, and a floor that size once hid a wrong one)
    for (k, a), (_, b) in zip(res[0], res[2]):
        if k == "M" or float(b.norm()) <= 1e-6 * gmax * b.numel() ** 0.5:          # (the key bias gradient: zero in exact arithmetic)
            continue
        err = float((a - b).norm() / b.norm())
        assert err <= 5e-2, (k, err, float(b.norm()))


@pytest.mark.parametrize("form", ["fold", "chain"])
def test_prebuilt_note_index_equals_the_derived_one(form):
    """PackedNotes.index() (immtsf_note_index_build: the batch's ragged index built once, by whoever builds the batch) is, array by array
    and bit for bit, what immtsf_ragged_index derives from the zero-padded tensor (a2), and TTF_T2V_XAttn on it (immtsf_fusion_cfg.
    note_index) returns the very E_txt, M_txt and gradients of the call that derives the index itself (config.note_index = False)."""
    dev = _dev()
    import ctypes as C
    from fusions.TTF_T2V_XAttn import TTF_T2V_XAttn
    from fusions.load_llm import register_d_model
    from immtsf import _lib, config
    from immtsf.ops import PackedNotes
    lib = _lib.load()
    B, N, T, d_m, d = 40, 11, 9, 64, 32
    register_d_model("IDX64", d_m)
    torch.manual_seed(3)
    ttf = TTF_T2V_XAttn("IDX64", 6, n_heads_fusion=2, dropout=0.0, d_txt=d).to(dev).train()
    g = torch.Generator().manual_seed(5)
    lengths = torch.randint(0, N + 1, (B,), generator=g)
    lengths[0], lengths[1] = N, 0
    keep = torch.arange(N).view(1, -1) < lengths.view(-1, 1)
    notes = ((torch.randn(B, N, d_m, generator=g) + 3.0) * keep.unsqueeze(-1)).to(dev)
    tau = (torch.sort(torch.rand(B, N, generator=g) * 24.0, dim=1).values * keep).to(dev)
    t_hat = torch.rand(B, T, generator=g).to(dev)
    up = torch.randn(B, T, d, generator=g).to(dev)
    rows = torch.arange(B * N, device=dev, dtype=torch.int32).view(B, N)[keep.to(dev)].contiguous()
    src = PackedNotes(notes.reshape(B * N, d_m).contiguous(), rows, lengths.to(dev).to(torch.int32), N)
    # the arrays against a2 on the padded tensor
    ix = src.index()[1]
    mask = torch.empty(B, N, dtype=torch.uint8, device=dev)
    i32 = lambda *s_: torch.empty(*s_, dtype=torch.int32, device=dev)      # noqa: E731
    le, of, rm, sg, mt = i32(B), i32(B + 1), i32(B * N), i32(B * N), torch.empty(B, dtype=torch.uint8, device=dev)
    _lib.check(lib.immtsf_ragged_index(_lib.ptr(notes), B, N, d_m, _lib.ptr(mask), _lib.ptr(le), _lib.ptr(of), _lib.ptr(rm), _lib.ptr(sg),
                                       _lib.ptr(mt), None, _lib.stream_ptr()), "ragged_index")
    torch.cuda.synchronize()
    total = int(of[B])
    assert torch.equal(ix["mask"].view(B, N), mask) and torch.equal(ix["mtxt"], mt) and torch.equal(ix["lengths"], le)
    assert torch.equal(ix["offsets"], of) and torch.equal(ix["rowmap"][:total], rm[:total]) and torch.equal(ix["seg"][:total], sg[:total])
    res = []
    try:
        config.t2v_form = form
        for use in (True, False):
            config.note_index = use
            ttf.zero_grad()
            E, M = ttf(src, tau, t_hat)
            (E * up).sum().backward()
            res.append([E.detach().clone(), M.clone()] + [p_.grad.clone() for p_ in ttf.parameters()])
    finally:
        config.t2v_form, config.note_index = "auto", True
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2:], res[1][2:]):          # (split-K weight gradients accumulate by atomics: equal up to their order)
        assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("form", ["fold", "chain"])
@pytest.mark.parametrize("B,N,T,d_m,d,H,Cc,pd", [(5, 6, 7, 48, 32, 1, 3, 0.0), (6, 32, 32, 96, 64, 2, 8, 0.2), (64, 32, 32, 768, 768, 1, 8, 0.1),
                                                 (144, 16, 32, 64, 64, 1, 8, 0.1)])      # (the last: >= 4096 rows -- rank_expand on the "_z" path, compact x_hat)
def test_fused_tail_equals_separate_blocks(B, N, T, d_m, d, H, Cc, pd, form, precision):
    """FusionModel with TTF_T2V_XAttn's proj_out composed into MMF_XAttn_Add's low-rank projection (immtsf.config.fuse_tail = True:
    the TTF block hands over Z, its LayerNorm + dropout output, and the P half computes [Z | 1] [W_fold W_po | W_fold b_po + b_fold]^T --
    csrc/xrank.hip immtsf_mmf_xrank_p_forward_z / _backward_data_z) against the two blocks as written (fuse_tail = False): output, dY_ts
    and every parameter gradient incl. proj_out's, both TTF forms, with dropout (same Philox sites).  reference:
    fusions/TTF_T2V_XAttn.py:177-182 feeding fusions/MMF_XAttn_Add.py:56-75."""
    dev = _dev()
    from fusions.FusionModel import FusionModel
    from fusions.load_llm import register_d_model
    from immtsf import config
    register_d_model(f"FT{d_m}", d_m)
    config.precision = precision
    torch.manual_seed(B * 10 + T)
    a = types.SimpleNamespace(TTF_module="TTF_T2V_XAttn", MMF_module="MMF_XAttn_Add", llm_model_fusion=f"FT{d_m}", llm_layers_fusion=6,
                              max_length=1024, device="cuda", use_text_embeddings=True, recency_sigma=1.0, n_heads_fusion=H, dropout=pd,
                              d_txt=d, C=Cc, kappa=0.5)
    m = FusionModel(a).to(dev).train()
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    g = torch.Generator().manual_seed(B + 3 * T)
    lengths = torch.randint(1, N + 1, (B,), generator=g)
    if B > 2:
        lengths[1] = 0
    keep = torch.arange(N).view(1, -1) < lengths.view(-1, 1)
    notes = (torch.randn(B, N, d_m, generator=g) * keep.unsqueeze(-1)).to(dev)
    tau = (torch.sort(torch.rand(B, N, generator=g) * 24.0, dim=1).values * keep).to(dev)
    t_hat = torch.rand(B, T, generator=g).to(dev)
    Y = torch.randn(B, T, Cc, generator=g).to(dev)
    up = torch.randn(B, T, Cc, generator=g).to(dev)
    res, seed0 = [], config.next_seed
    try:
        config.t2v_form = form
        for fuse in (True, False):
            config.fuse_tail = fuse
            seeds = iter([5151, 6262, 7373, 8484])
            config.next_seed = lambda: next(seeds)
            assert m.fused_tail(B * T, T) == fuse
            m.zero_grad()
            y = Y.clone().requires_grad_(True)
            out = m(notes, tau, t_hat, y)
            (out * up).sum().backward()
            res.append([("out", out.detach()), ("dY", y.grad)] + [(k, p_.grad.clone()) for k, p_ in m.named_parameters()])
    finally:
        config.next_seed, config.fuse_tail, config.t2v_form, config.precision = seed0, "auto", "auto", "fp32"
    tol = 2e-4 if precision == "fp32" else 4e-2
    gmax = max(float(b.abs().max()) for k, b in res[1][2:])
    for (k, a_), (_, b_) in zip(*res):
        assert torch.isfinite(a_).all(), k
        den = float(b_.norm()) + (1e-3 * gmax * b_.numel() ** 0.5 if k not in ("out", "dY") else 1e-6) + 1e-30
        err = float((a_ - b_).norm()) / den
        # (bf16: Time2Vec's two scalar gradients are sums over every note with heavy cancellation -- the bar tests/test_gpu_train.py's
        # oracle comparison uses for them)
        assert err <= (2.5e-1 if precision == "bf16" and k.startswith("ttf.time2vec.linear") else tol), (k, err)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("form", ["fold", "chain"])
@pytest.mark.parametrize("B,N,T,d_m,d,H,Cc,pd", [(64, 32, 32, 768, 768, 1, 8, 0.1), (40, 20, 16, 256, 256, 2, 4, 0.2), (20, 40, 32, 128, 1024, 4, 3, 0.1),
                                                 (300, 12, 30, 64, 512, 1, 15, 0.0)])      # (the last: >= 8192 rows -- four rows per wave)
def test_z_handover_equals_dense_handover(B, N, T, d_m, d, H, Cc, pd, form, precision):
    """the fused tail with immtsf.config.z_handover (Z leaves TTF_T2V_XAttn as its bf16 image alone -- IMMTSF_FORM_HALF_OUT --, the gradient
    comes back as the pair (dP, Wc) -- IMMTSF_FORM_LOWRANK_OUT / immtsf_fusion_cfg.lr_grad -- and dZ = dP Wc is formed inside the LayerNorm
    backward, csrc/rowops.hip layernorm_bwd_lr_kernel) against the dense hand-over in both directions: output, dY_ts and every parameter
    gradient, both TTF forms, with dropout.  reference: fusions/TTF_T2V_XAttn.py:176-182 feeding fusions/MMF_XAttn_Add.py:56-75."""
    dev = _dev()
    from fusions.FusionModel import FusionModel
    from fusions.load_llm import register_d_model
    from immtsf import config, ops
    register_d_model(f"FT{d_m}", d_m)
    config.precision = precision
    torch.manual_seed(B * 10 + T)
    a = types.SimpleNamespace(TTF_module="TTF_T2V_XAttn", MMF_module="MMF_XAttn_Add", llm_model_fusion=f"FT{d_m}", llm_layers_fusion=6,
                              max_length=1024, device="cuda", use_text_embeddings=True, recency_sigma=1.0, n_heads_fusion=H, dropout=pd,
                              d_txt=d, C=Cc, kappa=0.5)
    m = FusionModel(a).to(dev).train()
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    g = torch.Generator().manual_seed(B + 3 * T)
    lengths = torch.randint(1, N + 1, (B,), generator=g)
    lengths[1] = 0
    keep = torch.arange(N).view(1, -1) < lengths.view(-1, 1)
    notes = (torch.randn(B, N, d_m, generator=g) * keep.unsqueeze(-1)).to(dev)
    tau = (torch.sort(torch.rand(B, N, generator=g) * 24.0, dim=1).values * keep).to(dev)
    t_hat = torch.rand(B, T, generator=g).to(dev)
    Y = torch.randn(B, T, Cc, generator=g).to(dev)
    up = torch.randn(B, T, Cc, generator=g).to(dev)
    res, seed0, took = [], config.next_seed, []
    put0 = ops._reg_put
    try:
        config.t2v_form, config.fuse_tail = form, True
        for ho in (True, False):
            config.z_handover = ho
            seeds = iter([5151, 6262, 7373, 8484])
            config.next_seed = lambda: next(seeds)
            n_lr = [0]

            def counting_put(reg, t, payload, n_lr=n_lr):
                if reg is ops._lowrank:
                    n_lr[0] += 1
                return put0(reg, t, payload)
            ops._reg_put = counting_put
            m.zero_grad()
            y = Y.clone().requires_grad_(True)
            out = m(notes, tau, t_hat, y)
            (out * up).sum().backward()
            took.append(n_lr[0])
            res.append([("out", out.detach()), ("dY", y.grad)] + [(k, p_.grad.clone()) for k, p_ in m.named_parameters()])
    finally:
        ops._reg_put = put0
        config.next_seed, config.fuse_tail, config.t2v_form, config.precision, config.z_handover = seed0, "auto", "auto", "fp32", True
    # the low-rank gradient was really taken where the library says it can be (fold: the bf16 x_hat image only), and never without the knob
    assert took[1] == 0 and took[0] == (1 if (form == "chain" or precision == "bf16") else 0), took
    assert not ops._lowrank or ops._lowrank[-1][1] != 0
    tol = 2e-4 if precision == "fp32" else 4e-2
    gmax = max(float(b.abs().max()) for k, b in res[1][2:])
    for (k, a_), (_, b_) in zip(*res):
        assert torch.isfinite(a_).all(), k
        den = float(b_.norm()) + (1e-3 * gmax * b_.numel() ** 0.5 if k not in ("out", "dY") else 1e-6) + 1e-30
        err = float((a_ - b_).norm()) / den
        assert err <= (2.5e-1 if precision == "bf16" and k.startswith("ttf.time2vec.linear") else tol), (k, err)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("B,T,Cc,d,H,pd", [(5, 7, 3, 32, 1, 0.0), (6, 32, 8, 64, 2, 0.2), (64, 32, 8, 768, 1, 0.1), (300, 6, 15, 16, 1, 0.1)])
def test_xattn_add_head_loss_backward_in_one_launch(B, T, Cc, d, H, pd, precision):
    """MMF_XAttn_Add.forward_loss (immtsf_mmf_xrank_q_train: the Q half, the masked MSE with known observation counts and the backward
    of both as ONE kernel) against masked_mse(forward(...)) + autograd: the loss and every gradient, with dropout (same Philox sites)."""
    dev = _dev()
    from fusions.MMF_XAttn_Add import MMF_XAttn_Add
    from immtsf import config
    from immtsf.ops import backward_unit
    config.precision = precision
    torch.manual_seed(B * 7 + T)
    mmf = MMF_XAttn_Add(d, Cc, d, n_heads_fusion=H, dropout=pd, kappa=0.6).to(dev).train()
    Y, E = torch.randn(B, T, Cc, device=dev), torch.randn(B, T, d, device=dev)
    M = (torch.rand(B, device=dev) > 0.25).view(B, 1)
    M[0] = True
    truth = torch.randn(B, T, Cc, device=dev)
    mask = (torch.rand(B, T, Cc, device=dev) > 0.4).float()
    mask[..., 0] = 0.0 if Cc > 2 else mask[..., 0]        # a variable without observations: out of the mean
    cnt = mask.reshape(-1, Cc).sum(0)
    res, seed0 = [], config.next_seed
    try:
        config.next_seed = lambda: 777
        for fused in (True, False):
            config.xattn_fused_loss = fused
            mmf.zero_grad()
            y, e = Y.clone().requires_grad_(True), E.clone().requires_grad_(True)
            loss = mmf.forward_loss(y, e, M, truth, mask, cnt)
            if fused:
                backward_unit(loss)
            else:
                loss.backward()
            res.append([("loss", loss.detach().reshape(1)), ("dY", y.grad), ("dE", e.grad)] + [(k, p_.grad.clone()) for k, p_ in mmf.named_parameters()])
        # a seed other than backward_unit's: the stored gradients are scaled
        config.xattn_fused_loss = True
        mmf.zero_grad()
        y = Y.clone().requires_grad_(True)
        (3.0 * mmf.forward_loss(y, E, M, truth, mask, cnt)).backward()
        assert float((y.grad - 3.0 * res[0][1][1]).abs().max()) <= 1e-5 * float(res[0][1][1].abs().max())
        # ... also the ones that went straight into FlatTrainer's gradient sinks (LayerNorm's two: written by the forward with the
        # unit seed, rescaled in place by a backward with another seed -- round-3 advisor finding)
        from immtsf.train import FlatTrainer
        ref_g = {k: b.clone() for k, b in res[0][3:]}
        tr = FlatTrainer([list(mmf.parameters())], sink_buckets=(0,), overlap=False)
        try:
            tr.zero_grad()
            y = Y.clone().requires_grad_(True)
            (3.0 * mmf.forward_loss(y, E, M, truth, mask, cnt)).backward()
            torch.cuda.synchronize()
            for k, p_ in mmf.named_parameters():
                got, want = p_._immtsf_grad_sink, 3.0 * ref_g[k]
                assert float((got - want).abs().max()) <= (1e-4 if precision == "fp32" else 3e-2) * max(float(want.abs().max()), 1e-6), k
        finally:
            tr.close()
    finally:
        config.next_seed, config.xattn_fused_loss, config.precision = seed0, True, "fp32"
    tol = 1e-4 if precision == "fp32" else 2e-2
    gmax = max(float(b.abs().max()) for k, b in res[1][3:])
    for (k, a), (_, b) in zip(*res):
        assert torch.isfinite(a).all(), k
        den = float(b.norm()) + (1e-3 * gmax * b.numel() ** 0.5 if k not in ("loss", "dY", "dE") else 1e-9)
        err = float((a - b).norm()) / den
        assert err <= tol, (k, err)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("B,T,Cc,d,pd", [(5, 7, 3, 32, 0.0), (6, 32, 6, 768, 0.2), (64, 32, 6, 768, 0.1), (160, 32, 15, 64, 0.1), (3, 64, 16, 16, 0.0)])
def test_gr_add_split_form_loss_backward_in_one_launch(B, T, Cc, d, pd, precision):
    """MMF_GR_Add in split form (csrc/gr_train.hip: the text columns of the GRU's input map and of the gate net as the P half --
    project_kv --, the Y half + masked MSE + the backward of both through time as ONE launch -- forward_loss) against the block as
    written + masked_mse + autograd (immtsf.config.gr_split = False): the loss, dY_ts, dE_txt and every parameter gradient, with dropout
    (same Philox site), windows without text, a variable without observations; gradient sinks; the output-only launch under no_grad.
    reference: fusions/MMF_GR_Add.py:31-61."""
    dev = _dev()
    from fusions.MMF_GR_Add import MMF_GR_Add
    from immtsf import config
    from immtsf.ops import backward_unit
    config.precision = precision
    torch.manual_seed(B * 7 + T)
    mmf = MMF_GR_Add(d, Cc, Cc, dropout=pd).to(dev).train()
    Y, E = torch.randn(B, T, Cc, device=dev), torch.randn(B, T, d, device=dev)
    M = (torch.rand(B, device=dev) > 0.25).view(B, 1)
    M[0] = True
    truth = torch.randn(B, T, Cc, device=dev)
    mask = (torch.rand(B, T, Cc, device=dev) > 0.4).float()
    mask[..., 0] = 0.0 if Cc > 2 else mask[..., 0]
    cnt = mask.reshape(-1, Cc).sum(0)
    res, seed0 = [], config.next_seed
    try:
        config.next_seed = lambda: 777
        for split in (True, False):
            config.gr_split = split
            mmf.zero_grad()
            y, e = Y.clone().requires_grad_(True), E.clone().requires_grad_(True)
            kv = mmf.project_kv(e)
            assert (kv[0] is not None) == split
            loss = mmf.forward_loss(y, e, M, truth, mask, cnt, kv=kv)
            if split:
                backward_unit(loss)
            else:
                loss.backward()
            res.append([("loss", loss.detach().reshape(1)), ("dY", y.grad), ("dE", e.grad)] + [(k, p_.grad.clone()) for k, p_ in mmf.named_parameters()])
        config.gr_split = True
        # a seed other than backward_unit's scales the stored gradients
        mmf.zero_grad()
        y = Y.clone().requires_grad_(True)
        (3.0 * mmf.forward_loss(y, E, M, truth, mask, cnt)).backward()
        assert float((y.grad - 3.0 * res[0][1][1]).abs().max()) <= 1e-5 * float(res[0][1][1].abs().max())
        # gradient sinks: the two halves write / add their columns of d W_ih and d W_g into the same pre-zeroed buffers
        from immtsf.train import FlatTrainer
        ref_g = {k: b.clone() for k, b in res[0][3:]}
        tr = FlatTrainer([list(mmf.parameters())], sink_buckets=(0,), overlap=False)
        try:
            tr.zero_grad()
            y, e = Y.clone().requires_grad_(True), E.clone().requires_grad_(True)
            (3.0 * mmf.forward_loss(y, e, M, truth, mask, cnt)).backward()
            torch.cuda.synchronize()
            for k, p_ in mmf.named_parameters():
                got, want = p_._immtsf_grad_sink, 3.0 * ref_g[k]
                assert float((got - want).abs().max()) <= (1e-4 if precision == "fp32" else 3e-2) * max(float(want.abs().max()), 1e-6), k
        finally:
            tr.close()
        # output only (no gradient wanted): the split form's launch against the block as written
        mmf.eval()
        with torch.no_grad():
            out_split = mmf(Y, E, M, kv=mmf.project_kv(E))
            config.gr_split = False
            out_ref = mmf(Y, E, M)
        assert float((out_split - out_ref).abs().max()) <= (1e-4 if precision == "fp32" else 3e-2) * float(out_ref.abs().max())
    finally:
        config.next_seed, config.gr_split, config.precision = seed0, True, "fp32"
    tol = 1e-4 if precision == "fp32" else 2e-2
    gmax = max(float(b.abs().max()) for k, b in res[1][3:])
    for (k, a), (_, b) in zip(*res):
        assert torch.isfinite(a).all(), k
        den = float(b.norm()) + (1e-3 * gmax * b.numel() ** 0.5 if k not in ("loss", "dY", "dE") else 1e-9)
        err = float((a - b).norm()) / den
        assert err <= tol, (k, err)
