#!/usr/bin/env python3
"""Generate golden input/output/gradient vectors from the REAL reference.

Run in the build container only (it needs /root/reference, which does not
exist on the GPU box):

    python tests/golden/make_golden.py

It imports the unmodified reference modules (fusions/*, layers/*, lib/evaluation,
models/tPatchGNN) with the two shims SURVEY.md section 8(c) describes:

  * empty stub modules for third-party packages that are absent here and are
    only touched by out-of-scope code (reformer_pytorch, ...);
  * `get_d_model` in the TTF module namespaces replaced by a table lookup (the
    real one calls the HF hub, which is unreachable).

Nothing of the reference's source is written anywhere: the outputs are tensors
only (state_dict, inputs, outputs, gradients), saved as small .npz files next
to this script.  Every fixture is float32/bool/int -- data, not code.
"""
import os
import sys
import types
import importlib

import numpy as np
import torch

REF = os.environ.get("IMMTSF_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

D_MODEL_TABLE = {"GPT2": 768, "BERT": 768, "Llama": 4096, "DeepSeek": 4096,
                 "TOY16": 16, "TOY48": 48}


def _install_shims():
    for name in ["reformer_pytorch", "stribor", "geotorch", "torchdiffeq", "seaborn", "prettytable"]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.LSHSelfAttention = object
            m.PrettyTable = object
            sys.modules[name] = m
    os.environ.setdefault("HF_HUB_OFFLINE", "1")
    sys.path.insert(0, REF)


def _ref_modules():
    _install_shims()
    import fusions.load_llm as ll
    ll.get_d_model = lambda name: D_MODEL_TABLE[name]
    mods = {}
    for n in ["fusions.TTF_RecAvg", "fusions.TTF_T2V_XAttn", "fusions.MMF_GR_Add",
              "fusions.MMF_XAttn_Add", "fusions.FusionModel"]:
        mods[n] = importlib.import_module(n)
    mods["fusions.TTF_RecAvg"].get_d_model = ll.get_d_model
    mods["fusions.TTF_T2V_XAttn"].get_d_model = ll.get_d_model
    return mods


def _np(t):
    return t.detach().cpu().numpy()


def make_batch(seed, B, N, T, C, d_m, lengths, t_hat_1d=False):
    """Synthetic collate-shaped batch: zero-padded notes, raw tau, normalised t_hat."""
    g = torch.Generator().manual_seed(seed)
    notes = torch.randn(B, N, d_m, generator=g)
    tau = torch.zeros(B, N)
    for b, L in enumerate(lengths):
        notes[b, L:] = 0.0
        if L > 0:
            tau[b, :L] = torch.sort(torch.rand(L, generator=g) * 24.0).values
    if t_hat_1d:
        t_hat = torch.sort(torch.rand(T, generator=g)).values
    else:
        t_hat = torch.sort(torch.rand(B, T, generator=g), dim=1).values
    Y_ts = torch.randn(B, T, C, generator=g)
    return notes, tau, t_hat, Y_ts


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB, {len(arrays)} arrays)")


def run_module(mod, inputs, grad_inputs=(), upstream_seed=123, train=True):
    """forward (eval), forward (train, dropout=0) + backward with a fixed upstream gradient."""
    out = {}
    mod.eval()
    with torch.no_grad():
        r = mod(*inputs)
    rs = r if isinstance(r, tuple) else (r,)
    for i, t in enumerate(rs):
        out[f"out_eval.{i}"] = _np(t)
    if train:
        mod.train()
        ins = [x.clone().requires_grad_(True) if (i in grad_inputs) else x for i, x in enumerate(inputs)]
        r = mod(*ins)
        rs = r if isinstance(r, tuple) else (r,)
        for i, t in enumerate(rs):
            out[f"out_train.{i}"] = _np(t)
        g = torch.Generator().manual_seed(upstream_seed)
        up = torch.randn(rs[0].shape, generator=g)
        out["upstream"] = _np(up)
        (rs[0] * up).sum().backward()
        for k, p in mod.named_parameters():
            out[f"g.{k}"] = _np(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        for i in grad_inputs:
            out[f"gin.{i}"] = _np(ins[i].grad)
    return out


def gen_fusion(mods):
    TTF_T2V = mods["fusions.TTF_T2V_XAttn"].TTF_T2V_XAttn
    TTF_Rec = mods["fusions.TTF_RecAvg"].TTF_RecAvg
    MMF_X = mods["fusions.MMF_XAttn_Add"].MMF_XAttn_Add
    MMF_G = mods["fusions.MMF_GR_Add"].MMF_GR_Add
    FusionModel = mods["fusions.FusionModel"].FusionModel

    cases = [
        # name, llm, d_txt, H, B, N, T, C, lengths, t_hat_1d
        ("tiny_h1", "TOY16", 8, 1, 4, 5, 6, 3, [5, 3, 1, 2], False),
        ("tiny_h2", "TOY16", 8, 2, 4, 5, 6, 3, [5, 3, 1, 2], True),
        ("noproj_h2", "TOY16", None, 2, 3, 4, 5, 5, [4, 2, 3], False),
        ("mid_h4", "TOY48", 32, 4, 3, 7, 5, 4, [7, 1, 4], False),
    ]
    for (name, llm, d_txt, H, B, N, T, C, lengths, t1d) in cases:
        notes, tau, t_hat, Y_ts = make_batch(sum(map(ord, name)) + 7, B, N, T, C, D_MODEL_TABLE[llm], lengths, t1d)
        meta = dict(notes=_np(notes), tau=_np(tau), t_hat=_np(t_hat), Y_ts=_np(Y_ts),
                    lengths=np.asarray(lengths, np.int32),
                    H=np.int32(H), d_txt=np.int32(-1 if d_txt is None else d_txt),
                    d_m=np.int32(D_MODEL_TABLE[llm]))
        # --- TTF blocks
        for tname, ctor in [
            ("ttf_t2v", lambda: TTF_T2V(llm, 6, n_heads_fusion=H, dropout=0.0, d_txt=d_txt)),
            ("ttf_rec", lambda: TTF_Rec(llm, 6, recency_sigma=0.7, dropout=0.0, d_txt=d_txt)),
        ]:
            torch.manual_seed(11)
            m = ctor()
            with torch.no_grad():   # de-trivialise LayerNorm affine so its grads are tested
                m.layer_norm.weight.uniform_(0.5, 1.5)
                m.layer_norm.bias.uniform_(-0.3, 0.3)
            arrs = dict(meta)
            for k, v in m.state_dict().items():
                arrs[f"p.{k}"] = _np(v)
            arrs.update(run_module(m, (notes, tau, t_hat)))
            note_mask = (notes.abs().sum(2) > 0)
            arrs["note_mask"] = _np(note_mask)
            arrs["offsets"] = np.concatenate([[0], np.cumsum(_np(note_mask.sum(1)))]).astype(np.int32)
            save(f"{tname}_{name}", **arrs)
            E_txt = torch.from_numpy(arrs["out_eval.0"])
            M_txt = torch.from_numpy(arrs["out_eval.1"])
        # --- MMF blocks (driven by an independent E_txt so their grads w.r.t. E are pinned too)
        d = D_MODEL_TABLE[llm] if d_txt is None else d_txt
        g = torch.Generator().manual_seed(5)
        E_in = torch.randn(B, T, d, generator=g)
        M_in = torch.tensor([[L > 0] for L in lengths])
        for mname, ctor in [
            ("mmf_xattn", lambda: MMF_X(d, C, d, n_heads_fusion=H, dropout=0.0, kappa=0.5)),
            ("mmf_gr", lambda: MMF_G(d, C, C, dropout=0.0)),
        ]:
            torch.manual_seed(13)
            m = ctor()
            with torch.no_grad():
                m.layer_norm.weight.uniform_(0.5, 1.5)
                m.layer_norm.bias.uniform_(-0.3, 0.3)
            arrs = dict(Y_ts=_np(Y_ts), E_txt=_np(E_in), M_txt=_np(M_in), H=np.int32(H), kappa=np.float32(0.5))
            for k, v in m.state_dict().items():
                arrs[f"p.{k}"] = _np(v)
            arrs.update(run_module(m, (Y_ts, E_in, M_in), grad_inputs=(0, 1)))
            save(f"{mname}_{name}", **arrs)
        # --- composite FusionModel, all 4 pairs
        for ttf in ["TTF_RecAvg", "TTF_T2V_XAttn"]:
            for mmf in ["MMF_GR_Add", "MMF_XAttn_Add"]:
                args = types.SimpleNamespace(
                    TTF_module=ttf, MMF_module=mmf, llm_model_fusion=llm, llm_layers_fusion=6,
                    max_length=1024, device="cpu", use_text_embeddings=True, recency_sigma=1.3,
                    n_heads_fusion=H, dropout=0.0, d_txt=d_txt, C=C, kappa=0.5)
                torch.manual_seed(17)
                m = FusionModel(args)
                arrs = dict(meta)
                arrs["kappa"] = np.float32(0.5)
                arrs["recency_sigma"] = np.float32(1.3)
                for k, v in m.state_dict().items():
                    arrs[f"p.{k}"] = _np(v)
                arrs.update(run_module(m, (notes, tau, t_hat, Y_ts), grad_inputs=(3,)))
                save(f"fusion_{ttf}_{mmf}_{name}", **arrs)

    # --- zero-note sample: forward only (the reference's backward is NaN there, SURVEY 7 "hard parts")
    name, llm, d_txt, H, B, N, T, C, lengths = "zeronote", "TOY16", 8, 2, 4, 5, 6, 3, [5, 0, 1, 2]
    notes, tau, t_hat, Y_ts = make_batch(99, B, N, T, C, 16, lengths)
    for ttf in ["TTF_RecAvg", "TTF_T2V_XAttn"]:
        for mmf in ["MMF_GR_Add", "MMF_XAttn_Add"]:
            args = types.SimpleNamespace(
                TTF_module=ttf, MMF_module=mmf, llm_model_fusion=llm, llm_layers_fusion=6,
                max_length=1024, device="cpu", use_text_embeddings=True, recency_sigma=1.0,
                n_heads_fusion=H, dropout=0.0, d_txt=d_txt, C=C, kappa=0.5)
            torch.manual_seed(19)
            m = FusionModel(args)
            arrs = dict(notes=_np(notes), tau=_np(tau), t_hat=_np(t_hat), Y_ts=_np(Y_ts),
                        lengths=np.asarray(lengths, np.int32), H=np.int32(H), d_txt=np.int32(d_txt),
                        d_m=np.int32(16), kappa=np.float32(0.5), recency_sigma=np.float32(1.0))
            for k, v in m.state_dict().items():
                arrs[f"p.{k}"] = _np(v)
            m.eval()
            with torch.no_grad():
                E, M = m.ttf(notes, tau, t_hat)
                Y = m(notes, tau, t_hat, Y_ts)
            arrs["out_eval.0"] = _np(Y)
            arrs["E_txt"] = _np(E)
            arrs["M_txt"] = _np(M)
            # record the documented quirk: reference grads are NaN when a sample has no notes
            m.train()
            Y = m(notes, tau, t_hat, Y_ts)
            Y.sum().backward()
            arrs["ref_grad_has_nan"] = np.asarray(
                any(bool(torch.isnan(p.grad).any()) for p in m.parameters() if p.grad is not None))
            save(f"fusion_{ttf}_{mmf}_{name}", **arrs)


def gen_loss():
    _install_shims()
    # lib.evaluation imports sklearn/tqdm (present) and lib.utils
    ev = importlib.import_module("lib.evaluation")
    g = torch.Generator().manual_seed(3)
    B, T, C = 5, 7, 4
    truth = torch.randn(B, T, C, generator=g)
    pred = torch.randn(B, T, C, generator=g).requires_grad_(True)
    mask = (torch.rand(B, T, C, generator=g) < 0.6).float()
    mask[:, :, 2] = 0.0        # a variable with no observation at all
    loss = ev.compute_error(truth, pred, mask, "MSE", "mean")
    loss.backward()
    es, mc = ev.compute_error(truth, pred.detach(), mask, "MSE", "sum")
    mae = ev.compute_error(truth, pred.detach(), mask, "MAE", "mean")
    save("loss_mse", truth=_np(truth), pred=_np(pred), mask=_np(mask), loss=_np(loss),
         dpred=_np(pred.grad), err_sum=_np(es), mask_count=_np(mc), mae=_np(mae))


def gen_layers():
    _install_shims()
    SA = importlib.import_module("layers.SelfAttention_Family")
    EM = importlib.import_module("layers.Embed")
    TE = importlib.import_module("layers.Transformer_EncDec")
    g = torch.Generator().manual_seed(21)

    # FullAttention (mask_flag False, dropout 0) and AttentionLayer
    B, L, S, H, E = 3, 5, 7, 2, 4
    q = torch.randn(B, L, H, E, generator=g)
    k = torch.randn(B, S, H, E, generator=g)
    v = torch.randn(B, S, H, E, generator=g)
    fa = SA.FullAttention(False, attention_dropout=0.0)
    qq, kk, vv = [t.clone().requires_grad_(True) for t in (q, k, v)]
    o, _ = fa(qq, kk, vv, None)
    up = torch.randn(o.shape, generator=g)
    (o * up).sum().backward()
    save("layer_full_attention", q=_np(q), k=_np(k), v=_np(v), out=_np(o), upstream=_np(up),
         gq=_np(qq.grad), gk=_np(kk.grad), gv=_np(vv.grad))

    d_model = 8
    torch.manual_seed(23)
    al = SA.AttentionLayer(SA.FullAttention(False, attention_dropout=0.0), d_model, H)
    x = torch.randn(B, L, d_model, generator=g)
    xx = x.clone().requires_grad_(True)
    o, _ = al(xx, xx, xx, None)
    up = torch.randn(o.shape, generator=g)
    (o * up).sum().backward()
    arrs = dict(x=_np(x), out=_np(o), upstream=_np(up), gx=_np(xx.grad), H=np.int32(H))
    for kname, p in al.named_parameters():
        arrs[f"p.{kname}"] = _np(p)
        arrs[f"g.{kname}"] = _np(p.grad)
    save("layer_attention_layer", **arrs)

    # EncoderLayer + Encoder (gelu, d_ff), dropout 0
    torch.manual_seed(29)
    enc = TE.Encoder([TE.EncoderLayer(SA.AttentionLayer(SA.FullAttention(False, attention_dropout=0.0), d_model, H),
                                      d_model, 16, dropout=0.0, activation="gelu") for _ in range(2)],
                     norm_layer=torch.nn.LayerNorm(d_model))
    xx = x.clone().requires_grad_(True)
    o, _ = enc(xx)
    up = torch.randn(o.shape, generator=g)
    (o * up).sum().backward()
    arrs = dict(x=_np(x), out=_np(o), upstream=_np(up), gx=_np(xx.grad), H=np.int32(H))
    for kname, p in enc.named_parameters():
        arrs[f"p.{kname}"] = _np(p)
        arrs[f"g.{kname}"] = _np(p.grad)
    save("layer_encoder", **arrs)

    # PatchEmbedding (PatchTST geometry scaled down: patch_len 6, stride 3)
    torch.manual_seed(31)
    pe = EM.PatchEmbedding(d_model, 6, 3, 3, 0.0)
    xs = torch.randn(2, 3, 18, generator=g)
    xx = xs.clone().requires_grad_(True)
    o, n_vars = pe(xx)
    up = torch.randn(o.shape, generator=g)
    (o * up).sum().backward()
    save("layer_patch_embedding", x=_np(xs), out=_np(o), upstream=_np(up), gx=_np(xx.grad),
         w=_np(pe.value_embedding.weight), gw=_np(pe.value_embedding.weight.grad),
         n_vars=np.int32(n_vars), patch_len=np.int32(6), stride=np.int32(3))

    # DataEmbedding (x_mark None)
    torch.manual_seed(37)
    de = EM.DataEmbedding(5, d_model, dropout=0.0)
    xs = torch.randn(2, 9, 5, generator=g)
    xx = xs.clone().requires_grad_(True)
    o = de(xx, None)
    up = torch.randn(o.shape, generator=g)
    (o * up).sum().backward()
    save("layer_data_embedding", x=_np(xs), out=_np(o), upstream=_np(up), gx=_np(xx.grad),
         w=_np(de.value_embedding.tokenConv.weight), gw=_np(de.value_embedding.tokenConv.weight.grad))


def gen_tpatchgnn():
    _install_shims()
    torch.Tensor.cuda = lambda self, *a, **k: self      # models/tPatchGNN.py:131-132 hard-codes .cuda()
    TP = importlib.import_module("models.tPatchGNN")
    args = types.SimpleNamespace(device="cpu", hid_dim=8, C=3, npatch=2, nlayer=1, te_dim=4, n_heads=1,
                                 tf_layer=1, node_dim=4, hop=1, outlayer="Linear")
    torch.manual_seed(41)
    m = TP.tPatchGNN(args)
    m.eval()    # nn.TransformerEncoderLayer has dropout 0.1 inside; eval for determinism
    g = torch.Generator().manual_seed(43)
    B, M, L, N, Lp = 3, 2, 6, 3, 5
    X = torch.randn(B, M, L, N, generator=g)
    tt = torch.rand(B, M, L, N, generator=g)
    mask = (torch.rand(B, M, L, N, generator=g) < 0.6).float()
    mask[0, 1, :, 2] = 0.0          # an empty patch
    X = X * mask
    tt = tt * mask
    tp = torch.sort(torch.rand(B, Lp, generator=g), dim=1).values
    # TE + TTCN in isolation
    Xf = X.permute(0, 3, 1, 2).reshape(-1, L, 1)
    tf = tt.permute(0, 3, 1, 2).reshape(-1, L, 1)
    mf = mask.permute(0, 3, 1, 2).reshape(-1, L, 1)
    for p in m.parameters():
        p.grad = None
    te = m.LearnableTE(tf)
    h = m.TTCN(torch.cat([Xf, te], -1), mf)
    up = torch.randn(h.shape, generator=g)
    (h * up).sum().backward()
    arrs = dict(X=_np(X), tt=_np(tt), mask=_np(mask), tp=_np(tp), ttcn_out=_np(h), ttcn_upstream=_np(up),
                te=_np(te))
    for k, p in m.named_parameters():
        if p.grad is not None:
            arrs[f"g_ttcn.{k}"] = _np(p.grad)
    for p in m.parameters():
        p.grad = None
    out = m.forecasting(tp, X, tt, mask)
    up = torch.randn(out.shape, generator=g)
    (out * up).sum().backward()
    arrs["out"] = _np(out)
    arrs["upstream"] = _np(up)
    for k, v in m.state_dict().items():
        arrs[f"p.{k}"] = _np(v)
    for k, p in m.named_parameters():
        arrs[f"g.{k}"] = _np(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    save("model_tpatchgnn", **arrs)


def gen_models():
    """PatchTST / DLinear / TimesNet forecasting (eval-equivalent: dropout 0) on a short ragged history."""
    _install_shims()
    g = torch.Generator().manual_seed(51)
    B, L, Lp, K = 3, 6, 4, 3
    data = torch.randn(B, L, K, generator=g)
    mask = (torch.rand(B, L, K, generator=g) < 0.7).float()
    data = data * mask
    tp = torch.sort(torch.rand(B, L, generator=g), 1).values
    tpp = torch.sort(torch.rand(B, Lp, generator=g), 1).values
    base = dict(input_len=8, pred_len=6, d_model=8, d_ff=16, n_heads=2, e_layers=1, dropout=0.0, factor=5,
                activation="gelu", enc_in=K, c_out=K, batch_size=4, device="cpu", moving_avg=5, top_k=2, num_kernels=2,
                embed="fixed", freq="h")
    for name in ["PatchTST", "DLinear", "TimesNet"]:
        mod = importlib.import_module(f"models.{name}")
        torch.manual_seed(53)
        m = getattr(mod, name)(types.SimpleNamespace(**base))
        m.train()
        out = m.forecasting(tpp, data.clone(), tp, mask)
        up = torch.randn(out.shape, generator=g)
        (out * up).sum().backward()
        arrs = dict(data=_np(data), mask=_np(mask), tp=_np(tp), tpp=_np(tpp), out=_np(out), upstream=_np(up))
        for k, v in m.state_dict().items():
            arrs[f"p.{k}"] = _np(v)
        for k, p_ in m.named_parameters():
            arrs[f"g.{k}"] = _np(p_.grad) if p_.grad is not None else np.zeros(tuple(p_.shape), np.float32)
        save(f"model_{name.lower()}", **arrs)


def gen_layers_big():
    """PatchTST-size layers (d_model 512, 2 heads, d_ff 2048, 10 patches, 48 rows; patch_len 18, stride 9 on 96-step
    interleaved rows) and TimeLLM's ReprogrammingLayer, weights from tests/golden/seeded.py.  Stored: the first 8 rows of
    the output and of the input gradient, and a fingerprint (4 random projections + norm) of every parameter gradient."""
    _install_shims()
    sys.path.insert(0, OUT)
    import seeded
    SA = importlib.import_module("layers.SelfAttention_Family")
    EM = importlib.import_module("layers.Embed")
    TE = importlib.import_module("layers.Transformer_EncDec")
    R, P, D, H, DFF = 48, 10, 512, 2, 2048

    def run(mod, x, tag, seed, post=lambda o: o):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items() if v.dtype.is_floating_point and "pe" not in k.split(".")[-1]}
        sd = seeded.state_like(shapes, seed)
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        xx = torch.from_numpy(x).clone().requires_grad_(True)
        o = post(mod(xx))
        up = torch.from_numpy(seeded.rand(tuple(o.shape), seed + 1000))
        (o * up).sum().backward()
        arrs = dict(out8=_np(o)[:8], out_norm=np.float64(np.linalg.norm(_np(o).astype(np.float64))),
                    gx8=_np(xx.grad)[:8], gx_norm=np.float64(np.linalg.norm(_np(xx.grad).astype(np.float64))))
        for i, (k, p_) in enumerate(sorted(mod.named_parameters())):
            arrs["probe." + k] = seeded.probes(_np(p_.grad), seed + 2000 + i)
        save(tag, **arrs)

    x = seeded.rand((R, P, D), 501)
    al = SA.AttentionLayer(SA.FullAttention(False, attention_dropout=0.0), D, H)
    run(_Wrap3(al), x, "layer_big_attention_layer", 510)
    el = TE.EncoderLayer(SA.AttentionLayer(SA.FullAttention(False, attention_dropout=0.0), D, H), D, DFF, dropout=0.0, activation="gelu")
    run(el, x, "layer_big_encoder_layer", 520, post=lambda o: o[0])
    pe = EM.PatchEmbedding(D, 18, 9, 9, 0.0)
    xs = seeded.rand((8, 6, 96), 530)
    run(pe, xs, "layer_big_patch_embedding", 531, post=lambda o: o[0])
    TL = importlib.import_module("models.TimeLLM")
    rl = TL.ReprogrammingLayer(16, 8, d_llm=768, attention_dropout=0.0)
    src = torch.from_numpy(seeded.rand((1000, 768), 541))
    run(_WrapR(rl, src), seeded.rand((48, 5, 16), 540), "layer_big_reprogramming", 542)


class _Wrap3(torch.nn.Module):
    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, x):
        return self.m(x, x, x, None)[0]

    def state_dict(self, *a, **k):
        return self.m.state_dict(*a, **k)

    def load_state_dict(self, sd, strict=True):
        return self.m.load_state_dict(sd, strict)

    def named_parameters(self, *a, **k):
        return self.m.named_parameters(*a, **k)


class _WrapR(_Wrap3):
    def __init__(self, m, src):
        super().__init__(m)
        self.src = src

    def forward(self, x):
        return self.m(x, self.src, self.src)


class _ByteTok:
    """the offline tokenizer stand-in shared with the mirror (models/TimeLLM.py `_ByteTokenizer`): one token per byte,
    right-padded with id 0"""
    eos_token = "<eos>"
    pad_token = "<eos>"

    def __call__(self, prompts, return_tensors="pt", padding=True, truncation=True, max_length=512):
        ids = [list(q.encode("utf-8"))[:max_length] for q in prompts]
        n = max(len(i) for i in ids)
        t = torch.zeros((len(ids), n), dtype=torch.long)
        for r, i in enumerate(ids):
            t[r, :len(i)] = torch.tensor(i, dtype=torch.long)
        return type("Enc", (), {"input_ids": t})()

    def add_special_tokens(self, _):
        pass


TIMELLM_CFG = dict(input_len=16, pred_len=8, use_norm=True, d_ff=32, ts_vocab_size=20, input_token_len=8, stride=4,
                   domain_des="synthetic", top_k=3, C=3, llm_model_timellm="GPT2", llm_layers_timellm=2, dropout=0.0, d_model=16,
                   n_heads=2, batch_size=4, device="cpu")
# a small vocabulary (the byte tokenizer only emits ids < 256) and NO dropout inside the LLM body: the reference leaves the frozen
# GPT-2 in train mode (its own dropout draws from torch's CPU generator, which no GPU run can reproduce)
TIMELLM_GPT2 = dict(vocab_size=320, n_positions=512, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)


def gen_timellm():
    """The reference TimeLLM wrapper end to end (models/TimeLLM.py:167-278: stats -> prompt -> tokens -> two patch embeddings ->
    reprogramming -> LLM -> hidden[:, -total:, :d_ff] -> head -> de-normalise), with `_get_model_and_tokenizer` (:128-159, needs
    the hub) replaced by a random-init 2-layer GPT-2 + the byte tokenizer.  Weights of every tensor come from seeded.state_like:
    the fixture stores inputs, the output and gradient fingerprints only."""
    _install_shims()
    sys.path.insert(0, OUT)
    import seeded
    from transformers import GPT2Config, GPT2Model
    mod = importlib.import_module("models.TimeLLM")

    def fake(self, model_name, layers):
        self.llm_model = GPT2Model(GPT2Config(n_layer=layers, **TIMELLM_GPT2))
        self.tokenizer = _ByteTok()
    mod.TimeLLM._get_model_and_tokenizer = fake
    torch.manual_seed(61)
    m = mod.TimeLLM(types.SimpleNamespace(**TIMELLM_CFG))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if v.dtype.is_floating_point}
    sd = {k: torch.from_numpy(v) for k, v in seeded.state_like(shapes, 6100).items()}
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing
    m.word_embeddings = m.llm_model.get_input_embeddings().weight
    m.train()
    m.reprogramming_layer.dropout.p = 0.0       # (constructed with its default attention_dropout = 0.1 whatever configs.dropout says, :105)
    g = torch.Generator().manual_seed(62)
    B, L, Lp, K = 3, 12, 5, 3
    data = torch.randn(B, L, K, generator=g)
    mask = (torch.rand(B, L, K, generator=g) < 0.8).float()
    data = data * mask
    tp = torch.sort(torch.rand(B, L, generator=g), 1).values
    tpp = torch.sort(torch.rand(B, Lp, generator=g), 1).values
    seen = []
    real_prompt = m._get_prompt
    m._get_prompt = lambda x: seen.append(real_prompt(x)) or seen[-1]       # record the prompt strings of this batch
    out = m.forecasting(tpp, data.clone(), tp, mask)
    up = torch.randn(out.shape, generator=g)
    (out * up).sum().backward()
    arrs = dict(data=_np(data), mask=_np(mask), tp=_np(tp), tpp=_np(tpp), out=_np(out), upstream=_np(up),
                keys=np.array(sorted(shapes)), prompt0=np.array(real_prompt(torch.zeros(1, 16, 3))[0]), prompts=np.array(seen[0]))
    for i, (k, p_) in enumerate(sorted(m.named_parameters())):
        if p_.requires_grad:
            arrs[f"gp.{k}"] = seeded.probes(_np(p_.grad), 6200 + i)
    save("model_timellm", **arrs)


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["fusion", "loss", "layers", "layers_big", "tpatchgnn", "models"]
    if "layers_big" in which:
        gen_layers_big()
    if "fusion" in which:
        gen_fusion(_ref_modules())
    if "loss" in which:
        gen_loss()
    if "layers" in which:
        gen_layers()
    if "tpatchgnn" in which:
        gen_tpatchgnn()
    if "models" in which:
        gen_models()
    if "timellm" in which:
        gen_timellm()
