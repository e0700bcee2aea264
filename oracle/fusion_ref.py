"""CPU ORACLE -- test infrastructure, not product code.

A plain-PyTorch (CPU, fp32) restatement of the reference's multimodal-fusion hot
path, written functionally over a `params` dict that uses the reference's
state_dict keys.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this package; the shipped modules under
`imm-tsf_amd/` never do (they fail loudly when the HIP library is missing).

Parity pin: every function here is checked against golden vectors captured from
the real reference (tests/golden/make_golden.py -> tests/golden/*.npz) by
tests/test_oracle_golden.py.

What each function follows (file:line into the reference):
  note_mask / ragged index  fusions/TTF_T2V_XAttn.py:107,124,146 ; fusions/TTF_RecAvg.py:69,110
  time2vec                  fusions/TTF_T2V_XAttn.py:7-24
  mha                       torch.nn.functional.multi_head_attention_forward semantics at the call
                            sites fusions/TTF_T2V_XAttn.py:161-166 and fusions/MMF_XAttn_Add.py:76
  ttf_t2v_xattn             fusions/TTF_T2V_XAttn.py:93-184
  ttf_recavg                fusions/TTF_RecAvg.py:54-112
  mmf_xattn_add             fusions/MMF_XAttn_Add.py:56-103
  mmf_gr_add                fusions/MMF_GR_Add.py:31-61 (nn.GRU gate order r,z,n)
  fusion_forward            fusions/FusionModel.py:98-113
  masked_mse                lib/evaluation.py:17-62 (func="MSE", reduce="mean")

Dropout: the reference draws masks from torch's CPU RNG, which no GPU kernel can
reproduce.  Every function therefore takes an optional `drop` dict of explicit
KEEP masks (1 = keep) plus the rate `p`; with drop=None it is the dropout-free
(eval / p=0) computation.  The HIP path can export the masks it drew so the two
sides are compared on identical masks.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- ragged index (bit-exact part)
def note_mask_of(notes: Tensor) -> Tensor:
    """(B,N,d_m) -> bool (B,N): a note exists iff sum|embedding| > 0 (on the RAW embedding)."""
    return notes.abs().sum(dim=2) > 0


def ragged_index(notes: Tensor):
    """note_mask (B,N) bool, lengths (B,) int32, offsets (B+1,) int32, rowmap (sum N,) int32 = b*N+n."""
    mask = note_mask_of(notes)
    lengths = mask.sum(dim=1).to(torch.int32)
    offsets = torch.zeros(mask.shape[0] + 1, dtype=torch.int32)
    offsets[1:] = torch.cumsum(lengths, 0)
    rowmap = torch.nonzero(mask.reshape(-1), as_tuple=False).reshape(-1).to(torch.int32)
    return mask, lengths, offsets, rowmap


# ----------------------------------------------------------------------------- small pieces
def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


def time2vec(tau: Tensor, p: Dict[str, Tensor], prefix: str = "time2vec.") -> Tensor:
    """tau (...,) -> (..., d_tau): [w0*tau+b0 ; sin(W*tau+b)]"""
    x = tau.unsqueeze(-1)
    lin = linear(x, p[prefix + "linear.weight"], p[prefix + "linear.bias"])
    per = torch.sin(linear(x, p[prefix + "periodic.weight"], p[prefix + "periodic.bias"]))
    return torch.cat([lin, per], dim=-1)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _drop(x: Tensor, keep: Optional[Tensor], p: float) -> Tensor:
    if keep is None or p == 0.0:
        return x
    return x * keep.to(x.dtype) / (1.0 - p)


def mha(q_in: Tensor, k_in: Tensor, v_in: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor,
        H: int, key_padding_mask: Optional[Tensor] = None, keep: Optional[Tensor] = None, p: float = 0.0,
        nan_safe: bool = True) -> Tensor:
    """nn.MultiheadAttention(batch_first=True) slow path: q_in (B,L,E), k_in/v_in (B,S,E).

    key_padding_mask (B,S) bool, True = ignore.  keep: (B,H,L,S) dropout keep-mask on the weights.
    nan_safe: rows whose keys are all masked give 0 (the callers overwrite them with 0 anyway via
    torch.where; the reference's NaN there only poisons its backward)."""
    B, L, E = q_in.shape
    S = k_in.shape[1]
    hd = E // H
    wq, wk, wv = in_w[:E], in_w[E:2 * E], in_w[2 * E:]
    bq, bk, bv = in_b[:E], in_b[E:2 * E], in_b[2 * E:]
    q = linear(q_in, wq, bq).view(B, L, H, hd).transpose(1, 2)     # (B,H,L,hd)
    k = linear(k_in, wk, bk).view(B, S, H, hd).transpose(1, 2)
    v = linear(v_in, wv, bv).view(B, S, H, hd).transpose(1, 2)
    q = q * math.sqrt(1.0 / hd)
    s = q @ k.transpose(-1, -2)                                    # (B,H,L,S)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask.view(B, 1, 1, S), float("-inf"))
    if nan_safe and key_padding_mask is not None:
        dead = key_padding_mask.all(dim=1).view(B, 1, 1, 1)
        s = torch.where(dead, torch.zeros_like(s), s)
        a = torch.softmax(s, dim=-1)
        a = torch.where(dead, torch.zeros_like(a), a)
    else:
        a = torch.softmax(s, dim=-1)
    a = _drop(a, keep, p)
    o = (a @ v).transpose(1, 2).reshape(B, L, E)
    return linear(o, out_w, out_b)


# ----------------------------------------------------------------------------- TTF blocks
def _prep_t_hat(t_hat: Tensor, B: int) -> Tensor:
    if t_hat.dim() == 1:
        return t_hat.unsqueeze(0).repeat(B, 1)
    if t_hat.shape[0] != B:
        raise ValueError(f"Expected t_hat shape (B, T_f) or (T_f,), got {tuple(t_hat.shape)}")
    return t_hat


def ttf_t2v_xattn(p: Dict[str, Tensor], notes: Tensor, tau: Tensor, t_hat: Tensor, H: int,
                  drop: Optional[Dict[str, Tensor]] = None, p_drop: float = 0.0, expand_T: bool = True):
    """-> (E_txt (B,T,d), M_txt (B,1) bool).  drop keys: 'attn' (B,T,H,N), 'out' (B,T,d).

    expand_T=True materialises K/V once per forecast step exactly like the reference does
    (fusions/TTF_T2V_XAttn.py:150-159): that is the faithful CPU baseline.  expand_T=False computes
    the attention once per window and is only valid without attention dropout."""
    drop = drop or {}
    if torch.isnan(notes).any():
        raise ValueError("Input embeddings V contain NaN values.")
    mask = note_mask_of(notes)
    V = notes
    if "input_proj.weight" in p:
        V = linear(V, p["input_proj.weight"], p["input_proj.bias"])
    M_txt = mask.any(dim=1, keepdim=True)
    B, N, d = V.shape
    t_hat = _prep_t_hat(t_hat, B)
    T = t_hat.shape[1]
    feat = time2vec(tau, p)
    KV = linear(torch.cat([V, feat], dim=-1), p["KV_proj.weight"], p["KV_proj.bias"])      # (B,N,d)
    Qp = p["Q_param"].reshape(1, 1, d)
    pad = ~mask
    keep_a = drop.get("attn")
    if expand_T:
        Qf = Qp.expand(B, T, d).reshape(B * T, 1, d)
        KVf = KV.unsqueeze(1).expand(-1, T, -1, -1).reshape(B * T, N, d)
        padf = pad.unsqueeze(1).expand(-1, T, -1).reshape(B * T, N)
        ka = None if keep_a is None else keep_a.reshape(B * T, H, 1, N)
        att = mha(Qf, KVf, KVf, p["attn.in_proj_weight"], p["attn.in_proj_bias"],
                  p["attn.out_proj.weight"], p["attn.out_proj.bias"], H, padf, ka, p_drop)
        E_attn = att.reshape(B, T, d)
    else:
        assert keep_a is None or p_drop == 0.0
        att = mha(Qp.expand(B, 1, d), KV, KV, p["attn.in_proj_weight"], p["attn.in_proj_bias"],
                  p["attn.out_proj.weight"], p["attn.out_proj.bias"], H, pad)
        E_attn = att.expand(B, T, d)
    E_attn = torch.where(M_txt.view(B, 1, 1), E_attn, torch.zeros_like(E_attn))
    x = layer_norm(E_attn + Qp, p["layer_norm.weight"], p["layer_norm.bias"])
    x = _drop(x, drop.get("out"), p_drop)
    return linear(x, p["proj_out.weight"], p["proj_out.bias"]), M_txt


def ttf_recavg(p: Dict[str, Tensor], notes: Tensor, tau: Tensor, t_hat: Tensor,
               drop: Optional[Dict[str, Tensor]] = None, p_drop: float = 0.0):
    """-> (E_txt (B,T,d), M_txt (B,1) bool).  drop keys: 'out' (B,T,d)."""
    drop = drop or {}
    if torch.isnan(notes).any():
        raise ValueError("Input embeddings V contain NaN values.")
    mask = note_mask_of(notes)
    V = notes
    if "input_proj.weight" in p:
        V = linear(V, p["input_proj.weight"], p["input_proj.bias"])
    B = V.shape[0]
    t_hat = _prep_t_hat(t_hat, B)
    delta = (t_hat[:, None, :] - tau[:, :, None]).clamp_min(0)           # (B,N,T)
    sigma = p["log_recency_sigma"].exp()
    w = torch.exp(-((delta / sigma) ** 2)) * mask.to(V.dtype)[:, :, None]
    wsum = torch.einsum("bnt,bnd->btd", w, V)
    denom = w.sum(dim=1).clamp_min(1e-6)
    x = layer_norm(wsum / denom.unsqueeze(-1), p["layer_norm.weight"], p["layer_norm.bias"])
    x = _drop(x, drop.get("out"), p_drop)
    return linear(x, p["proj.weight"], p["proj.bias"]), mask.any(dim=1, keepdim=True)


# ----------------------------------------------------------------------------- MMF blocks
def mmf_xattn_add(p: Dict[str, Tensor], Y_ts: Tensor, E_txt: Tensor, M_txt: Tensor, H: int, kappa: float,
                  drop: Optional[Dict[str, Tensor]] = None, p_drop: float = 0.0) -> Tensor:
    """drop keys: 'attn' (B,H,T,T), 'out' (B,T,C)."""
    drop = drop or {}
    B, T, C = Y_ts.shape
    Q = linear(Y_ts, p["proj_q.weight"])
    K = linear(E_txt, p["proj_k.weight"])
    V = linear(E_txt, p["proj_v.weight"])
    Mb = M_txt.view(B, 1)
    att = mha(Q, K, V, p["attn.in_proj_weight"], p["attn.in_proj_bias"], p["attn.out_proj.weight"],
              p["attn.out_proj.bias"], H, (~Mb).expand(-1, T), drop.get("attn"), p_drop)
    att = torch.where(Mb.view(B, 1, 1), att, torch.zeros_like(att))
    dy = linear(att, p["residual_head.weight"], p["residual_head.bias"])
    dn = layer_norm(dy, p["layer_norm.weight"], p["layer_norm.bias"])
    dn = _drop(dn, drop.get("out"), p_drop)
    dn = torch.where(Mb.view(B, 1, 1), dn, torch.zeros_like(dn))
    return (Y_ts + kappa * dn) / (1.0 + kappa)


def gru_seq(x: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor) -> Tensor:
    """nn.GRU(batch_first=True), one layer, h0 = 0.  x (B,T,I) -> (B,T,Hd); gate order r,z,n."""
    B, T, _ = x.shape
    Hd = w_hh.shape[1]
    gi = linear(x, w_ih, b_ih)                                   # (B,T,3Hd) input side hoisted
    h = x.new_zeros(B, Hd)
    outs = []
    for t in range(T):
        gh = linear(h, w_hh, b_hh)
        i_r, i_z, i_n = gi[:, t].chunk(3, dim=-1)
        h_r, h_z, h_n = gh.chunk(3, dim=-1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1.0 - z) * n + z * h
        outs.append(h)
    return torch.stack(outs, dim=1)


def mmf_gr_add(p: Dict[str, Tensor], Y_ts: Tensor, E_txt: Tensor, M_txt: Tensor,
               drop: Optional[Dict[str, Tensor]] = None, p_drop: float = 0.0) -> Tensor:
    """drop keys: 'out' (B,T,C)."""
    drop = drop or {}
    B, T, C = Y_ts.shape
    x = torch.cat([Y_ts, E_txt], dim=-1)
    h = gru_seq(x, p["gru.weight_ih_l0"], p["gru.weight_hh_l0"], p["gru.bias_ih_l0"], p["gru.bias_hh_l0"])
    dy = linear(h, p["residual_head.weight"], p["residual_head.bias"])
    dn = layer_norm(dy, p["layer_norm.weight"], p["layer_norm.bias"])
    dn = _drop(dn, drop.get("out"), p_drop)
    g = torch.sigmoid(linear(x, p["gate_net.weight"], p["gate_net.bias"]))
    g = torch.where(M_txt.view(B, 1, 1), g, torch.ones_like(g))
    return g * Y_ts + (1 - g) * (Y_ts + dn)


# ----------------------------------------------------------------------------- composite + loss
def _sub(p: Dict[str, Tensor], prefix: str) -> Dict[str, Tensor]:
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def fusion_forward(ttf: str, mmf: str, p: Dict[str, Tensor], notes: Tensor, tau: Tensor, t_hat: Tensor,
                   Y_ts: Tensor, H: int = 1, kappa: float = 0.5, drop: Optional[Dict[str, Dict]] = None,
                   p_drop: float = 0.0, expand_T: bool = True) -> Tensor:
    """FusionModel.forward: NaN guard -> ttf -> NaN guard -> mmf -> NaN guard.  `p` has ttf./mmf. prefixes."""
    drop = drop or {}
    if torch.isnan(Y_ts).any():
        raise ValueError("Y_ts contains NaN values.")
    if ttf == "TTF_T2V_XAttn":
        E, M = ttf_t2v_xattn(_sub(p, "ttf."), notes, tau, t_hat, H, drop.get("ttf"), p_drop, expand_T)
    elif ttf == "TTF_RecAvg":
        E, M = ttf_recavg(_sub(p, "ttf."), notes, tau, t_hat, drop.get("ttf"), p_drop)
    else:
        raise KeyError(ttf)
    if torch.isnan(E).any():
        raise ValueError("E_txt contains NaN values.")
    if mmf == "MMF_XAttn_Add":
        Y = mmf_xattn_add(_sub(p, "mmf."), Y_ts, E, M, H, kappa, drop.get("mmf"), p_drop)
    elif mmf == "MMF_GR_Add":
        Y = mmf_gr_add(_sub(p, "mmf."), Y_ts, E, M, drop.get("mmf"), p_drop)
    else:
        raise KeyError(mmf)
    if torch.isnan(Y).any():
        raise ValueError("Y_out contains NaN values.")
    return Y


def masked_err_sums(truth: Tensor, pred: Tensor, mask: Tensor, func: str = "MSE"):
    """per-variable (sum of masked error, count of observations): compute_error(..., reduce="sum")."""
    C = pred.shape[-1]
    d = truth - pred
    err = (d * d if func == "MSE" else d.abs()) * mask
    return err.reshape(-1, C).sum(0), mask.reshape(-1, C).sum(0)


def masked_mse(truth: Tensor, pred: Tensor, mask: Tensor, func: str = "MSE") -> Tensor:
    """compute_error(truth, pred, mask, func, "mean"): mean over observed variables of per-variable means."""
    es, mc = masked_err_sums(truth, pred, mask, func)
    return (es / (mc + 1e-8)).sum() / torch.count_nonzero(mc)


def params_from_npz(z, prefix: str = "p.") -> Dict[str, Tensor]:
    return {k[len(prefix):]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(prefix)}
