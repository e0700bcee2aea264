"""CPU restatement of the reference's attention / embedding layers -- TEST INFRASTRUCTURE (only tests/ may import it).

Functional plain-torch versions over parameter dicts with the reference's state_dict keys, each citing the lines it
follows; pinned against fixtures generated from the real reference (tests/test_oracle_golden.py: the toy-size
layer_*.npz and the PatchTST-size layer_big_*.npz).  Any device, fp32.
"""
import math

import torch
import torch.nn.functional as F


def full_attention(q, k, v, scale=None, causal=False):
    """layers/SelfAttention_Family.py:50-77 (dropout 0): q (B,L,H,E), k (B,S,H,E), v (B,S,H,D) -> (B,L,H,D)"""
    B, L, H, E = q.shape
    scale = scale or 1.0 / math.sqrt(E)
    scores = torch.einsum("blhe,bshe->bhls", q, k)
    if causal:
        mask = torch.triu(torch.ones(L, k.shape[1], dtype=torch.bool, device=q.device), diagonal=1)
        scores = scores.masked_fill(mask, float("-inf"))
    A = torch.softmax(scale * scores, dim=-1)
    return torch.einsum("bhls,bshd->blhd", A, v).contiguous()


def attention_layer(p, prefix, xq, xk, xv, H):
    """layers/SelfAttention_Family.py:181-215: three biased projections -> heads -> inner attention -> out projection"""
    B, L, _ = xq.shape
    S = xk.shape[1]
    q = F.linear(xq, p[prefix + "query_projection.weight"], p[prefix + "query_projection.bias"]).view(B, L, H, -1)
    k = F.linear(xk, p[prefix + "key_projection.weight"], p[prefix + "key_projection.bias"]).view(B, S, H, -1)
    v = F.linear(xv, p[prefix + "value_projection.weight"], p[prefix + "value_projection.bias"]).view(B, S, H, -1)
    out = full_attention(q, k, v).view(B, L, -1)
    return F.linear(out, p[prefix + "out_projection.weight"], p[prefix + "out_projection.bias"])


def encoder_layer(p, prefix, x, H, activation="gelu"):
    """layers/Transformer_EncDec.py:27-51 (dropout 0): post-LN block, the 1x1 convolutions are the FFN's two linear maps"""
    act = F.relu if activation == "relu" else F.gelu
    x = x + attention_layer(p, prefix + "attention.", x, x, x, H)
    d = x.shape[-1]
    y = x = F.layer_norm(x, (d,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"])
    y = act(F.linear(y, p[prefix + "conv1.weight"].squeeze(-1), p[prefix + "conv1.bias"]))
    y = F.linear(y, p[prefix + "conv2.weight"].squeeze(-1), p[prefix + "conv2.bias"])
    return F.layer_norm(x + y, (d,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"])


def encoder(p, x, H, n_layers, activation="gelu"):
    """layers/Transformer_EncDec.py:54-80 without conv layers, final LayerNorm `norm.*` when present"""
    for i in range(n_layers):
        x = encoder_layer(p, f"attn_layers.{i}.", x, H, activation)
    if "norm.weight" in p:
        x = F.layer_norm(x, (x.shape[-1],), p["norm.weight"], p["norm.bias"])
    return x


def sinusoid(n, d_model, device=None):
    """layers/Embed.py:8-26"""
    w = torch.zeros(n, d_model)
    pos = torch.arange(0, n).float().unsqueeze(1)
    div = (torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model)).exp()
    w[:, 0::2] = torch.sin(pos * div)
    w[:, 1::2] = torch.cos(pos * div)
    return w.to(device) if device is not None else w


def patch_embedding(w, x, patch_len, stride, pad):
    """layers/Embed.py:165-190 (dropout 0): x (B, n_vars, L) -> (B*n_vars, P, d_model)"""
    x = F.pad(x, (0, pad), mode="replicate").unfold(dimension=-1, size=patch_len, step=stride)
    x = x.reshape(x.shape[0] * x.shape[1], x.shape[2], x.shape[3])
    return F.linear(x, w) + sinusoid(x.shape[1], w.shape[0], x.device).unsqueeze(0)


def data_embedding(w, x):
    """layers/Embed.py:29-42,109-126 without time marks (dropout 0): x (B, L, c_in), w (d_model, c_in, 3) circular conv"""
    y = F.conv1d(F.pad(x.permute(0, 2, 1), (1, 1), mode="circular"), w).transpose(1, 2)
    return y + sinusoid(x.shape[1], w.shape[0], x.device).unsqueeze(0)


def reprogramming_layer(p, target, source, value, H):
    """models/TimeLLM.py:32-61 (dropout 0): target (B,L,d_model), source / value (S,d_llm) -> (B,L,d_llm)"""
    B, L, _ = target.shape
    S = source.shape[0]
    q = F.linear(target, p["query_projection.weight"], p["query_projection.bias"]).view(B, L, H, -1)
    k = F.linear(source, p["key_projection.weight"], p["key_projection.bias"]).view(S, H, -1)
    v = F.linear(value, p["value_projection.weight"], p["value_projection.bias"]).view(S, H, -1)
    scale = 1.0 / math.sqrt(source.shape[-1] // H)      # the reference scales by d_llm / H, not by the head width (:51,55)
    A = torch.softmax(scale * torch.einsum("blhe,she->bhls", q, k), dim=-1)
    out = torch.einsum("bhls,she->blhe", A, v).reshape(B, L, -1)
    return F.linear(out, p["out_projection.weight"], p["out_projection.bias"])
