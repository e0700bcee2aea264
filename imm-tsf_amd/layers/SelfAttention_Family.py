"""FullAttention / AttentionLayer on MI355X (reference layers/SelfAttention_Family.py:50-77, 181-215).

scores = einsum("blhe,bshe->bhls") and V = einsum("bhls,bshd->blhd") run as batched MFMA GEMMs over (batch, head)
directly on the (B,L,H,E) layout (no permutes/copies); softmax(scale*scores) + attention dropout is one fused row
kernel whose Philox mask is regenerated in backward.  `mask_flag=True` without an explicit mask is the causal
TriangularCausalMask (utils/masking.py); explicit attn_mask tensors are not on any configured path.
"""
from math import sqrt

import torch
import torch.nn as nn

from immtsf import config
from immtsf.ops import full_attention, linear, linear_multi


class FullAttention(nn.Module):
    _next_site = 0

    def __init__(self, mask_flag=True, factor=5, scale=None, attention_dropout=0.1, output_attention=False):
        super().__init__()
        self.scale = scale
        self.mask_flag = mask_flag
        self.output_attention = output_attention
        self.dropout = nn.Dropout(attention_dropout)
        self.p_drop = float(attention_dropout)
        self.site = 16 + (FullAttention._next_site % 1024)     # a Philox subsequence per attention instance
        FullAttention._next_site += 1
        self.precision = None

    def forward(self, queries, keys, values, attn_mask, tau=None, delta=None):
        if self.output_attention:
            raise NotImplementedError("output_attention=True is not provided by the fused path")
        if self.mask_flag and attn_mask is not None:
            raise NotImplementedError("explicit attn_mask tensors are not supported; mask_flag=True means causal")
        E = queries.shape[-1]
        scale = self.scale or 1.0 / sqrt(E)
        training = self.training and self.p_drop > 0.0
        seed = config.next_seed() if training else 0
        out = full_attention(queries, keys, values, scale, self.p_drop, training, seed, self.site, self.mask_flag,
                             self.precision)
        return out, None


class AttentionLayer(nn.Module):
    def __init__(self, attention, d_model, n_heads, d_keys=None, d_values=None):
        super().__init__()
        d_keys = d_keys or (d_model // n_heads)
        d_values = d_values or (d_model // n_heads)
        self.inner_attention = attention
        self.query_projection = nn.Linear(d_model, d_keys * n_heads)
        self.key_projection = nn.Linear(d_model, d_keys * n_heads)
        self.value_projection = nn.Linear(d_model, d_values * n_heads)
        self.out_projection = nn.Linear(d_values * n_heads, d_model)
        self.n_heads = n_heads

    def forward(self, queries, keys, values, attn_mask, tau=None, delta=None):
        B, L, _ = queries.shape
        S, H = keys.shape[1], self.n_heads
        prec = getattr(self.inner_attention, "precision", None)
        if queries is keys and keys is values:      # self-attention: the three projections share their input -- one launch (bf16 mode)
            q, k, v = linear_multi(queries, [self.query_projection.weight, self.key_projection.weight, self.value_projection.weight],
                                   [self.query_projection.bias, self.key_projection.bias, self.value_projection.bias], prec)
            q, k, v = q.view(B, L, H, -1), k.view(B, S, H, -1), v.view(B, S, H, -1)
        else:
            q = linear(queries, self.query_projection.weight, self.query_projection.bias, prec).view(B, L, H, -1)
            k = linear(keys, self.key_projection.weight, self.key_projection.bias, prec).view(B, S, H, -1)
            v = linear(values, self.value_projection.weight, self.value_projection.bias, prec).view(B, S, H, -1)
        out, attn = self.inner_attention(q, k, v, attn_mask, tau=tau, delta=delta)
        return linear(out.reshape(B, L, -1), self.out_projection.weight, self.out_projection.bias, prec), attn


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
