"""Embeddings used by PatchTST / TimesNet / TimeLLM (reference layers/Embed.py:8-42, 109-126, 165-190).

PatchEmbedding (replication pad -> unfold -> Linear(patch_len -> d_model) -> + positional table -> dropout) and
DataEmbedding without time marks (circular 3-tap token convolution -> + positional table -> dropout) each run as ONE HIP
kernel per direction (immtsf.ops.patch_embed / token_embed, csrc/embed.hip): no padded copy, no unfolded tensor, no
separate add / dropout passes.  Shapes outside the kernel's limits (patch_len or 3 c_in > 64, d_model > 512) and
DataEmbedding with time marks use the HIP GEMM on gathered taps.  Fixed sinusoid tables are buffers with the
reference's names so state_dict keys match."""
import math

import torch
import torch.nn as nn

from immtsf.ops import embed_supported, linear, patch_embed, token_embed


def _sinusoid(n, d_model):
    w = torch.zeros(n, d_model).float()
    pos = torch.arange(0, n).float().unsqueeze(1)
    div = (torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model)).exp()
    w[:, 0::2] = torch.sin(pos * div)
    w[:, 1::2] = torch.cos(pos * div)
    return w


class PositionalEmbedding(nn.Module):
    def __init__(self, d_model, max_len=5000):
        super().__init__()
        self.register_buffer("pe", _sinusoid(max_len, d_model).unsqueeze(0))

    def forward(self, x):
        return self.pe[:, :x.size(1)]


class TokenEmbedding(nn.Module):
    def __init__(self, c_in, d_model):
        super().__init__()
        self.tokenConv = nn.Conv1d(in_channels=c_in, out_channels=d_model, kernel_size=3, padding=1,
                                   padding_mode="circular", bias=False)
        nn.init.kaiming_normal_(self.tokenConv.weight, mode="fan_in", nonlinearity="leaky_relu")

    def forward(self, x):                       # x (B, L, c_in) -> (B, L, d_model)
        taps = torch.cat([torch.roll(x, 1, dims=1), x, torch.roll(x, -1, dims=1)], dim=-1)      # [x(l-1); x(l); x(l+1)]
        w = self.tokenConv.weight.permute(0, 2, 1).reshape(self.tokenConv.out_channels, -1)    # (d_model, 3*c_in)
        return linear(taps, w, None)


class FixedEmbedding(nn.Module):
    def __init__(self, c_in, d_model):
        super().__init__()
        self.emb = nn.Embedding(c_in, d_model)
        self.emb.weight = nn.Parameter(_sinusoid(c_in, d_model), requires_grad=False)

    def forward(self, x):
        return self.emb(x).detach()


class TemporalEmbedding(nn.Module):
    def __init__(self, d_model, embed_type="fixed", freq="h"):
        super().__init__()
        Embed = FixedEmbedding if embed_type == "fixed" else nn.Embedding
        if freq == "t":
            self.minute_embed = Embed(4, d_model)
        self.hour_embed = Embed(24, d_model)
        self.weekday_embed = Embed(7, d_model)
        self.day_embed = Embed(32, d_model)
        self.month_embed = Embed(13, d_model)

    def forward(self, x):
        x = x.long()
        minute = self.minute_embed(x[:, :, 4]) if hasattr(self, "minute_embed") else 0.0
        return self.hour_embed(x[:, :, 3]) + self.weekday_embed(x[:, :, 2]) + self.day_embed(x[:, :, 1]) + \
            self.month_embed(x[:, :, 0]) + minute


class TimeFeatureEmbedding(nn.Module):
    def __init__(self, d_model, embed_type="timeF", freq="h"):
        super().__init__()
        d_inp = {"h": 4, "t": 5, "s": 6, "m": 1, "a": 1, "w": 2, "d": 3, "b": 3}[freq]
        self.embed = nn.Linear(d_inp, d_model, bias=False)

    def forward(self, x):
        return self.embed(x)


class DataEmbedding(nn.Module):
    def __init__(self, c_in, d_model, embed_type="fixed", freq="h", dropout=0.1):
        super().__init__()
        self.value_embedding = TokenEmbedding(c_in=c_in, d_model=d_model)
        self.position_embedding = PositionalEmbedding(d_model=d_model)
        self.temporal_embedding = TemporalEmbedding(d_model=d_model, embed_type=embed_type, freq=freq) \
            if embed_type != "timeF" else TimeFeatureEmbedding(d_model=d_model, embed_type=embed_type, freq=freq)
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, x, x_mark=None):
        conv = self.value_embedding.tokenConv
        if x_mark is None and x.is_cuda and embed_supported(3 * conv.in_channels, conv.out_channels):
            return token_embed(x, conv.weight, self.position_embedding.pe, self.dropout.p, self.training)
        y = self.value_embedding(x) + self.position_embedding(x)
        if x_mark is not None:
            y = y + self.temporal_embedding(x_mark)
        return self.dropout(y)


class PatchEmbedding(nn.Module):
    def __init__(self, d_model, patch_len, stride, padding, dropout):
        super().__init__()
        self.patch_len = patch_len
        self.stride = stride
        self.padding_patch_layer = nn.ReplicationPad1d((0, padding))
        self.value_embedding = nn.Linear(patch_len, d_model, bias=False)
        self.position_embedding = PositionalEmbedding(d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):                       # x (B, n_vars, L) -> ((B*n_vars, P, d_model), n_vars)
        n_vars = x.shape[1]
        W = self.value_embedding.weight
        if x.is_cuda and embed_supported(self.patch_len, W.shape[0]):
            pad = self.padding_patch_layer.padding[1]
            return patch_embed(x, W, self.position_embedding.pe, self.patch_len, self.stride, pad, self.dropout.p, self.training), n_vars
        x = self.padding_patch_layer(x).unfold(dimension=-1, size=self.patch_len, step=self.stride)
        x = torch.reshape(x, (x.shape[0] * x.shape[1], x.shape[2], x.shape[3]))
        y = linear(x, self.value_embedding.weight, None)
        return self.dropout(y + self.position_embedding(x)), n_vars


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
