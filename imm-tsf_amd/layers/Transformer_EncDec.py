"""EncoderLayer / Encoder (reference layers/Transformer_EncDec.py:27-80): post-LN block whose 1x1 convolutions are the
two FFN GEMMs.  The two residual joints are block calls (immtsf.ops.residual_layer_norm, ffn_block: residual add and
dropout inside the LayerNorm kernels, activation -- ReLU or GELU -- and dropout in the GEMM epilogues); widths the
row kernels do not take (d_model % 4 != 0 or > 1024) and CPU tensors (which raise inside the ops) use the op-by-op form."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from immtsf.ops import SITE_LAYER_BASE, ffn_block, layer_norm, linear, residual_layer_norm, residual_layernorm_supported


def _ln(norm: nn.LayerNorm, x):
    return layer_norm(x, norm.weight, norm.bias, norm.eps)


class EncoderLayer(nn.Module):
    def __init__(self, attention, d_model, d_ff=None, dropout=0.1, activation="relu"):
        super().__init__()
        d_ff = d_ff or 4 * d_model
        self.attention = attention
        self.conv1 = nn.Conv1d(in_channels=d_model, out_channels=d_ff, kernel_size=1)
        self.conv2 = nn.Conv1d(in_channels=d_ff, out_channels=d_model, kernel_size=1)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.activation = F.relu if activation == "relu" else F.gelu
        self._act = "relu" if activation == "relu" else "gelu"

    def forward(self, x, attn_mask=None, tau=None, delta=None):
        new_x, attn = self.attention(x, x, x, attn_mask=attn_mask, tau=tau, delta=delta)
        if x.is_cuda and residual_layernorm_supported(x.shape[-1]):
            base = SITE_LAYER_BASE + 128         # every call draws its own Philox key, so the sites can be shared by all layers
            x = residual_layer_norm(x, new_x, self.norm1, self.dropout.p, self.training, base)
            return ffn_block(x, self.conv1, self.conv2, self.norm2, self._act, self.dropout.p, self.training, base + 1), attn
        x = x + self.dropout(new_x)
        y = x = _ln(self.norm1, x)
        y = self.dropout(self.activation(linear(y, self.conv1.weight.squeeze(-1), self.conv1.bias)))
        y = self.dropout(linear(y, self.conv2.weight.squeeze(-1), self.conv2.bias))
        return _ln(self.norm2, x + y), attn


class Encoder(nn.Module):
    def __init__(self, attn_layers, conv_layers=None, norm_layer=None):
        super().__init__()
        self.attn_layers = nn.ModuleList(attn_layers)
        self.conv_layers = nn.ModuleList(conv_layers) if conv_layers is not None else None
        self.norm = norm_layer

    def forward(self, x, attn_mask=None, tau=None, delta=None):
        attns = []
        if self.conv_layers is not None:
            for i, (attn_layer, conv_layer) in enumerate(zip(self.attn_layers, self.conv_layers)):
                x, attn = attn_layer(x, attn_mask=attn_mask, tau=tau, delta=delta if i == 0 else None)
                x = conv_layer(x)
                attns.append(attn)
            x, attn = self.attn_layers[-1](x, tau=tau, delta=None)
            attns.append(attn)
        else:
            for attn_layer in self.attn_layers:
                x, attn = attn_layer(x, attn_mask=attn_mask, tau=tau, delta=delta)
                attns.append(attn)
        if self.norm is not None:
            x = _ln(self.norm, x) if isinstance(self.norm, nn.LayerNorm) else self.norm(x)
        return x, attns


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
