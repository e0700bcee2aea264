"""MMF_GR_Add on MI355X: GRU over [Y_ts ; E_txt] -> residual, LayerNorm(C), dropout, sigmoid gate, gated add.

Interface/state_dict follow the reference (fusions/MMF_GR_Add.py:9-61); computed by
`immtsf_mmf_gr_add_forward/backward`: the input-side GRU product and the gate logits for all (b,t) are two MFMA
GEMMs, the hidden-state recurrence runs one workgroup per window, the tail is one fused row kernel.
"""
import torch
import torch.nn as nn

from fusions._common import f32, resolve_precision
from immtsf import config
from immtsf.ops import MMFGRAddFn


class MMF_GR_Add(nn.Module):
    def __init__(self, d_txt: int, C: int, hidden_dim: int, dropout: float = 0.1):
        super().__init__()
        if hidden_dim != C:
            # the reference's residual_head maps hidden_dim -> C and FusionModel passes hidden_dim=C (:81-86)
            pass
        self.C = C
        self.d_txt = d_txt
        self.hidden_dim = hidden_dim
        self.p_drop = float(dropout)
        self.gru = nn.GRU(input_size=C + d_txt, hidden_size=hidden_dim, batch_first=True)
        self.residual_head = nn.Linear(hidden_dim, C)
        self.gate_net = nn.Linear(C + d_txt, C)
        self.layer_norm = nn.LayerNorm(C)
        self.dropout = nn.Dropout(dropout)
        self.precision = None
        self.last_seed = 0

    def _params(self):
        return (self.gru.weight_ih_l0, self.gru.weight_hh_l0, self.gru.bias_ih_l0, self.gru.bias_hh_l0,
                self.residual_head.weight, self.residual_head.bias, self.gate_net.weight, self.gate_net.bias,
                self.layer_norm.weight, self.layer_norm.bias)

    def forward(self, Y_ts, E_txt, M_txt):
        B = Y_ts.shape[0]
        M_u8 = M_txt.reshape(B).to(torch.bool).view(torch.uint8)
        training = self.training and self.p_drop > 0.0
        self.last_seed = config.next_seed() if training else 0
        return MMFGRAddFn.apply(f32(Y_ts), f32(E_txt), M_u8, self.hidden_dim, self.p_drop, training,
                                resolve_precision(self), self.last_seed, *self._params())


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
