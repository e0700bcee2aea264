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

    # ---- split form (csrc/gr_train.hip): the text columns of the GRU's input map and of the gate net ahead of the backbone, the rest --
    # with the loss and the backward of both -- as one launch between the backbone's forward and backward
    def _split(self, T: int, d: int) -> bool:
        if not config.gr_split:
            return False
        from immtsf.ops import gr_split_pw
        return gr_split_pw(T, self.C, d, self.hidden_dim) > 0

    def project_kv(self, E_txt, with_fold: bool = True):
        """the text-only half of the block (the name FusionModel.text_side / the step engines look for): (P, None) with P (B, T, pw) =
        E_txt [W_ih[:, C:] ; W_g[:, C:]]^T + [b_ih ; b_g] in the split form, (None, None) where it does not apply"""
        if not self._split(E_txt.shape[1], E_txt.shape[2]):
            return None, None
        from immtsf.ops import MMFGRPFn
        return MMFGRPFn.apply(f32(E_txt), self.C, self.hidden_dim, resolve_precision(self), *self._params()), None

    def forward_loss(self, Y_ts, E_txt, M_txt, truth, mask, global_cnt, kv=None):
        """masked_mse(forward(Y_ts, E_txt, M_txt), truth, mask, global_cnt=global_cnt) for a training step: in the split form the Y half,
        the loss and the backward of both are one launch (immtsf.ops.MMFGRQLossFn); otherwise the two calls."""
        from immtsf.ops import MMFGRQLossFn, masked_mse
        P = kv[0] if kv is not None else None
        if P is None and torch.is_grad_enabled() and global_cnt is not None and self._split(E_txt.shape[1], E_txt.shape[2]):
            P = self.project_kv(E_txt)[0]
        if P is None or not torch.is_grad_enabled() or global_cnt is None:
            return masked_mse(self.forward(Y_ts, E_txt, M_txt), truth, mask, None, global_cnt)
        B = Y_ts.shape[0]
        M_u8 = M_txt.reshape(B).to(torch.bool).view(torch.uint8)
        training = self.training and self.p_drop > 0.0
        self.last_seed = config.next_seed() if training else 0
        return MMFGRQLossFn.apply(f32(Y_ts), P, M_u8, f32(truth), f32(mask), global_cnt, self.hidden_dim, self.p_drop, training,
                                  resolve_precision(self), self.last_seed, *self._params())

    def forward(self, Y_ts, E_txt, M_txt, kv=None):
        """kv: the result of project_kv(E_txt) when the caller computed it ahead of time (used where no gradient is wanted: the split
        form's output-only launch; with autograd on, the block as written runs -- training goes through forward_loss)"""
        B = Y_ts.shape[0]
        M_u8 = M_txt.reshape(B).to(torch.bool).view(torch.uint8)
        if kv is not None and kv[0] is not None and not torch.is_grad_enabled() and not (self.training and self.p_drop > 0.0):
            from immtsf.ops import MMFGRQFn
            return MMFGRQFn.apply(f32(Y_ts), kv[0], M_u8, self.hidden_dim, resolve_precision(self), *self._params())
        training = self.training and self.p_drop > 0.0
        self.last_seed = config.next_seed() if training else 0
        return MMFGRAddFn.apply(f32(Y_ts), f32(E_txt), M_u8, self.hidden_dim, self.p_drop, training,
                                resolve_precision(self), self.last_seed, *self._params())


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
