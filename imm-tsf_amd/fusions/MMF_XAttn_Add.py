"""MMF_XAttn_Add on MI355X: cross-attention from the time-series forecast (queries) to the time-aligned text
embedding (keys/values), residual head, LayerNorm(C), dropout and a fixed-kappa convex blend.

Interface/state_dict follow the reference (fusions/MMF_XAttn_Add.py:9-103); computed by
`immtsf_mmf_xattn_{kv,q}_forward/backward` (the block as a key/value half and a query half; the back-to-back linear maps
proj_{q,k,v} -> MHA in-projection and out_proj -> residual_head run as per-step PRODUCT weights, the original parameters'
gradients follow by the chain rule; batched MFMA GEMMs for QK^T and A*V over (window, head), fused softmax+dropout rows,
fused head + LN/blend tail).  Quirk kept: a window without text
returns Y_ts/(1+kappa).
"""
import torch
import torch.nn as nn

from fusions._common import f32, resolve_precision
from immtsf import config
from immtsf.ops import MMFXAttnKVFn, MMFXAttnQFn, MMFXRankPFn, MMFXRankQFn, MMFXRankQLossFn, masked_mse, mmf_xattn_q_fold, mmf_xrank_pw


class MMF_XAttn_Add(nn.Module):
    def __init__(self, d_txt: int, C: int, d_attn: int, n_heads_fusion: int = 1, dropout: float = 0.1,
                 kappa: float = 1.0):
        super().__init__()
        if d_attn != d_txt:
            # FusionModel always passes d_attn=d_txt (fusions/FusionModel.py:88-91); the HIP block assumes it
            raise NotImplementedError("MMF_XAttn_Add on MI355X requires d_attn == d_txt")
        self.C = C
        self.d_attn = d_attn
        self.kappa = kappa
        self.n_heads = n_heads_fusion
        self.p_drop = float(dropout)
        self.proj_q = nn.Linear(C, d_attn, bias=False)
        self.proj_k = nn.Linear(d_txt, d_attn, bias=False)
        self.proj_v = nn.Linear(d_txt, d_attn, bias=False)
        self.attn = nn.MultiheadAttention(embed_dim=d_attn, num_heads=n_heads_fusion, dropout=dropout, batch_first=True)
        self.residual_head = nn.Linear(d_attn, C)
        self.layer_norm = nn.LayerNorm(C)
        self.dropout = nn.Dropout(dropout)
        self.precision = None
        self.last_seed = 0
        self._rank_ok = {}

    def _params(self):
        return (self.proj_q.weight, self.proj_k.weight, self.proj_v.weight, self.attn.in_proj_weight,
                self.attn.in_proj_bias, self.attn.out_proj.weight, self.attn.out_proj.bias, self.residual_head.weight,
                self.residual_head.bias, self.layer_norm.weight, self.layer_norm.bias)

    def _rank(self, T):
        """the low-rank form (immtsf.ops.MMFXRankPFn / MMFXRankQFn, csrc/xrank.hip) takes these dimensions"""
        key = (int(T), bool(config.xattn_rank))
        if key not in self._rank_ok:
            self._rank_ok[key] = mmf_xrank_pw(T, self.C, self.d_attn, self.n_heads) > 0
        return self._rank_ok[key]

    def fold_weights(self):
        """the query half's product weights (parameters only: any stream, any time before forward())"""
        return mmf_xattn_q_fold(self.C, self.d_attn, self.n_heads, resolve_precision(self), self._params())

    def project_kv(self, E_txt, with_fold=True, proj=None):
        """key/value half (proj_k / proj_v + their MHA in-projections -> one (B,T,2d) tensor k | v): depends only on the text
        side, so a caller can run it on the text stream while the backbone is still producing Y_ts
        (lib.evaluation.forecast_and_fuse).  In the low-rank form the pair is (P, b_HO): the text side's projection onto the
        (2C+1) H columns the attention needs and the folded output bias (both carry gradients back to this half)."""
        if self._rank(E_txt.shape[1]):
            # proj = the nn.Linear the text side would have applied last (TTF_T2V_XAttn.proj_out, when E_txt is really its input Z):
            # composed into the low-rank projection, so that the (B T) x d x d product and its two gradient products never run
            extra = () if proj is None else (proj.weight, proj.bias)
            return MMFXRankPFn.apply(f32(E_txt), self.C, self.n_heads, resolve_precision(self),
                                     getattr(self.proj_q.weight, "_immtsf_bwd_hook", None), *self._params()[:9], *extra)
        if proj is not None:
            raise ValueError("project_kv(proj=...) needs the low-rank form (immtsf.config.xattn_rank, C <= 15, H <= 4)")
        KV = MMFXAttnKVFn.apply(f32(E_txt), self.n_heads, resolve_precision(self),
                                getattr(self.proj_q.weight, "_immtsf_bwd_hook", None), self.proj_k.weight, self.proj_v.weight,
                                self.attn.in_proj_weight, self.attn.in_proj_bias)
        # the query half's product weights depend on parameters only: form them here too, beside the backbone (with_fold=False:
        # the caller forms them elsewhere with fold_weights() and passes kv=(KV, fold) to forward())
        return KV, (self.fold_weights() if with_fold else None)

    def forward(self, Y_ts, E_txt, M_txt, kv=None):
        """Y_ts (B,T,C), E_txt (B,T,d_txt), M_txt (B,1)|(B,) bool -> (B,T,C).  kv: the result of project_kv(E_txt)
        when the caller computed it ahead of time."""
        B = Y_ts.shape[0]
        M_u8 = M_txt.reshape(B).to(torch.bool).view(torch.uint8)
        training = self.training and self.p_drop > 0.0
        self.last_seed = config.next_seed() if training else 0
        KV, fold = self.project_kv(E_txt) if kv is None else kv
        p = self._params()
        if self._rank(Y_ts.shape[1]):
            if fold is None:        # (a caller that asked for with_fold=False: b_HO only comes with P)
                KV, fold = self.project_kv(E_txt)
            return MMFXRankQFn.apply(f32(Y_ts), KV, fold, M_u8, self.d_attn, self.n_heads, float(self.kappa), self.p_drop, training,
                                     resolve_precision(self), self.last_seed, p[9], p[10])
        return MMFXAttnQFn.apply(f32(Y_ts), KV, M_u8, fold, self.n_heads, float(self.kappa), self.p_drop, training,
                                 resolve_precision(self), self.last_seed, p[0], *p[3:])

    def forward_loss(self, Y_ts, E_txt, M_txt, truth, mask, global_cnt, kv=None):
        """masked_mse(forward(Y_ts, E_txt, M_txt, kv), truth, mask, global_cnt=global_cnt) for a training step: in the low-rank form
        the head, the loss and the backward of both are one launch (immtsf.ops.MMFXRankQLossFn; the gradients exist when this
        returns, loss.backward() hands them on); otherwise the two calls."""
        if not (self._rank(Y_ts.shape[1]) and torch.is_grad_enabled() and global_cnt is not None and config.xattn_fused_loss):
            return masked_mse(self.forward(Y_ts, E_txt, M_txt, kv=kv), truth, mask, None, global_cnt)
        B = Y_ts.shape[0]
        M_u8 = M_txt.reshape(B).to(torch.bool).view(torch.uint8)
        training = self.training and self.p_drop > 0.0
        self.last_seed = config.next_seed() if training else 0
        P, bHO = self.project_kv(E_txt) if (kv is None or kv[1] is None) else kv
        p = self._params()
        return MMFXRankQLossFn.apply(f32(Y_ts), P, bHO, M_u8, f32(truth), f32(mask), global_cnt, self.d_attn, self.n_heads, float(self.kappa),
                                     self.p_drop, training, resolve_precision(self), self.last_seed, p[9], p[10])


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
