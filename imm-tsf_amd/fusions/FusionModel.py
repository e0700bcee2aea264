"""FusionModel on MI355X: TTF (text-timestamp fusion) followed by MMF (modality fusion).

Same registry, constructor (reads the same `args.*` attributes) and forward signature as the reference
(fusions/FusionModel.py:14-113); `args.TTF_module` / `args.MMF_module` may be a registry string or a class.
The NaN guards of the reference (three `torch.isnan(x).any()` host syncs per step, :103-112) are honoured
according to `immtsf.config.nan_check`: "sync" (reference behaviour), "deferred" (the default: no sync; lib.evaluation raises at the next call, or call
`check_nan()`), or "off".
"""
import torch
import torch.nn as nn

from fusions.MMF_GR_Add import MMF_GR_Add
from fusions.MMF_XAttn_Add import MMF_XAttn_Add
from fusions.TTF_RecAvg import TTF_RecAvg
from fusions.TTF_T2V_XAttn import TTF_T2V_XAttn
from immtsf import config

_TTF_CLASSES = {"TTF_RecAvg": TTF_RecAvg, "TTF_T2V_XAttn": TTF_T2V_XAttn}
_MMF_CLASSES = {"MMF_GR_Add": MMF_GR_Add, "MMF_XAttn_Add": MMF_XAttn_Add}


def _resolve(ref, table):
    return table.get(ref, ref) if isinstance(ref, str) else ref


class FusionModel(nn.Module):
    def __init__(self, args):
        super().__init__()
        ttf_cls = _resolve(args.TTF_module, _TTF_CLASSES)
        mmf_cls = _resolve(args.MMF_module, _MMF_CLASSES)
        print(f"Using TTF module: {args.TTF_module}")
        print(f"Using MMF module: {args.MMF_module}")
        common = dict(max_length=args.max_length, device=args.device, use_text_embeddings=args.use_text_embeddings,
                      dropout=args.dropout, d_txt=args.d_txt)
        if ttf_cls is TTF_RecAvg:
            self.ttf = ttf_cls(args.llm_model_fusion, args.llm_layers_fusion, recency_sigma=args.recency_sigma, **common)
        else:
            self.ttf = ttf_cls(args.llm_model_fusion, args.llm_layers_fusion, n_heads_fusion=args.n_heads_fusion, **common)
        d_txt = self.ttf.d_txt
        if mmf_cls is MMF_GR_Add:
            self.mmf = mmf_cls(d_txt=d_txt, C=args.C, hidden_dim=args.C, dropout=args.dropout)
        else:
            self.mmf = mmf_cls(d_txt=d_txt, C=args.C, d_attn=d_txt, n_heads_fusion=args.n_heads_fusion,
                               dropout=args.dropout, kappa=args.kappa)
        precision = getattr(args, "immtsf_precision", None)
        if precision is not None:
            self.ttf.precision = precision
            self.mmf.precision = precision

    def fused_tail(self, rows: int, T: int) -> bool:
        """TTF_T2V_XAttn's proj_out composed into MMF_XAttn_Add's low-rank projection (csrc/xrank.hip "_z" form): three (B T)-row d x d
        products become 24-column ones plus a short parameter-only chain.  Measured (cfg2, bf16, fused / separate): 64 windows 0.538 /
        0.550 ms, 256: 0.929 / 0.996, 1024: 1.89 / 2.05, 4096: 6.33 / 6.31 (the backbone's branch bounds that step) -- so "auto" takes it
        wherever the pair allows.  Same function, same parameters, same gradients
        (tests/test_gpu_fusion.py::test_fused_tail_equals_separate_blocks)."""
        mode = config.fuse_tail
        if mode is False or mode == "off" or type(self.ttf) is not TTF_T2V_XAttn or not hasattr(self.mmf, "_rank"):
            return False
        if not (config.xattn_rank and self.mmf._rank(T)):
            return False
        return True

    def grad_buckets(self, T: int):
        """the parameters as the groups whose gradients a training step completes together, for immtsf.train.FlatTrainer's buckets:
        [MMF's (+ TTF's proj_out when the pair runs fused: its gradient leaves MMF's parameter chain)], then TTF's backward phases
        (TTF_T2V_XAttn.grad_phases) or TTF's parameters as one group.  Returns (buckets, names); names[i] in {"mmf", "ttf", "ttf_a",
        "ttf_b", "ttf_c"}."""
        fused = self.fused_tail(0, T)
        mmf = list(self.mmf.parameters())
        if hasattr(self.ttf, "grad_phases"):
            ph = self.ttf.grad_phases(tail=not fused)
            if fused:
                mmf = mmf + [self.ttf.proj_out.weight, self.ttf.proj_out.bias]
            return [mmf] + ph, ["mmf", "ttf_a", "ttf_b", "ttf_c"]
        return [mmf, list(self.ttf.parameters())], ["mmf", "ttf"]

    def text_side(self, notes_input, tau, t_hat):
        """everything that depends on the text only: (E_txt or its pre-projection Z, M_txt, kv) with kv = mmf.project_kv(...) when the
        modality block has a text-only half (None otherwise).  What forward(), lib.evaluation.forecast_and_fuse and the step engines
        call; E_txt is Z when fused_tail() holds (its only consumer, the low-rank projection, then applies proj_out itself)."""
        from fusions._common import prep_t_hat
        B = tau.shape[0]
        T = prep_t_hat(t_hat, B).shape[1]
        if self.fused_tail(B * T, T):
            Z, M_txt = self.ttf(notes_input, tau, t_hat, tail="handover")      # (Z's only consumer is the projection below: config.z_handover)
            return Z, M_txt, self.mmf.project_kv(Z, proj=self.ttf.proj_out)
        E_txt, M_txt = self.ttf(notes_input, tau, t_hat)
        return E_txt, M_txt, (self.mmf.project_kv(E_txt) if hasattr(self.mmf, "project_kv") else None)

    def forward(self, notes_input, tau, t_hat, Y_ts):
        sync = config.nan_check == "sync"
        if sync and torch.isnan(Y_ts).any():
            print(f"Y_ts: {Y_ts}")
            raise ValueError("Y_ts contains NaN values.")
        E_txt, M_txt, kv = self.text_side(notes_input, tau, t_hat)
        if sync and torch.isnan(E_txt).any():
            raise ValueError("E_txt contains NaN values.")
        Y_out = self.mmf(Y_ts, E_txt, M_txt) if kv is None else self.mmf(Y_ts, E_txt, M_txt, kv=kv)
        if sync and torch.isnan(Y_out).any():
            raise ValueError("Y_out contains NaN values.")
        return Y_out

    def check_nan(self):
        """deferred-mode check of the note-embedding NaN flag set by the kernels."""
        if hasattr(self.ttf, "check_nan"):
            self.ttf.check_nan()


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
