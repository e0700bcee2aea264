"""TTF_RecAvg on MI355X: Gaussian recency-weighted average of a window's ragged notes for every forecast step.

Interface/state_dict follow the reference (fusions/TTF_RecAvg.py:8-112); computed by
`immtsf_ttf_recavg_forward/backward`: note mask -> packed rows -> input_proj GEMM (gathered) -> recency weights and
weighted mean per (window, step) on packed rows -> LayerNorm+dropout -> proj GEMM.  tau and t_hat are used exactly
as given (the reference mixes raw tau with normalised t_hat; that quirk is data, not code).
"""
import torch
import torch.nn as nn

from fusions._common import NanFlag, f32, prep_t_hat, resolve_precision
from fusions.load_llm import get_d_model, load_llm
from immtsf import config
from immtsf.ops import TTFRecAvgFn


class TTF_RecAvg(nn.Module):
    def __init__(self, llm_model_fusion: str, llm_layers_fusion: int, max_length: int = 1024, device: str = "cpu",
                 use_text_embeddings: bool = True, recency_sigma: float = 1.0, dropout: float = 0.1,
                 d_txt: int | None = 768):
        super().__init__()
        self.use_text_embeddings = use_text_embeddings
        if not use_text_embeddings:
            self.tokenizer, self.llm_model = load_llm(llm_model_fusion, llm_layers_fusion, device)
        d_model = get_d_model(llm_model_fusion)
        if d_txt is not None:
            self.input_proj = nn.Linear(d_model, d_txt)
            self.d_txt = d_txt
        else:
            self.input_proj = None
            self.d_txt = d_model
        self.max_length = max_length
        assert recency_sigma > 0, "recency_sigma must be > 0"
        self.log_recency_sigma = nn.Parameter(torch.log(torch.tensor(recency_sigma)))
        self.proj = nn.Linear(self.d_txt, self.d_txt)
        self.layer_norm = nn.LayerNorm(self.d_txt)
        self.dropout = nn.Dropout(dropout)
        self.p_drop = float(dropout)
        self.precision = None
        self.last_seed = 0
        self._nan = NanFlag()

    def _params(self):
        ip = self.input_proj
        return (self.log_recency_sigma, None if ip is None else ip.weight, None if ip is None else ip.bias,
                self.layer_norm.weight, self.layer_norm.bias, self.proj.weight, self.proj.bias)

    def forward(self, notes_input, tau: torch.Tensor, t_hat: torch.Tensor):
        if not self.use_text_embeddings:
            raise NotImplementedError("raw-text mode is not part of the MI355X hot path")
        V = f32(notes_input)
        B = V.shape[0]
        t_hat = prep_t_hat(t_hat, B)
        training = self.training and self.p_drop > 0.0
        self.last_seed = config.next_seed() if training else 0
        mode = config.nan_check
        flag = None if mode == "off" else self._nan.get(V.device)
        E_txt, M = TTFRecAvgFn.apply(V, f32(tau), f32(t_hat), self.p_drop, training, resolve_precision(self),
                                     self.last_seed, flag, *self._params())
        if mode == "sync":
            self._nan.raise_if_set("Input embeddings V contain NaN values.")
        return E_txt, M.view(torch.bool).view(B, 1)

    def check_nan(self):
        self._nan.raise_if_set("Input embeddings V contain NaN values.")


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
