"""TTF_T2V_XAttn on MI355X: Time2Vec-augmented cross-attention from one learned query to a window's ragged notes.

Interface and state_dict keys follow the reference module (fusions/TTF_T2V_XAttn.py:27-184); the computation is
the HIP pipeline `immtsf_ttf_t2v_xattn_forward/backward` (imm-tsf_amd/csrc/fusion_blocks.hip):
  note mask -> packed ragged rows (offsets) -> input_proj GEMM (gathered) + Time2Vec -> KV_proj GEMM -> packed k|v
  in-projection once per NOTE (the reference projects T-fold copies) -> single-query masked softmax per window and
  head (+ per-(b,t) attention-weight dropout) -> out_proj (+no-note zeroing, +Q residual) -> LayerNorm+dropout ->
  proj_out.
The nn.Linear / nn.MultiheadAttention / nn.LayerNorm children are parameter containers only (they give identical
key names, shapes and initialisation); their own forward is never called.
"""
import torch
import torch.nn as nn

from fusions._common import NanFlag, f32, prep_t_hat, resolve_precision
from fusions.load_llm import get_d_model, load_llm
from immtsf import config
from immtsf.ops import PackedNotes, TTFT2VXAttnFn


class Time2Vec(nn.Module):
    """Parameters of the Time2Vec encoding [w0*t+b0 ; sin(W*t+b)] (reference :7-24)."""

    def __init__(self, d_tau: int):
        super().__init__()
        assert d_tau > 1, "d_tau must be > 1"
        self.linear = nn.Linear(1, 1)
        self.periodic = nn.Linear(1, d_tau - 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:   # kept for API completeness (eager torch, any device)
        return torch.cat([self.linear(x), torch.sin(self.periodic(x))], dim=-1)


class TTF_T2V_XAttn(nn.Module):
    def __init__(self, llm_model_fusion: str, llm_layers_fusion: int, max_length: int = 1024, device: str = "cpu",
                 use_text_embeddings: bool = True, n_heads_fusion: int = 1, dropout: float = 0.1,
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
        self.d_model = d_model
        self.d_tau = self.d_txt // 2
        self.max_length = max_length
        self.n_heads = n_heads_fusion
        self.p_drop = float(dropout)
        self.time2vec = Time2Vec(self.d_tau)
        self.KV_proj = nn.Linear(self.d_txt + self.d_tau, self.d_txt)
        self.attn = nn.MultiheadAttention(embed_dim=self.d_txt, num_heads=n_heads_fusion, dropout=dropout,
                                          batch_first=True)
        self.layer_norm = nn.LayerNorm(self.d_txt)
        self.dropout = nn.Dropout(dropout)
        self.proj_out = nn.Linear(self.d_txt, self.d_txt)
        self.Q_param = nn.Parameter(torch.randn(1, 1, self.d_txt))
        self.precision = None          # None -> immtsf.config.precision
        self.last_seed = 0             # Philox key of the most recent training forward (mask export in tests)
        self._nan = NanFlag()

    def _params(self):
        ip = self.input_proj
        return (self.Q_param, None if ip is None else ip.weight, None if ip is None else ip.bias,
                self.time2vec.linear.weight, self.time2vec.linear.bias, self.time2vec.periodic.weight,
                self.time2vec.periodic.bias, self.KV_proj.weight, self.KV_proj.bias, self.attn.in_proj_weight,
                self.attn.in_proj_bias, self.attn.out_proj.weight, self.attn.out_proj.bias, self.layer_norm.weight,
                self.layer_norm.bias, self.proj_out.weight, self.proj_out.bias)

    def grad_phases(self, tail: bool = True):
        """the parameters in the three groups whose gradients the backward completes one after the other (include/immtsf.h
        IMMTSF_BWD_PHASE_A / B / C): [out_proj, layer_norm (, proj_out)], [attn.in_proj, Q_param], [input_proj, time2vec, KV_proj].  A
        trainer that makes each group a bucket (immtsf.train.FlatTrainer) lets a data-parallel step all-reduce a group while the later
        phases still run.  tail=False: without proj_out (a consumer that folds it -- FusionModel.fused_tail -- produces its gradient)."""
        a = [self.attn.out_proj.weight, self.attn.out_proj.bias, self.layer_norm.weight, self.layer_norm.bias]
        if tail:
            a += [self.proj_out.weight, self.proj_out.bias]
        b = [self.attn.in_proj_weight, self.attn.in_proj_bias, self.Q_param]
        c = ([] if self.input_proj is None else [self.input_proj.weight, self.input_proj.bias]) + \
            list(self.time2vec.parameters()) + [self.KV_proj.weight, self.KV_proj.bias]
        return [a, b, c]

    def forward(self, notes_input, tau: torch.Tensor, t_hat: torch.Tensor, tail: bool = True):
        """notes_input (B,N,d_model) zero-padded embeddings -- or a PackedNotes over a resident embedding matrix
        (immtsf.data.ResidentStore.collate: no padded tensor, no note-mask re-derivation) --, tau (B,N),
        t_hat (B,T) or (T,) -> E_txt (B,T,d_txt), M_txt (B,1) bool.  tail=False (not part of the reference's signature; used by
        FusionModel.text_side): stop in front of proj_out and return Z = dropout(LayerNorm(E_attn + Q)) in E_txt's place -- for a consumer
        that composes proj_out into its own projection (MMF_XAttn_Add.project_kv(Z, proj=self.proj_out)); tail="handover": the same, and
        the caller promises that this projection is Z's ONLY consumer (immtsf.config.z_handover: in bf16 mode Z then exists as its bf16
        image alone -- the returned fp32 tensor is a placeholder -- and the gradient comes back in low-rank form)."""
        if not self.use_text_embeddings:
            raise NotImplementedError("raw-text mode is not part of the MI355X hot path")
        packed = notes_input if isinstance(notes_input, PackedNotes) else None
        V = packed.emb if packed is not None else f32(notes_input)
        B = tau.shape[0]
        t_hat = prep_t_hat(t_hat, B)
        T = t_hat.shape[1]
        training = self.training and self.p_drop > 0.0
        self.last_seed = config.next_seed() if training else 0
        mode = config.nan_check
        flag = None if mode == "off" else self._nan.get(V.device)
        E_txt, M = TTFT2VXAttnFn.apply(V, f32(tau), T, self.n_heads, self.p_drop, training, resolve_precision(self),
                                       self.last_seed, flag, packed,
                                       None if packed is None else packed.lengths,
                                       "handover" if tail == "handover" else (not tail), *self._params())
        if mode == "sync":
            self._nan.raise_if_set("Input embeddings V contain NaN values.")
        return E_txt, M.view(torch.bool).view(B, 1)

    def check_nan(self):
        """deferred-mode companion: raises if any forward since the last check saw NaN note embeddings."""
        self._nan.raise_if_set("Input embeddings V contain NaN values.")


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
