"""TimeLLM backbone (reference models/TimeLLM.py:19-278): prompt + patch reprogramming onto a frozen LLM.

Same constructor / `forecasting` signature and state_dict keys.  The frozen LLM and its tokenizer come from the HF hub
in the reference (:128-159); with no network they cannot be instantiated, so `configs.immtsf_offline_llm = True`
builds a RANDOM-INIT GPT-2 of the requested depth and a deterministic byte-level tokenizer stand-in.  The wrapper's
composition (prompt -> tokens -> two patch embeddings -> reprogramming -> LLM -> hidden[:, -total:, :d_ff] -> head) is pinned
against the reference run the same way (its loader patched to the same stand-ins, tests/golden/make_golden.py gen_timellm ->
tests/golden/model_timellm.npz, test_gpu_fusion.py::test_timellm_forecasting_vs_reference_golden); parity with PRETRAINED
weights stays unpinned.  The sub-layers on the
hot-path scope -- PatchEmbedding on values and on timestamps, ReprogrammingLayer projections, FlattenHead -- use the
HIP GEMM."""
from math import sqrt

import torch
import torch.nn as nn
import torch.nn.functional as F

from immtsf import config
from immtsf.ops import linear, shared_kv_attention
from layers.Embed import PatchEmbedding
from models._common import masked_instance_norm


class FlattenHead(nn.Module):
    def __init__(self, nf, target_window, head_dropout=0.0):
        super().__init__()
        self.flatten = nn.Flatten(start_dim=-2)
        self.linear = nn.Linear(nf, target_window)
        self.dropout = nn.Dropout(head_dropout)

    def forward(self, x):
        return self.dropout(linear(self.flatten(x), self.linear.weight, self.linear.bias))


class ReprogrammingLayer(nn.Module):
    """cross-attention of the patch tokens (queries) onto `ts_vocab_size` mapped word prototypes (keys/values shared by
    the whole batch), reference :32-61"""

    def __init__(self, d_model, n_heads, d_keys=None, d_llm=None, attention_dropout=0.1):
        super().__init__()
        d_keys = d_keys or (d_model // n_heads)
        self.n_heads = n_heads
        self.query_projection = nn.Linear(d_model, d_keys * n_heads)
        self.key_projection = nn.Linear(d_llm, d_keys * n_heads)
        self.value_projection = nn.Linear(d_llm, d_keys * n_heads)
        self.out_projection = nn.Linear(d_keys * n_heads, d_llm)
        self.dropout = nn.Dropout(attention_dropout)

    def forward(self, Q, K_src, V_src):
        Bm, Lq, _ = Q.shape
        Vs, H = K_src.shape[0], self.n_heads
        q = linear(Q, self.query_projection.weight, self.query_projection.bias).view(Bm, Lq, H, -1)
        k = linear(K_src, self.key_projection.weight, self.key_projection.bias).view(Vs, H, -1)
        v = linear(V_src, self.value_projection.weight, self.value_projection.bias).view(Vs, H, -1)
        scale = 1.0 / sqrt(K_src.size(-1) // H)          # d_llm / H, as the reference has it (:51)
        p = float(self.dropout.p)
        training = self.training and p > 0.0
        out = shared_kv_attention(q, k, v, scale, p, training, config.next_seed() if training else 0, 16 + 1023)
        return linear(out.reshape(Bm, Lq, -1), self.out_projection.weight, self.out_projection.bias)


class _ByteTokenizer:
    """offline stand-in: one token per byte (ids < 256), right-padded with the pad id"""
    eos_token = "<eos>"
    pad_token = "<eos>"

    def __call__(self, prompts, return_tensors="pt", padding=True, truncation=True, max_length=512):
        ids = [list(p.encode("utf-8"))[:max_length] for p in prompts]
        n = max(len(i) for i in ids)
        t = torch.full((len(ids), n), 0, dtype=torch.long)
        for r, i in enumerate(ids):
            t[r, :len(i)] = torch.tensor(i, dtype=torch.long)
        return type("Enc", (), {"input_ids": t})()

    def add_special_tokens(self, _):
        pass


class TimeLLM(nn.Module):
    def __init__(self, configs):
        super().__init__()
        self.input_len = self.seq_len = configs.input_len
        self.pred_len = configs.pred_len
        self.use_norm = configs.use_norm
        self.d_ff = configs.d_ff
        self.num_tokens = configs.ts_vocab_size
        self.patch_len = configs.input_token_len
        self.stride = configs.stride
        self.domain_des = configs.domain_des
        self.top_k = configs.top_k
        self.C = configs.C
        if configs.llm_model_timellm == "LLAMA":
            self.d_llm = 4096
        elif configs.llm_model_timellm in ("GPT2", "BERT"):
            self.d_llm = 768
        else:
            raise ValueError("Unknown llm_model for TimeLLM")
        self.patch_nums = max(1, (self.seq_len - self.patch_len) // self.stride + 2)
        self.head_nf = self.d_ff * self.patch_nums
        self._get_model_and_tokenizer(configs.llm_model_timellm, configs.llm_layers_timellm,
                                      getattr(configs, "immtsf_offline_llm", False))
        self._get_llm_pad_token()
        for p in self.llm_model.parameters():
            p.requires_grad = False
        self.dropout = nn.Dropout(configs.dropout)
        self.patch_embedding = PatchEmbedding(configs.d_model, self.patch_len, self.stride, self.stride, configs.dropout)
        self.word_embeddings = self.llm_model.get_input_embeddings().weight
        self.mapping_layer = nn.Linear(self.word_embeddings.size(0), self.num_tokens)
        self.reprogramming_layer = ReprogrammingLayer(configs.d_model, configs.n_heads, d_llm=self.d_llm)
        self.output_projection = FlattenHead(self.head_nf, self.pred_len, head_dropout=configs.dropout)
        self.zeros_pad = torch.zeros(configs.batch_size, max(self.input_len, self.pred_len), self.C, device=configs.device)

    def _get_model_and_tokenizer(self, model_name, layers, offline):
        from transformers import BertConfig, BertModel, BertTokenizer, GPT2Config, GPT2Model, GPT2Tokenizer
        from transformers import LlamaConfig, LlamaModel, LlamaTokenizer
        table = {"LLAMA": ("huggyllama/llama-7b", LlamaConfig, LlamaModel, LlamaTokenizer),
                 "GPT2": ("openai-community/gpt2", GPT2Config, GPT2Model, GPT2Tokenizer),
                 "BERT": ("google-bert/bert-base-uncased", BertConfig, BertModel, BertTokenizer)}
        if model_name not in table:
            raise ValueError("Unsupported LLM")
        repo, Cfg, Model, Tok = table[model_name]
        if offline:
            if model_name != "GPT2":
                raise ValueError("immtsf_offline_llm supports GPT2 only")
            # (a dict: extra GPT2Config fields -- the parity fixture uses a 320-entry vocabulary, tests/golden/make_golden.py)
            extra = offline if isinstance(offline, dict) else {}
            self.llm_model = GPT2Model(GPT2Config(n_layer=layers, **extra))
            self.tokenizer = _ByteTokenizer()
            return
        cfg = Cfg.from_pretrained(repo)
        cfg.num_hidden_layers = layers
        cfg.output_hidden_states = True
        cfg.output_attentions = True
        self.llm_model = Model.from_pretrained(repo, config=cfg)
        self.tokenizer = Tok.from_pretrained(repo)

    def _get_llm_pad_token(self):
        if self.tokenizer.eos_token:
            self.tokenizer.pad_token = self.tokenizer.eos_token
        else:
            self.tokenizer.add_special_tokens({"pad_token": "[PAD]"})
            self.tokenizer.pad_token = "[PAD]"

    def _get_prompt(self, x_enc):
        B, L, N = x_enc.shape
        mins, maxs = x_enc.min(dim=1)[0], x_enc.max(dim=1)[0]
        meds = x_enc.median(dim=1).values
        trend = x_enc.diff(dim=1).sum(dim=1).mean(dim=1)
        spec = torch.fft.rfft(x_enc.permute(0, 2, 1), dim=-1)
        corr = torch.fft.irfft(spec * spec.conj(), n=L, dim=-1).mean(dim=1)
        _, lags = corr.topk(min(self.top_k, L), dim=-1)
        if lags.size(1) < self.top_k:
            lags = torch.cat([lags, lags[:, -1, None].repeat(1, self.top_k - lags.size(1))], dim=-1)
        prompts = []
        for b in range(B):
            tr = "upward" if trend[b].item() > 0 else "downward"
            prompts.append(f"<|start_prompt|>Dataset: {self.domain_des}. Forecast next {self.pred_len} from past "
                           f"{self.input_len}. Min {mins[b].tolist()}, Max {maxs[b].tolist()}, Median {meds[b].tolist()}, "
                           f"Trend {tr}, Top lags {lags[b].tolist()}.<|end_prompt|>")
        return prompts

    def forecasting(self, tp_to_predict, observed_data, observed_tp, observed_mask):
        B, L, N = observed_data.shape
        if L < self.input_len:
            n = self.input_len - L
            observed_data = torch.cat([observed_data, self.zeros_pad[:B, :n, :]], dim=1)
            observed_mask = torch.cat([observed_mask, self.zeros_pad[:B, :n, :]], dim=1)
            observed_tp = torch.cat([observed_tp, self.zeros_pad[:B, :n, 0]], dim=1)
        Lp = tp_to_predict.size(1)
        x, means, stdev = masked_instance_norm(observed_data, observed_mask)
        tokens = self.tokenizer(self._get_prompt(x), return_tensors="pt", padding=True, truncation=True,
                                max_length=512).input_ids.to(x.device)
        prompt_embeds = self.llm_model.get_input_embeddings()(tokens)

        def patches(series):                                   # (B, N, L) -> (B*N, Pn, d_model)
            if series.size(-1) < self.patch_len:
                series = F.pad(series, (0, self.patch_len - series.size(-1)))
            return self.patch_embedding(series)
        ts_out, n_vars = patches(x.permute(0, 2, 1))
        tp_out, _ = patches(observed_tp.unsqueeze(1).repeat(1, N, 1))
        src = linear(self.word_embeddings.permute(1, 0), self.mapping_layer.weight, self.mapping_layer.bias).permute(1, 0)
        rep = self.reprogramming_layer(ts_out + tp_out, src, src)
        rep = rep.view(B, N, self.patch_nums, self.d_llm).permute(0, 2, 1, 3).reshape(B, -1, self.d_llm)
        hidden = self.llm_model(inputs_embeds=torch.cat([prompt_embeds, rep], dim=1)).last_hidden_state
        total = self.patch_nums * n_vars
        dec = hidden[:, -total:, :self.d_ff].view(B, self.patch_nums, n_vars, self.d_ff)
        dec = dec.permute(0, 2, 3, 1).reshape(B * n_vars, self.d_ff, self.patch_nums)
        out = self.output_projection(dec).view(B, n_vars, self.pred_len).permute(0, 2, 1)
        if self.use_norm:
            out = out * stdev + means
        return out[:, :Lp, :]


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
