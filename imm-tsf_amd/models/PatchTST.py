"""PatchTST backbone with the reference's constructor/forecasting signature and state_dict keys
(models/PatchTST.py:9-131).  The (value, mask, time)-interleaved series is patched by `PatchEmbedding`
(6 timestamps x 3 channels per 18-wide patch: the time-aware patch embedding of this backbone) and encoded by the
HIP-backed Encoder[EncoderLayer(AttentionLayer(FullAttention(mask_flag=False)))]."""
import torch
from torch import nn

from layers.Embed import PatchEmbedding
from layers.SelfAttention_Family import AttentionLayer, FullAttention
from layers.Transformer_EncDec import Encoder, EncoderLayer
from immtsf.ops import linear
from models._common import pad_history, plain_instance_norm


class FlattenHead(nn.Module):
    def __init__(self, n_vars, nf, target_window, head_dropout=0):
        super().__init__()
        self.n_vars = n_vars
        self.flatten = nn.Flatten(start_dim=-2)
        self.linear = nn.Linear(nf, target_window)
        self.dropout = nn.Dropout(head_dropout)

    def forward(self, x, tp_to_predict):          # x (B, n_vars, d_model, patch_num)
        K = x.shape[1]
        x = torch.cat([self.flatten(x), tp_to_predict.unsqueeze(1).expand(-1, K, -1)], dim=-1)
        return self.dropout(linear(x, self.linear.weight, self.linear.bias))


class PatchTST(nn.Module):
    immtsf_graphable = True      # no host syncs / data-dependent shapes in forecasting(): a step may be captured into a hipGraph

    def immtsf_sink_params(self):
        """the parameters whose gradients the HIP backward writes in place when they are gradient sinks (immtsf.train.FlatTrainer): the
        encoder layers' attention projections and feed-forward block -- the large products, whose weight gradients a step with a
        parameter branch takes off the backbone's dependent chain (ops.LinearBf16Fn / FFNBlockFn)"""
        out = []
        for lyr in self.encoder.attn_layers:
            a = lyr.attention
            for m in (a.query_projection, a.key_projection, a.value_projection, a.out_projection, lyr.conv1, lyr.conv2, lyr.norm2):
                out += [m.weight, m.bias]
        return out

    def __init__(self, configs, patch_len=6 * 3, stride=3 * 3):
        super().__init__()
        self.input_len = configs.input_len
        self.seq_len = 3 * configs.input_len
        self.pred_len = configs.pred_len
        self.patch_embedding = PatchEmbedding(configs.d_model, patch_len, stride, stride, configs.dropout)
        self.encoder = Encoder(
            [EncoderLayer(AttentionLayer(FullAttention(False, configs.factor, attention_dropout=configs.dropout,
                                                       output_attention=False), configs.d_model, configs.n_heads),
                          configs.d_model, configs.d_ff, dropout=configs.dropout, activation=configs.activation)
             for _ in range(configs.e_layers)],
            norm_layer=torch.nn.LayerNorm(configs.d_model))
        self.head_nf = configs.d_model * int((self.seq_len - patch_len) / stride + 2)
        self.head = FlattenHead(configs.enc_in, self.head_nf + configs.pred_len, configs.pred_len,
                                head_dropout=configs.dropout)
        self.zeros_pad = torch.zeros(configs.batch_size, max(configs.input_len, configs.pred_len), configs.enc_in).to(configs.device)

    def forecasting(self, tp_to_predict, observed_data, observed_tp, observed_mask):
        """tp_to_predict (B,Lp); observed_data/mask (B,L,K); observed_tp (B,L) -> (B,Lp,K)"""
        tp_to_predict, data, tp, mask, Lp = pad_history(self.zeros_pad, self.input_len, self.pred_len, tp_to_predict,
                                                        observed_data, observed_tp, observed_mask)
        B, L, K = data.shape
        x, means, stdev = plain_instance_norm(data)
        # interleave (value, mask, time) along time: (B, 3L, K) -> (B, K, 3L)
        x = torch.stack([x, mask, tp.unsqueeze(-1).expand(-1, -1, K)], dim=2).reshape(B, 3 * L, K).permute(0, 2, 1)
        enc, n_vars = self.patch_embedding(x)                       # (B*K, P, d_model)
        enc, _ = self.encoder(enc)
        enc = enc.reshape(-1, n_vars, enc.shape[-2], enc.shape[-1]).permute(0, 1, 3, 2)
        dec = self.head(enc, tp_to_predict).permute(0, 2, 1)        # (B, pred_len, K)
        dec = dec * stdev[:, 0, :].unsqueeze(1) + means[:, 0, :].unsqueeze(1)
        return dec[:, -self.pred_len:, :][:, :Lp, :]


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
