"""TimesNet backbone (reference models/TimesNet.py:9-152): DataEmbedding(2C+1 -> d_model) on [value; mask; time],
FFT period selection, 2-D Inception convolutions per period, adaptive aggregation.  Same signature/state_dict.
DataEmbedding is one HIP kernel, the two Linear maps run on the HIP GEMM, and each Inception block (the mean of six
same-padded convolutions, layers/Conv_Blocks.py:5-31) runs as ONE merged convolution on channels-last images: im2col +
MFMA GEMM with the bias / GELU epilogue (immtsf.ops.inception_merge / conv2d_same_cl, csrc/conv.hip).  The FFT period
selection keeps the reference's host sync (:13-16): the image shapes depend on it."""
import torch
import torch.fft
import torch.nn as nn
import torch.nn.functional as F

from immtsf.ops import INCEPTION_MAX, conv2d_same_cl, inception_merge, layer_norm, linear
from layers.Conv_Blocks import Inception_Block_V1
from layers.Embed import DataEmbedding
from models._common import pad_history, plain_instance_norm


class PeriodControl:
    """The period selection of a TimesNet (reference models/TimesNet.py:9-18) decides tensor SHAPES from data: a top-k over the
    batch-mean spectrum, read on the host (`.detach().cpu().numpy()`: a host sync per TimesBlock and step, in the reference and in this
    mirror's eager path).  A step engine that replays the step from a hipGraph (immtsf.train.SpecGraphStep) cannot sync: it ASSUMES
    the selection (`assumed`: per TimesBlock call of a forward, the top-k frequency indices as a tuple), and the captured kernels
    re-derive it on the device and raise `mismatch` when it differs -- the engine's optimizer is guarded by that word, and the engine
    repeats a mismatched step eagerly.  `observed`: what the last eager forward selected (the key of the graph to use next)."""

    def __init__(self):
        self.assumed, self.observed, self.mismatch, self._i, self._dev = None, [], None, 0, {}

    def begin(self):
        self._i, self.observed = 0, []

    def key(self):
        return tuple(self.observed)

    def assumed_indices(self, i, device):
        k = (i, self.assumed[i], str(device))
        if k not in self._dev:
            self._dev[k] = torch.tensor(self.assumed[i], dtype=torch.int64, device=device)
        return self._dev[k]


def FFT_for_Period(x, k=2, ctl=None):
    xf = torch.fft.rfft(x, dim=1)
    amp = xf.abs()
    freq = amp.mean(0).mean(-1)
    freq[0:1].zero_()          # (= the reference's `freq[0] = 0`, as a fill kernel: a Python scalar would be a host-to-device copy)
    if ctl is not None and ctl.assumed is not None:       # no host sync: shapes from the assumption, the device checks it
        top_dev = torch.topk(freq, k).indices
        want = ctl.assumed_indices(ctl._i, x.device)
        ctl.mismatch.logical_or_((top_dev != want).any().reshape(1).to(ctl.mismatch.dtype))
        top = torch.tensor(ctl.assumed[ctl._i]).numpy()
        ctl._i += 1
        return x.shape[1] // top, amp.mean(-1).index_select(1, want)
    top = torch.topk(freq, k).indices.detach().cpu().numpy()      # host sync, as in the reference (:13-16)
    if ctl is not None:
        ctl.observed.append(tuple(int(t) for t in top))
    return x.shape[1] // top, amp.mean(-1)[:, top]


class TimesBlock(nn.Module):
    def __init__(self, configs):
        super().__init__()
        self.ctl = None           # the owning TimesNet's PeriodControl
        self.seq_len, self.pred_len, self.k = configs.input_len, configs.pred_len, configs.top_k
        self.conv = nn.Sequential(Inception_Block_V1(configs.d_model, configs.d_ff, num_kernels=configs.num_kernels),
                                  nn.GELU(),
                                  Inception_Block_V1(configs.d_ff, configs.d_model, num_kernels=configs.num_kernels))

    def forward(self, x):
        B, T, N = x.size()
        total = self.seq_len + self.pred_len
        periods, weight = FFT_for_Period(x, self.k, self.ctl)
        inc1, act, inc2 = self.conv[0], self.conv[1], self.conv[2]
        merged = (x.is_cuda and isinstance(inc1, Inception_Block_V1) and isinstance(inc2, Inception_Block_V1) and isinstance(act, nn.GELU)
                  and getattr(act, "approximate", "none") == "none" and max(len(inc1.kernels), len(inc2.kernels)) <= INCEPTION_MAX)
        if merged:      # one averaged kernel per block and step, shared by all periods
            W1, b1, K1 = inception_merge(inc1)
            W2, b2, K2 = inception_merge(inc2)
        res = []
        for period in periods:
            period = int(period)
            length = total if total % period == 0 else (total // period + 1) * period
            out = F.pad(x, (0, 0, 0, length - total)) if length != total else x
            if merged:      # (B, length, N) IS the channels-last image (B, length / period, period, N): no permutes
                img = conv2d_same_cl(out.reshape(B, length // period, period, N), W1, b1, K1, act="gelu")
                out = conv2d_same_cl(img, W2, b2, K2).reshape(B, -1, N)
            else:
                out = out.reshape(B, length // period, period, N).permute(0, 3, 1, 2).contiguous()
                out = self.conv(out).permute(0, 2, 3, 1).reshape(B, -1, N)
            res.append(out[:, :total, :])
        res = torch.stack(res, dim=-1)
        w = F.softmax(weight, dim=1).unsqueeze(1).unsqueeze(1)
        return (res * w).sum(-1) + x


class TimesNet(nn.Module):
    def __init__(self, configs):
        super().__init__()
        self.configs = configs
        self.input_len = self.seq_len = configs.input_len
        self.pred_len = configs.pred_len
        print("seq len:", self.seq_len, self.pred_len)
        self.model = nn.ModuleList([TimesBlock(configs) for _ in range(configs.e_layers)])
        self.immtsf_period_ctl = PeriodControl()      # (shared by the blocks; plain attribute: not part of the state_dict)
        for blk in self.model:
            blk.ctl = self.immtsf_period_ctl
        self.enc_embedding = DataEmbedding(2 * configs.enc_in + 1, configs.d_model, configs.embed, configs.freq, configs.dropout)
        self.layer = configs.e_layers
        self.layer_norm = nn.LayerNorm(configs.d_model)
        self.predict_linear = nn.Linear(self.seq_len + self.pred_len, self.pred_len + self.seq_len)
        self.projection = nn.Linear(configs.d_model, configs.c_out, bias=True)
        self.zeros_pad = torch.zeros(configs.batch_size, max(configs.input_len, configs.pred_len), configs.enc_in).to(configs.device)

    def forecasting(self, tp_to_predict, observed_data, observed_tp, observed_mask):
        self.immtsf_period_ctl.begin()
        tp_to_predict, data, tp, mask, Lp = pad_history(self.zeros_pad, self.input_len, self.pred_len, tp_to_predict,
                                                        observed_data, observed_tp, observed_mask)
        x, means, stdev = plain_instance_norm(data)
        enc = self.enc_embedding(torch.cat([x, mask, tp.unsqueeze(-1)], dim=-1))             # (B, L, d_model)
        enc = torch.cat([enc, tp_to_predict.unsqueeze(-1).expand(-1, -1, enc.size(-1))], dim=1)
        enc = linear(enc.permute(0, 2, 1), self.predict_linear.weight, self.predict_linear.bias).permute(0, 2, 1)
        for i in range(self.layer):
            enc = layer_norm(self.model[i](enc), self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        dec = linear(enc, self.projection.weight, self.projection.bias)
        dec = dec * stdev[:, 0, :].unsqueeze(1) + means[:, 0, :].unsqueeze(1)
        return dec[:, -self.pred_len:, :][:, :Lp, :]


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
