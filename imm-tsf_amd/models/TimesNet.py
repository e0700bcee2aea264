"""TimesNet backbone (reference models/TimesNet.py:9-152): DataEmbedding(2C+1 -> d_model) on [value; mask; time],
FFT period selection, 2-D Inception convolutions per period, adaptive aggregation.  Same signature/state_dict.
DataEmbedding is one HIP kernel, the two Linear maps run on the HIP GEMM, and each Inception block (the mean of six
same-padded convolutions, layers/Conv_Blocks.py:5-31) runs as ONE merged convolution on channels-last images: im2col +
MFMA GEMM with the bias / GELU epilogue (immtsf.ops.inception_merge / conv2d_period, csrc/conv.hip).  The FFT period
selection -- a host read in the reference (:13-16), because the periods shape its images -- stays on the device: the images are
prefixes of one static position-major buffer and the kernels read the period from device memory, so forecasting() has no host sync
and a training step can be replayed from a hipGraph."""
import torch
import torch.fft
import torch.nn as nn
import torch.nn.functional as F

from immtsf.ops import (INCEPTION_MAX, conv2d_periods, conv2d_same_cl, inception_merge, inception_periods, inception_periods_ok, layer_norm, linear,
                        period_aggregate, period_rows)
from layers.Conv_Blocks import Inception_Block_V1
from layers.Embed import DataEmbedding
from models._common import pad_history, plain_instance_norm


def FFT_for_Period(x, k=2):
    xf = torch.fft.rfft(x, dim=1)
    amp = xf.abs()
    freq = amp.mean(0).mean(-1)
    freq[0] = 0
    top = torch.topk(freq, k).indices.detach().cpu().numpy()      # host sync, as in the reference (:13-16)
    return x.shape[1] // top, amp.mean(-1)[:, top]


def fft_for_period_device(x, k):
    """FFT_for_Period without the host read: (top (k) int64 frequency indices ON THE DEVICE, weight (B, k)).  The periods they stand
    for only ever reach kernels as device numbers (ops.period_rows / conv2d_period)."""
    amp = torch.fft.rfft(x, dim=1).abs()
    freq = amp.mean(0).mean(-1)
    freq = torch.cat([freq.new_zeros(1), freq[1:]])       # (= `freq[0] = 0`, without a host-to-device scalar copy)
    top = torch.topk(freq, k).indices
    return top, amp.mean(-1).index_select(1, top)


class TimesBlock(nn.Module):
    def __init__(self, configs):
        super().__init__()
        self.seq_len, self.pred_len, self.k = configs.input_len, configs.pred_len, configs.top_k
        self.conv = nn.Sequential(Inception_Block_V1(configs.d_model, configs.d_ff, num_kernels=configs.num_kernels),
                                  nn.GELU(),
                                  Inception_Block_V1(configs.d_ff, configs.d_model, num_kernels=configs.num_kernels))

    def forward(self, x):
        B, T, N = x.size()
        total = self.seq_len + self.pred_len
        inc1, act, inc2 = self.conv[0], self.conv[1], self.conv[2]
        merged = (x.is_cuda and isinstance(inc1, Inception_Block_V1) and isinstance(inc2, Inception_Block_V1) and isinstance(act, nn.GELU)
                  and getattr(act, "approximate", "none") == "none" and max(len(inc1.kernels), len(inc2.kernels)) <= INCEPTION_MAX)
        Lmax = 2 * total                   # (an image's length < total + period <= 2 total)
        implicit = merged and T == total and inception_periods_ok(inc1, Lmax) and inception_periods_ok(inc2, Lmax)
        if merged and not implicit:      # one averaged kernel per block and step, shared by all periods
            W1, b1, K1 = inception_merge(inc1)
            W2, b2, K2 = inception_merge(inc2)
        if merged and T == total:
            # The periods stay ON THE DEVICE (round 5): the reference reads the top-k on the host (:13-16) because they shape its images --
            # a host sync per TimesBlock and a step no hipGraph can hold.  Here the series is laid out position-major, (Lmax B, N) with
            # row l B + b, so an image of ANY length is a prefix of one static buffer, and the period reaches the kernels as a device
            # number (csrc/conv.hip conv2d_period_*): every host-side shape is static.
            top, weight = fft_for_period_device(x, self.k)
            period, rows = period_rows(top, total, B)
            xl = F.pad(x.transpose(0, 1), (0, 0, 0, 0, 0, Lmax - total)).reshape(Lmax * B, N)
            # the k period images in ONE call per convolution (the same merged kernel for each), then the aggregation + residual as one launch
            if implicit:
                # bf16 mode: no im2col image -- a workgroup holds a (window, period) image in LDS (ops.InceptionPeriodsFn)
                img = inception_periods(xl, period, rows, inc1, B, Lmax, act="gelu")             # (k, Lmax B, d_ff)
                out = inception_periods(img, period, rows, inc2, B, Lmax)                        # (k, Lmax B, N)
            else:
                img = conv2d_periods(xl, period, rows, W1, b1, K1, B, Lmax, act="gelu")
                out = conv2d_periods(img, period, rows, W2, b2, K2, B, Lmax)
            return period_aggregate(out, F.softmax(weight, dim=1), x, B, total, Lmax)
        periods, weight = FFT_for_Period(x, self.k)
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
    immtsf_graphable = True      # (on the GPU: the period selection stays on the device -- TimesBlock.forward -- no host sync in forecasting())

    def immtsf_sink_params(self):
        """the parameters whose gradients the HIP backward writes in place when they are gradient sinks (immtsf.train.FlatTrainer): the
        Inception kernels of every TimesBlock -- what lets a step with a parameter branch take their gradient (an im2col image, the summed
        product, the un-merge) off the backbone's dependent chain (ops.InceptionPeriodsFn)"""
        out = []
        for blk in self.model:
            for inc in (blk.conv[0], blk.conv[2]):
                for c in inc.kernels:
                    out += [c.weight, c.bias]
        return out

    def __init__(self, configs):
        super().__init__()
        self.configs = configs
        self.input_len = self.seq_len = configs.input_len
        self.pred_len = configs.pred_len
        print("seq len:", self.seq_len, self.pred_len)
        self.model = nn.ModuleList([TimesBlock(configs) for _ in range(configs.e_layers)])
        self.enc_embedding = DataEmbedding(2 * configs.enc_in + 1, configs.d_model, configs.embed, configs.freq, configs.dropout)
        self.layer = configs.e_layers
        self.layer_norm = nn.LayerNorm(configs.d_model)
        self.predict_linear = nn.Linear(self.seq_len + self.pred_len, self.pred_len + self.seq_len)
        self.projection = nn.Linear(configs.d_model, configs.c_out, bias=True)
        self.zeros_pad = torch.zeros(configs.batch_size, max(configs.input_len, configs.pred_len), configs.enc_in).to(configs.device)

    def forecasting(self, tp_to_predict, observed_data, observed_tp, observed_mask):
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
