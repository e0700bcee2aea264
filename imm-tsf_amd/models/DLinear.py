"""DLinear backbone (reference models/DLinear.py:7-134): masked instance norm, moving-average decomposition, three
Linear(seq_len -> pred_len) maps on (seasonal, trend, timestamps).  Same signature/state_dict; the three projections
run as ONE grouped launch of the HIP GEMM when shared across channels."""
import torch
import torch.nn as nn

from immtsf.ops import linear
from layers.Autoformer_EncDec import series_decomp
from models._common import masked_instance_norm


class DLinear(nn.Module):
    immtsf_graphable = True      # no host syncs / data-dependent shapes in forecasting()

    def __init__(self, configs, individual=False):
        super().__init__()
        self.input_len = configs.input_len
        self.seq_len = configs.input_len
        self.pred_len = configs.pred_len
        self.individual = individual
        self.C = configs.enc_in
        self.decomposition = series_decomp(configs.moving_avg)

        def make():
            lin = nn.Linear(self.seq_len, self.pred_len)
            lin.weight = nn.Parameter((1 / self.seq_len) * torch.ones_like(lin.weight))
            return lin
        if individual:
            self.Linear_Seasonal = nn.ModuleList([make() for _ in range(self.C)])
            self.Linear_Trend = nn.ModuleList([make() for _ in range(self.C)])
            self.Linear_Time = nn.ModuleList([make() for _ in range(self.C)])
        else:
            self.Linear_Seasonal, self.Linear_Trend, self.Linear_Time = make(), make(), make()
        self.zeros_pad = torch.zeros(configs.batch_size, max(self.seq_len, self.pred_len), self.C, device=configs.device)

    def _project(self, lin, x):                                  # x (B, C, L) -> (B, C, pred_len)
        if self.individual:
            return torch.stack([linear(x[:, i, :], lin[i].weight, lin[i].bias) for i in range(self.C)], dim=1)
        return linear(x, lin.weight, lin.bias)

    def forecasting(self, tp_to_predict, observed_data, observed_tp, observed_mask):
        B, L, C = observed_data.shape
        assert C == self.C
        if L < self.input_len:
            n = self.input_len - L
            observed_data = torch.cat([observed_data, self.zeros_pad[:B, :n, :]], dim=1)
            observed_mask = torch.cat([observed_mask, self.zeros_pad[:B, :n, :]], dim=1)
            observed_tp = torch.cat([observed_tp, self.zeros_pad[:B, :n, 0]], dim=1)
        Lp = tp_to_predict.size(1)
        x, means, stdev = masked_instance_norm(observed_data, observed_mask)
        seasonal, trend = self.decomposition(x)
        time = observed_tp.unsqueeze(1).expand(-1, C, -1)
        dec = (self._project(self.Linear_Seasonal, seasonal.permute(0, 2, 1)) +
               self._project(self.Linear_Trend, trend.permute(0, 2, 1)) +
               self._project(self.Linear_Time, time)).permute(0, 2, 1)
        dec = dec * stdev + means
        return dec[:, :Lp, :]


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
