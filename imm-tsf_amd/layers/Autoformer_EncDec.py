"""Moving-average series decomposition used by DLinear (reference layers/Autoformer_EncDec.py:21-52). Parameter-free;
stock torch pooling (not on the fusion hot path)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class moving_avg(nn.Module):
    def __init__(self, kernel_size, stride):
        super().__init__()
        self.kernel_size = kernel_size
        self.stride = stride

    def forward(self, x):                      # (B, L, C): replicate-pad both ends, average over the window
        half = (self.kernel_size - 1) // 2
        xp = torch.cat([x[:, :1].expand(-1, half, -1), x, x[:, -1:].expand(-1, half, -1)], dim=1)
        return F.avg_pool1d(xp.permute(0, 2, 1), self.kernel_size, self.stride).permute(0, 2, 1)


class series_decomp(nn.Module):
    def __init__(self, kernel_size):
        super().__init__()
        self.moving_avg = moving_avg(kernel_size, stride=1)

    def forward(self, x):
        trend = self.moving_avg(x)
        return x - trend, trend


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
