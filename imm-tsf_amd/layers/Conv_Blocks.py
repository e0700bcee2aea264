"""Inception block of TimesNet (reference layers/Conv_Blocks.py:5-31): mean of `num_kernels` same-padded 2-D convs.
models/TimesNet.py evaluates it as ONE merged convolution on the HIP GEMM (immtsf.ops.inception_merge + conv2d_same_cl);
this module keeps the reference's parameters / state_dict keys and its op-by-op forward for callers that use it directly."""
import torch
import torch.nn as nn


class Inception_Block_V1(nn.Module):
    def __init__(self, in_channels, out_channels, num_kernels=6, init_weight=True):
        super().__init__()
        self.in_channels, self.out_channels, self.num_kernels = in_channels, out_channels, num_kernels
        self.kernels = nn.ModuleList(
            [nn.Conv2d(in_channels, out_channels, kernel_size=2 * i + 1, padding=i) for i in range(num_kernels)])
        if init_weight:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0)

    def forward(self, x):
        return torch.stack([conv(x) for conv in self.kernels], dim=-1).mean(-1)


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror
