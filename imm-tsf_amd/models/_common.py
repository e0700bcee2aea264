"""Shared pieces of the irregular-series backbone wrappers (pad to the model length, masked instance norm)."""
import torch


def pad_history(zeros_pad, input_len, pred_len, tp_to_predict, data, tp, mask):
    """pad history to input_len and horizon times to pred_len with the module's zero buffer (B <= args.batch_size)."""
    B, L, _ = data.shape
    if L < input_len:
        n = input_len - L
        data = torch.cat([data, zeros_pad[:B, :n, :]], dim=1)
        mask = torch.cat([mask, zeros_pad[:B, :n, :]], dim=1)
        tp = torch.cat([tp, zeros_pad[:B, :n, 0]], dim=1)
    Lp = tp_to_predict.shape[1]
    if Lp < pred_len:
        tp_to_predict = torch.cat([tp_to_predict, zeros_pad[:B, :pred_len - Lp, 0]], dim=1)
    return tp_to_predict, data, tp, mask, Lp


def plain_instance_norm(x):
    """Non-stationary-Transformer normalisation over time (mean detached, biased variance).  On the GPU, for data (no gradient
    wanted): one launch (immtsf_instance_norm) instead of the expression's six."""
    if x.is_cuda and not x.requires_grad and x.dim() == 3 and x.dtype == torch.float32 and x.shape[1] * x.shape[2] <= 16000:
        from immtsf import _lib
        x = x.contiguous()
        B, L, C = x.shape
        xn = torch.empty_like(x)
        means = torch.empty(B, 1, C, dtype=torch.float32, device=x.device)
        stdev = torch.empty(B, 1, C, dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().immtsf_instance_norm(_lib.ptr(x), B, L, C, _lib.ptr(xn), _lib.ptr(means), _lib.ptr(stdev), _lib.stream_ptr()),
                   "instance_norm")
        return xn, means, stdev
    means = x.mean(1, keepdim=True).detach()
    xc = x - means
    stdev = torch.sqrt(torch.var(xc, dim=1, keepdim=True, unbiased=False) + 1e-5)
    return xc / stdev, means, stdev


def masked_instance_norm(data, mask):
    x = data * mask
    cnt = mask.sum(1, keepdim=True).clamp(min=1)
    means = x.sum(1, keepdim=True) / cnt
    x = x - means
    stdev = torch.sqrt(((x * mask) ** 2).sum(1, keepdim=True) / cnt + 1e-5)
    return x / stdev, means, stdev
