"""Shared host-side plumbing of the fusion modules."""
import torch

from immtsf import config


def prep_t_hat(t_hat: torch.Tensor, B: int) -> torch.Tensor:
    """(T,) -> (B,T); anything whose first dim is not B is an error (same contract as the reference)."""
    if t_hat.dim() == 1:
        return t_hat.unsqueeze(0).repeat(B, 1)
    if t_hat.shape[0] != B:
        raise ValueError(f"Expected t_hat shape (B, T_f) or (T_f,), got {t_hat.shape}")
    return t_hat


class NanFlag:
    """device int32 that kernels OR with 1 when they meet a NaN in the note embeddings."""

    def __init__(self):
        self.flag = None

    def get(self, device):
        if self.flag is None or self.flag.device != device:
            self.flag = torch.zeros(1, dtype=torch.int32, device=device)
        return self.flag

    def raise_if_set(self, message: str):
        if self.flag is not None and int(self.flag.item()) != 0:
            self.flag.zero_()
            raise ValueError(message)


def resolve_precision(module) -> int:
    return config.precision_code(getattr(module, "precision", None) or config.precision)


def f32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.float32 else t.float()
