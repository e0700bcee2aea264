"""immtsf -- Python host side of the MI355X-native IMM-TSF fusion hot path.

`immtsf._lib` binds libimmtsf_hip.so (C ABI in include/immtsf.h); `immtsf.ops` wraps the block-level entry points
in torch.autograd.Functions; the drop-in modules live in the sibling packages `fusions/`, `layers/`, `models/`, `lib/`
(same import paths as the reference, so its main.py runs unchanged with this directory first on sys.path).
"""
from . import config  # noqa: F401
