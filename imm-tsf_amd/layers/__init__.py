"""Drop-in `layers` package: the attention / embedding layers the configured backbones use,
same class names, constructor and forward signatures and state_dict keys as the reference, contractions on the HIP
MFMA GEMM."""
from immtsf.dropin import extend_package_path as _extend

_extend(__name__, __path__)     # unmirrored modules of the reference keep resolving (immtsf/dropin.py)
