"""Drop-in `fusions` package: same module/class names, constructor and forward signatures and state_dict keys as
the reference's fusions/ directory, computed by the HIP kernels in libimmtsf_hip.so."""
from immtsf.dropin import extend_package_path as _extend

_extend(__name__, __path__)     # unmirrored modules of the reference keep resolving (immtsf/dropin.py)
