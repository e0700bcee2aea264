"""Drop-in `models` package (same import paths and Model(args).forecasting(...) signatures as the reference)."""
from immtsf.dropin import extend_package_path as _extend

_extend(__name__, __path__)     # unmirrored modules of the reference keep resolving (immtsf/dropin.py)
