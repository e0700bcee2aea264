"""Drop-in `lib` package: the pieces of the reference's lib/ that sit on the fusion hot path (loss / train step)."""
from immtsf.dropin import extend_package_path as _extend

_extend(__name__, __path__)     # unmirrored modules of the reference keep resolving (immtsf/dropin.py)
