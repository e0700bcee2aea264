"""How the mirrored packages coexist with the reference tree they are dropped in front of.

`imm-tsf_amd/` goes FIRST on sys.path and ships packages with the reference's names (`fusions`, `layers`, `models`,
`lib`).  A regular package would hide the reference's directory of the same name, and with it every module this build
does not mirror (`lib.utils`, `lib.parse_datasets`, `models.Informer`, ...), so:

  * `extend_package_path` (called from each package's __init__) appends the same-named directories found further along
    sys.path to the package's __path__: mirrored modules resolve here, everything else resolves to the reference;
  * `reexport_missing` (called at the end of a mirrored module) loads the shadowed reference module of the same name, if
    one exists, and copies the public names this build does not define (`layers.Embed.DataEmbedding_wo_pos`,
    `layers.Transformer_EncDec.Decoder`, ...), so the reference's other models keep importing what they need.

Both are no-ops when no reference tree is on the path (the tests, the GPU box).
"""
from __future__ import annotations

import importlib.util
import os
import sys


def extend_package_path(name: str, path: list) -> None:
    own = {os.path.realpath(p) for p in path}
    for entry in sys.path:
        cand = os.path.join(entry or ".", *name.split("."))
        if os.path.isdir(cand) and os.path.realpath(cand) not in own:
            path.append(cand)
            own.add(os.path.realpath(cand))


def reexport_missing(module_globals: dict) -> None:
    mod_name, mod_file = module_globals["__name__"], module_globals.get("__file__")
    if not mod_file or "." not in mod_name:
        return
    pkg_name, leaf = mod_name.rsplit(".", 1)
    pkg = sys.modules.get(pkg_name)
    if pkg is None:
        return
    for d in list(getattr(pkg, "__path__", [])):
        cand = os.path.join(d, leaf + ".py")
        if os.path.isfile(cand) and os.path.realpath(cand) != os.path.realpath(mod_file):
            alias = f"{pkg_name}._shadowed_{leaf}"
            if alias in sys.modules:
                other = sys.modules[alias]
            else:
                spec = importlib.util.spec_from_file_location(alias, cand)
                other = importlib.util.module_from_spec(spec)
                sys.modules[alias] = other
                try:
                    spec.loader.exec_module(other)
                except Exception:          # a dependency of the shadowed module is missing: nothing to re-export
                    sys.modules.pop(alias, None)
                    return
            for k, v in vars(other).items():
                if not k.startswith("_") and k not in module_globals:
                    module_globals[k] = v
            return
