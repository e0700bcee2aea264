"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads without a GPU and exports every symbol
include/immtsf.h declares; the Python binding lists exactly those; the product modules refuse to run on the CPU."""
import ctypes
import os
import re
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "immtsf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(immtsf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from immtsf import _lib
    names = _declared()
    assert len(names) >= 25
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/immtsf.h but not exported"
    assert sorted(_lib.exported_names()) == names, "ctypes prototypes and header disagree"
    assert _lib.load().immtsf_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_the_library():
    """every ABI struct of the ctypes binding has the size the library was compiled with (immtsf_abi_sizes), and the config struct's
    fields are, name by name and in order, the members include/immtsf.h declares -- a field added on one side only cannot drift in
    silently (round-4 review: INTEGRATION.md documented a struct two fields short)"""
    from immtsf import _lib
    lib = _lib.load()
    structs = _lib.abi_structs()
    out = (ctypes.c_int32 * 32)()
    n = lib.immtsf_abi_sizes(out, 32)
    assert n == len(structs)
    for i, t in enumerate(structs):
        assert int(out[i]) == ctypes.sizeof(t), (t.__name__, int(out[i]), ctypes.sizeof(t))
    src = open(os.path.join(ROOT, "include", "immtsf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = re.search(r"typedef struct immtsf_fusion_cfg \{(.*?)\} immtsf_fusion_cfg;", src, flags=re.S).group(1)
    members = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            members += [re.sub(r"[^A-Za-z0-9_]", "", x.split()[-1]) for x in decl.split(",")]
    assert members == [f for f, _ in _lib.FusionCfg._fields_]
    # INTEGRATION.md shows the same struct to a maintainer binding the library by hand
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for f, _ in _lib.FusionCfg._fields_:
        assert f'"{f}"' in doc, f"INTEGRATION.md's Cfg example lacks {f}"
    assert f"IMMTSF_ABI_VERSION` is {_lib.ABI_VERSION}" in doc


def test_workspace_queries_run_without_gpu():
    from immtsf import _lib
    lib = _lib.load()
    cfg = _lib.FusionCfg(64, 32, 32, 8, 768, 768, 1, 0, 1, 0.1, 0.5, 1)
    for fn in ("immtsf_ttf_t2v_xattn_workspace_bytes", "immtsf_ttf_t2v_xattn_scratch_bytes",
               "immtsf_ttf_recavg_workspace_bytes", "immtsf_ttf_recavg_scratch_bytes",
               "immtsf_mmf_xattn_add_workspace_bytes", "immtsf_mmf_xattn_add_scratch_bytes"):
        n = getattr(lib, fn)(ctypes.byref(cfg))
        assert 1 << 20 < n < 1 << 32, (fn, n)
    assert lib.immtsf_mmf_gr_add_workspace_bytes(ctypes.byref(cfg), 8) > 0
    bad = _lib.FusionCfg(64, 32, 32, 8, 768, 770, 4, 0, 1, 0.1, 0.5, 1)     # d % H != 0
    assert lib.immtsf_ttf_t2v_xattn_workspace_bytes(ctypes.byref(bad)) == 0


def test_modules_keep_reference_state_dict_keys_and_refuse_cpu():
    import numpy as np
    from fusions.FusionModel import FusionModel, _MMF_CLASSES, _TTF_CLASSES
    from fusions.load_llm import register_d_model
    from immtsf._lib import ImmtsfError
    register_d_model("TOY16", 16)
    for ttf in _TTF_CLASSES:
        for mmf in _MMF_CLASSES:
            a = types.SimpleNamespace(TTF_module=ttf, MMF_module=mmf, llm_model_fusion="TOY16", llm_layers_fusion=6,
                                      max_length=1024, device="cpu", use_text_embeddings=True, recency_sigma=1.0,
                                      n_heads_fusion=2, dropout=0.1, d_txt=8, C=3, kappa=0.5)
            m = FusionModel(a)
            z = np.load(os.path.join(ROOT, "tests", "golden", f"fusion_{ttf}_{mmf}_tiny_h2.npz"))
            ref = {k[2:]: z[k].shape for k in z.files if k.startswith("p.")}
            assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == ref
            with pytest.raises(ImmtsfError):
                m(torch.randn(2, 3, 16), torch.rand(2, 3), torch.rand(2, 4), torch.randn(2, 4, 3))
    # class objects are accepted in place of registry strings (fusions/FusionModel.py:45-50)
    a.TTF_module, a.MMF_module = _TTF_CLASSES["TTF_RecAvg"], _MMF_CLASSES["MMF_GR_Add"]
    assert isinstance(FusionModel(a).ttf, _TTF_CLASSES["TTF_RecAvg"])


def test_philox_host_matches_documented_vector():
    """Known-answer test for Philox4x32-10 (Random123 kat_vectors: counter=0,key=0)."""
    def mulhi(a, b):
        return ((a * b) >> 32) & 0xFFFFFFFF

    def philox(seed, site, ctr):
        k0, k1 = seed & 0xFFFFFFFF, seed >> 32
        c = [ctr & 0xFFFFFFFF, ctr >> 32, site & 0xFFFFFFFF, site >> 32]
        for _ in range(10):
            hi0, lo0 = mulhi(0xD2511F53, c[0]), (0xD2511F53 * c[0]) & 0xFFFFFFFF
            hi1, lo1 = mulhi(0xCD9E8D57, c[2]), (0xCD9E8D57 * c[2]) & 0xFFFFFFFF
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
            k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
        return c
    assert philox(0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
