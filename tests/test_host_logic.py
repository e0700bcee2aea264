"""Host-side logic of the round-5 additions, on the CPU (no kernel is called): the optimizer shim's routing (reference main.py:1024,
1098-1101), the registries that carry the fused tail's hand-over between the autograd ops, the guard against two writers of one
gradient sink."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "imm-tsf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def test_optim_shim_routes_cpu_and_unusual_arguments_to_torch():
    from immtsf import optim
    p = [torch.nn.Parameter(torch.zeros(3, 2)), torch.nn.Parameter(torch.ones(4))]
    o = optim.Adam(p, lr=1e-2, weight_decay=1e-3)
    assert isinstance(o, optim._torch_adam) and not isinstance(o, optim.FusedAdam)          # CPU parameters: torch's Adam
    for kw in (dict(amsgrad=True), dict(maximize=True), dict(capturable=True), dict(foreach=True)):
        assert isinstance(optim.Adam(p, **kw), optim._torch_adam)
    groups = [{"params": [p[0]]}, {"params": [p[1]], "lr": 1e-4}]
    assert isinstance(optim.Adam(groups), optim._torch_adam)                                  # parameter groups: torch's Adam
    # the clip of parameters no FusedAdam owns is torch's clip: same norm, gradients scaled in place
    for q in p:
        q.grad = torch.full_like(q, 2.0)
    want = torch.linalg.vector_norm(torch.cat([q.grad.reshape(-1) for q in p]))
    got = optim.clip_grad_norm_(p, 1.0)
    assert torch.allclose(got, want)
    assert torch.allclose(torch.linalg.vector_norm(torch.cat([q.grad.reshape(-1) for q in p])), torch.tensor(1.0), atol=1e-5)


def test_optim_shim_install_is_idempotent_and_reversible():
    from immtsf import optim
    a0, c0 = torch.optim.Adam, torch.nn.utils.clip_grad_norm_
    was = optim._installed
    try:
        optim.uninstall()
        assert torch.optim.Adam is optim._torch_adam and torch.nn.utils.clip_grad_norm_ is optim._torch_clip
        optim.install()
        optim.install()
        assert torch.optim.Adam is optim.Adam and torch.nn.utils.clip_grad_norm_ is optim.clip_grad_norm_
        optim.uninstall()
        assert torch.optim.Adam is optim._torch_adam
    finally:
        if was:
            optim.install()
        else:
            torch.optim.Adam, torch.nn.utils.clip_grad_norm_ = a0, c0


def test_handover_registry_matches_storage_shape_and_version():
    from immtsf import ops
    reg = []
    t = torch.zeros(4, 3)
    ops._reg_put(reg, t, "payload")
    assert ops._reg_take(reg, t) == "payload"
    assert ops._reg_take(reg, t.view(3, 4)) is None                  # another shape over the same data: not the tensor that was registered
    assert ops._reg_take(reg, torch.zeros(4, 3)) is None             # other storage
    assert ops._reg_take(reg, None) is None
    assert ops._reg_take(reg, t.t()) is None                         # not contiguous
    t.add_(1.0)                                                      # written since: the entry no longer describes it
    assert ops._reg_take(reg, t) is None
    u = torch.ones(2)
    ops._reg_put(reg, u, 1)
    assert ops._reg_take(reg, u, pop=True) == 1 and ops._reg_take(reg, u) is None
    for i in range(ops._SHADOW_CAP + 3):                            # bounded: the oldest entries go
        ops._reg_put(reg, torch.zeros(1), i)
    assert len(reg) == ops._SHADOW_CAP


def test_two_writers_of_one_undeclared_sink_raise_on_the_host():
    from immtsf import _lib, ops
    p = torch.nn.Parameter(torch.zeros(5))
    q = torch.nn.Parameter(torch.zeros(5))
    ops._claim_sinks([p, None], [None, None], "a")                   # no sink: nothing to guard
    ops._claim_sinks([p], [None], "b")
    p._immtsf_grad_sink = torch.zeros(5)
    ops._claim_sinks([p], [p._immtsf_grad_sink], "a")
    ops._claim_sinks([p], [p._immtsf_grad_sink], "a")                # the same op again (next step): fine
    with pytest.raises(_lib.ImmtsfError, match="sink_shared"):
        ops._claim_sinks([p], [p._immtsf_grad_sink], "b")
    q._immtsf_grad_sink = torch.zeros(5)
    q._immtsf_grad_shared = True                                     # declared shared: every writer accumulates
    ops._claim_sinks([q], [q._immtsf_grad_sink], "a")
    ops._claim_sinks([q], [q._immtsf_grad_sink], "b")


def test_t2v_form_selection_is_a_host_decision():
    """immtsf_ttf_t2v_xattn_folded (no kernel, no GPU): which formulation of TTF_T2V_XAttn a cfg gets -- the folded form from 8192 padded
    note rows on inside its limits (N <= 64), its mix-first variant for longer windows in bf16 mode with one head (or form 3), the chain
    as written otherwise (form 1, fp32 with long windows, several heads with long windows).  include/immtsf.h immtsf_fusion_cfg.form."""
    import ctypes as C
    from immtsf import _lib
    from immtsf.ops import make_cfg
    lib = _lib.load()

    def folded(B, N, T, d_m, d, H, bf16, form):
        cfg = make_cfg(B, N, T, 0, d_m, d, H, 1 if bf16 else 0, True, 0.1, 0.0, 0, None)
        cfg.form = form
        return lib.immtsf_ttf_t2v_xattn_folded(C.byref(cfg))

    assert folded(64, 32, 32, 768, 768, 1, True, 0) == 0            # 2048 padded rows: below the fold's fixed cost
    assert folded(256, 32, 32, 768, 768, 1, True, 0) == 1           # 8192 rows: folded
    assert folded(64, 32, 32, 768, 768, 1, True, 2) == 1
    assert folded(256, 32, 32, 768, 768, 1, True, 1) == 0           # the chain on request
    assert folded(64, 4096, 32, 4096, 768, 1, True, 0) == 1         # cfg5: long windows, bf16, one head -> mix-first
    assert folded(64, 4096, 32, 4096, 768, 1, False, 0) == 0        # fp32 mode: the chain
    assert folded(64, 4096, 32, 4096, 768, 2, True, 0) == 0         # two heads: the chain
    assert folded(64, 4096, 33, 4096, 768, 1, True, 3) == 0         # T > 32: outside both folded forms
    assert folded(4, 16, 8, 64, 32, 1, True, 3) == 1                # form 3 at any N inside its limits
    assert folded(4, 16, 8, 64, 32, 1, True, 1) == 0
