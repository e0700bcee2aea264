"""`torch.optim.Adam` + `clip_grad_norm_` of an UNEDITED main.py on the fused clip + Adam kernels (reference: main.py:1024 builds
`optim.Adam(trainable_parameters, lr=args.lr, weight_decay=args.w_decay)`, main.py:1098-1101 runs `clip_grad_norm_(.., max_norm=1.0)`
and `optimizer.step()` every step: ~25 eager launches -- per-tensor norms, a foreach chain per moment -- which is what kept the zero-edit
seam at 1.1 ms when forward + loss + backward had become one graph replay).

`Adam(params, lr, betas, eps, weight_decay, ...)` has torch's constructor.  When every parameter is a dense fp32 CUDA tensor in ONE
group with the default flags it returns `FusedAdam`: the parameters are re-homed as views of one flat buffer (state_dict keys / shapes
unchanged, like immtsf.train.FlatTrainer), `.grad` of every parameter is a view of one flat gradient buffer, and `step()` is two
launches (immtsf_adam_prepare + immtsf_adam_range: clip by the global norm, Adam, the gradient left zero for the next step).  Anything
else gets the real `torch.optim.Adam`.  `clip_grad_norm_(parameters, max_norm)` on exactly a FusedAdam's parameters returns the norm and
leaves the clipping to that optimizer's next `step()`; any other call goes to torch's.

`install()` puts the two in torch's namespaces (`torch.optim.Adam`, `torch.nn.utils.clip_grad_norm_`) so that an unmodified script gets
them; the drop-in `lib.evaluation` calls it on import unless IMMTSF_OPTIM_SHIM=0.  Same arithmetic as torch.optim.Adam (L2-style
weight decay, bias corrections, eps outside the square root) -- tests/test_gpu_train.py::test_optim_shim_trains_like_torch_adam."""
from __future__ import annotations

import os

import torch

from . import _lib

_torch_adam = torch.optim.Adam
_torch_clip = torch.nn.utils.clip_grad_norm_


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        ps = self.param_groups[0]["params"]
        dev = ps[0].device
        pad8 = lambda k: (k + 7) // 8 * 8      # noqa: E731
        n = sum(pad8(p.numel()) for p in ps)
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.norm_scratch = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self._views, self._gviews, off = [], [], 0
        for p in ps:
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat_param[off:off + k].view(p.shape)
            g = self.flat_grad[off:off + k].view(p.shape)
            if p.grad is not None:
                g.copy_(p.grad)
            p.grad = g
            p._immtsf_optim = self
            self._views.append((off, k))
            self._gviews.append(g)
            # torch's per-parameter state layout, as views of the flat buffers: state_dict() / load_state_dict() keep torch's format
            self.state[p] = {"step": self.step_dev, "exp_avg": self.exp_avg[off:off + k].view(p.shape),
                             "exp_avg_sq": self.exp_avg_sq[off:off + k].view(p.shape)}
            off += pad8(k)
        self._ids = tuple(id(p) for p in ps)
        self._pending_clip = 0.0
        self._zeroed = True

    # ------------------------------------------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True):
        """the gradients stay views of the flat buffer (never None: autograd and the seam graph then accumulate in place); the fused
        step leaves the buffer zero, so the usual zero_grad() at the top of the loop costs nothing"""
        ps = self.param_groups[0]["params"]
        for p, g in zip(ps, self._gviews):
            if p.grad is not g:                 # somebody replaced or dropped it (set_to_none by hand, a foreign tensor): re-home it
                if p.grad is not None:
                    g.copy_(p.grad)
                    self._zeroed = False
                p.grad = g
        if not self._zeroed:
            self.flat_grad.zero_()
        self._zeroed = False                    # (a backward may write from here on)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        ps = g["params"]
        for p, gv in zip(ps, self._gviews):     # a gradient that autograd (or the caller) put beside the flat buffer: bring it home
            if p.grad is not None and p.grad is not gv and p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
                p.grad = gv
        lib = _lib.load()
        n = self.flat_param.numel()
        st = _lib.stream_ptr()
        _lib.check(lib.immtsf_adam_prepare(_lib.ptr(self.flat_grad), None, n, _lib.ptr(self.norm_scratch), _lib.ptr(self.step_dev), None,
                                           None, None, None, None, None, st), "adam_prepare")
        _lib.check(lib.immtsf_adam_range(_lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), None, _lib.ptr(self.exp_avg),
                                         _lib.ptr(self.exp_avg_sq), n, 0, n, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                         float(g["eps"]), float(g["weight_decay"]), _lib.ptr(self.step_dev), float(self._pending_clip),
                                         _lib.ptr(self.norm_scratch), 1, None, st), "adam_range")
        self._pending_clip = 0.0
        self._zeroed = True
        return loss

    def owns(self, parameters) -> bool:
        return tuple(id(p) for p in parameters) == self._ids

    def state_dict(self):
        """torch.optim.Adam's format: per parameter `step` (a float32 scalar on the host, what torch's Adam keeps), `exp_avg`, `exp_avg_sq`"""
        sd = super().state_dict()
        step = torch.tensor(float(self.step_dev.item()))
        sd["state"] = {k: {**v, "step": step.clone()} for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # torch put fresh tensors into self.state: copy them into the flat buffers and point the state back at the views
        ps = self.param_groups[0]["params"]
        step = None
        for p, (off, k) in zip(ps, self._views):
            st = self.state.get(p, {})
            if "exp_avg" in st:
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            if "step" in st and step is None:
                step = int(float(st["step"]))
            self.state[p] = {"step": self.step_dev, "exp_avg": self.exp_avg[off:off + k].view(p.shape),
                             "exp_avg_sq": self.exp_avg_sq[off:off + k].view(p.shape)}
        if step is not None:
            self.step_dev.fill_(step)


def Adam(params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, foreach=None, maximize=False,
         capturable=False, differentiable=False, fused=None, **kw):
    """torch.optim.Adam's constructor: FusedAdam where it applies (see the module docstring), torch's Adam otherwise"""
    params = list(params)
    plain = (params and all(torch.is_tensor(p) for p in params) and not amsgrad and not maximize and not capturable and not differentiable
             and not fused and not kw and not torch.is_tensor(lr) and
             all(p.is_cuda and p.dtype == torch.float32 and p.layout == torch.strided and p.requires_grad and
                 not hasattr(p, "_immtsf_bucket") and not hasattr(p, "_immtsf_optim") for p in params) and
             len({id(p) for p in params}) == len(params) and len({p.device for p in params}) == 1)
    if plain:
        try:
            _lib.load()
        except _lib.ImmtsfError:
            plain = False
    if plain:
        return FusedAdam(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
    return _torch_adam(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, foreach=foreach, maximize=maximize,
                       capturable=capturable, differentiable=differentiable, fused=fused, **kw)


def clip_grad_norm_(parameters, max_norm, norm_type=2.0, error_if_nonfinite=False, foreach=None):
    """torch.nn.utils.clip_grad_norm_: on exactly a FusedAdam's parameters (2-norm) the norm is returned and the clipping itself rides in
    that optimizer's next step() (immtsf_adam_range clips by the global norm it is given); anything else is torch's"""
    if torch.is_tensor(parameters):
        parameters = [parameters]
    parameters = list(parameters)
    opt = getattr(parameters[0], "_immtsf_optim", None) if parameters else None
    if opt is None or float(norm_type) != 2.0 or error_if_nonfinite or not opt.owns(parameters):
        return _torch_clip(parameters, max_norm, norm_type=norm_type, error_if_nonfinite=error_if_nonfinite, foreach=foreach)
    for p, gv in zip(parameters, opt._gviews):
        if p.grad is not None and p.grad is not gv and p.grad.data_ptr() != gv.data_ptr():
            gv.copy_(p.grad)
            p.grad = gv
    opt._pending_clip = float(max_norm)
    return torch.linalg.vector_norm(opt.flat_grad)


_installed = False


def install():
    """route torch.optim.Adam and torch.nn.utils.clip_grad_norm_ through this module (idempotent)"""
    global _installed
    if _installed:
        return
    torch.optim.Adam = Adam
    torch.nn.utils.clip_grad_norm_ = clip_grad_norm_
    if hasattr(torch.nn.utils, "clip_grad"):
        torch.nn.utils.clip_grad.clip_grad_norm_ = clip_grad_norm_
    _installed = True


def uninstall():
    global _installed
    torch.optim.Adam = _torch_adam
    torch.nn.utils.clip_grad_norm_ = _torch_clip
    if hasattr(torch.nn.utils, "clip_grad"):
        torch.nn.utils.clip_grad.clip_grad_norm_ = _torch_clip
    _installed = False


def install_from_env():
    if os.environ.get("IMMTSF_OPTIM_SHIM", "1") != "0" and torch.cuda.is_available():
        install()
