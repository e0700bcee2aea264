"""Data-parallel training step plumbing for the fusion hot path (new: the reference is single-process).

`FlatTrainer` owns one flat fp32 buffer for all parameters, one for all gradients and two for Adam's moments:
  * parameters become views into the flat buffer (state_dict keys/shapes unchanged);
  * the HIP backward of every fusion block writes its parameter gradients straight into the flat gradient
    buffer (`_immtsf_grad_sink`), backbone gradients are accumulated there by autograd (`p.grad` is a view);
  * entities shard across ranks (pure data parallel); the only collective is the gradient all-reduce (RCCL
    over xGMI via torch.distributed, backend "nccl"; gloo in the CPU tests), issued per BUCKET on a side stream
    as soon as a bucket's gradients are final -- MMF bucket while the TTF backward still runs, TTF bucket while
    the backbone backward runs -- and joined before the optimizer step;
  * clip_grad_norm_(max_norm) + Adam (main.py:1024,1098-1101) is one fused HIP kernel pair over the flat buffers.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional, Sequence

import os

import torch

from . import _lib, ops


def shard_range(n_items: int, rank: int, world: int):
    """contiguous entity shard of rank `rank`: [lo, hi) (rank r gets entities r*(E/W) .. (r+1)*(E/W)-1)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


class FlatTrainer:
    def __init__(self, buckets: Sequence[Iterable[torch.nn.Parameter]], lr: float = 1e-3, weight_decay: float = 0.0,
                 betas=(0.9, 0.999), eps: float = 1e-8, max_norm: float = 1.0, group=None, overlap: bool = True,
                 sink_buckets: Sequence[int] = (), device_step: bool = False, bf16_twin: Optional[bool] = None,
                 grad_wire: str = "fp32", sink_exclude: Iterable[torch.nn.Parameter] = (), shard_optimizer: bool = False,
                 param_wire: str = "fp32", sink_shared: Iterable[torch.nn.Parameter] = ()):
        """buckets: parameter groups in the order their gradients become final during backward (first = earliest).
        sink_buckets: indices of buckets whose gradients are written by the HIP backward directly (fusion blocks; the
        backbone too when every op that uses its parameters is an immtsf.ops function).  A sink parameter must be used by
        exactly ONE such op per step (the op overwrites, or atomically accumulates onto, the zero-filled slice);
        sink_exclude: parameters of sink buckets that are used more than once or by stock torch ops (e.g. tPatchGNN's
        time-embedding weights, shared by the patch encoder and the decoder's time features) -- autograd accumulates those
        as usual and they are copied into the flat buffer after the backward.
        sink_shared: parameters of sink buckets used by SEVERAL immtsf.ops functions that support accumulation
        (ops.time2vec and ops.ttcn_patch_encode: tPatchGNN's four time-embedding parameters): every user ADDS its
        gradient into the zero-filled slice, so no autograd accumulation kernels run for them either.
        grad_wire: "fp32" (exact: the sum of the ranks' gradients) or "bf16" (the gradients are rounded to bf16 for the
        all-reduce and widened again: half the bytes on xGMI -- the collective is per-link bandwidth bound -- at the
        cost of ~3 significant digits per element, which clip + Adam's normalised update tolerates)."""
        if grad_wire not in ("fp32", "bf16"):
            raise ValueError("grad_wire must be 'fp32' or 'bf16'")
        if param_wire not in ("fp32", "bf16"):
            raise ValueError("param_wire must be 'fp32' or 'bf16'")
        self.grad_wire = grad_wire
        self.group = group
        self.world, self.rank = 1, 0
        if group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        # shard_optimizer (ZeRO-1 style; world > 1 or a 1-rank group): the flat gradient is REDUCE-SCATTERED (each rank
        # receives the sum of its 1/W slice: half the bytes of an all-reduce), clip + Adam run on that slice only -- the
        # moments exist only there: 1/W of the optimizer state and of its 28 bytes/parameter of HBM traffic -- and the
        # updated parameters are ALL-GATHERED.  param_wire "fp32": the gathered parameters are the exact fp32 values (what a
        # single process would hold; 4 bytes/parameter on the wire).  "bf16": only the bf16 image is gathered (2
        # bytes/parameter: the whole step then moves what one bf16 all-reduce moves) and every rank's replicated fp32
        # parameters are its widening -- mixed-precision semantics: the fp32 master copy lives with the owner's shard.
        self.sharded = bool(shard_optimizer and group is not None)
        self.param_wire = param_wire
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        self.step_count = 0
        seen, self.buckets = set(), []
        for b in buckets:
            ps = [p for p in b if p.requires_grad and id(p) not in seen]
            seen.update(id(p) for p in ps)
            self.buckets.append(ps)
        params = [p for b in self.buckets for p in b]
        if not params:
            raise ValueError("no trainable parameters")
        dev = params[0].device
        # every parameter starts on a 32-byte boundary of the flat buffers, so that it (and its bf16 twin) can be fetched
        # with 16-byte loads; the padding elements stay zero everywhere
        pad8 = lambda k: (k + 7) // 8 * 8      # noqa: E731
        n = sum(pad8(p.numel()) for p in params)
        if self.sharded:
            q = 8 * self.world
            n = (n + q - 1) // q * q              # equal, 32-byte aligned shards
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
        # (8 more elements behind the payload: the GUARD SLOT of a data-parallel FlagStep -- each rank's time-out word, summed over the
        # ranks by the step's last collective, so that every rank drops the same step)
        self._grad_store = torch.zeros(n + 8, dtype=torch.float32, device=dev)
        self.flat_grad = self._grad_store[:n]
        per = n // self.world if self.sharded else n
        self.shard = (self.rank * per, (self.rank + 1) * per) if self.sharded else (0, n)
        self.exp_avg = torch.zeros(per, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(per, dtype=torch.float32, device=dev)
        self._wire_shard = torch.empty(per, dtype=torch.bfloat16, device=dev) if self.sharded and grad_wire == "bf16" else None
        self._grad_shard = torch.empty(per, dtype=torch.float32, device=dev) if self.sharded and grad_wire == "fp32" else None
        self.master = None               # param_wire "bf16": fp32 master copy of this rank's parameter shard (set below)
        self.norm_scratch = torch.zeros(1024, dtype=torch.float32, device=dev)
        self._wire_store = torch.zeros(n + 8, dtype=torch.bfloat16, device=dev) if grad_wire == "bf16" and group is not None else None
        self._wire = None if self._wire_store is None else self._wire_store[:n]
        self.ranges = []
        self._views = []
        self._collected = True
        excl = {id(p) for p in sink_exclude}
        shared = {id(p) for p in sink_shared}
        self._autograd_owned = []        # (parameter, flat gradient view) pairs autograd accumulates itself
        off = 0
        for bi, b in enumerate(self.buckets):
            start = off
            views = []
            for p in b:
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + k].view(p.shape)
                gview = self.flat_grad[off:off + k].view(p.shape)
                views.append(gview)
                if hasattr(p, "_immtsf_sink_writers"):      # (immtsf.ops._claim_sinks: the writers of an earlier trainer's sink)
                    del p._immtsf_sink_writers
                if bi in sink_buckets and id(p) not in excl:
                    p._immtsf_grad_sink = gview
                    p._immtsf_grad_prezeroed = True   # zero_grad() memsets the whole flat buffer every step
                    p._immtsf_grad_shared = id(p) in shared
                else:
                    self._autograd_owned.append((p, gview))
                p.grad = gview              # optimizers / clip utilities that look at .grad still work
                p._immtsf_bucket = (id(self), bi)      # (ops with several backward phases fire a bucket's hook behind the phase that completes it)
                off += pad8(k)
            self._views.append(views)
            self.ranges.append((start, off))
        self.sink_buckets = set(sink_buckets)
        # bf16 twin of the parameters: the forward / data-gradient GEMMs read their weight operand from it in bf16 mode
        # (half the operand bytes, no conversion); the fused Adam kernel writes it together with the parameters
        self.flat_twin = None
        if (bf16_twin if bf16_twin is not None else dev.type == "cuda") and dev.type == "cuda":
            lib = _lib.load()
            self.flat_twin = torch.empty(n, dtype=torch.bfloat16, device=dev)
            _lib.check(lib.immtsf_bf16_twin_register(_lib.ptr(self.flat_param), _lib.ptr(self.flat_twin), n), "bf16_twin_register")
            self.refresh_twins()
        # device_step: Adam's step number (and the dropout key offset) live in device memory so that a captured
        # hipGraph advances them on every replay (immtsf.config.enable_device_counters)
        self.device_step = device_step and dev.type == "cuda"
        if self.device_step:
            from . import config
            _, self.drop_dev = config.enable_device_counters(dev)      # dropout key counter: shared by the device's modules
            self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)    # Adam's step number: this trainer's own
        if self.sharded and param_wire == "bf16":
            if self.flat_twin is None:
                self.flat_twin = torch.empty(n, dtype=torch.bfloat16, device=dev)
                self.flat_twin.copy_(self.flat_param)
            self.master = self.flat_param[self.shard[0]:self.shard[1]].clone()
            self.flat_param.copy_(self.flat_twin)       # the replicated parameters are the widened bf16 image on every rank
        self.collective = group is not None          # a 1-rank group still runs the (trivial) collectives: lets one GPU
        # (the sharded optimizer reduce-scatters the whole flat gradient in sync_grads: per-bucket all-reduces from the backward
        # hooks would sum the sink buckets twice and race with it on the communication stream)
        self.overlap = overlap and self.collective and dev.type == "cuda" and not self.sharded
        self.comm_stream = torch.cuda.Stream(device=dev) if self.overlap else None
        self._pending: List = []
        self._reduced = [False] * len(self.buckets)
        if self.collective and dev.type == "cuda" and not self.sharded:
            for bi in self.sink_buckets:
                head = self.buckets[bi][0]
                hook = (lambda burst=None, i=bi: self._on_bucket_done(i, burst))
                hook._immtsf_bucket_index = bi
                head._immtsf_bwd_hook = hook

    def gather(self, flat: torch.Tensor) -> torch.Tensor:
        """the parameters' elements of a flat buffer (param / grad / moment), concatenated in bucket order without the
        alignment padding"""
        out, off = [], 0
        for b in self.buckets:
            for p in b:
                out.append(flat[off:off + p.numel()])
                off += (p.numel() + 7) // 8 * 8
        return torch.cat(out)

    def refresh_twins(self):
        """re-derive the bf16 twin (and, with a bf16 parameter wire, the owner's fp32 master shard) from the fp32
        parameters: needed after writing parameters by any means other than step() -- load_state_dict, a torch.optim
        optimizer, manual edits.  `watch(modules)` arranges for it to run by itself after load_state_dict."""
        if self.master is not None:
            self.master.copy_(self.flat_param[self.shard[0]:self.shard[1]])
        if self.flat_twin is not None:
            if self.flat_param.is_cuda:
                lib = _lib.load()
                _lib.check(lib.immtsf_f32_to_bf16(_lib.ptr(self.flat_param), _lib.ptr(self.flat_twin), self.flat_param.numel(),
                                                  _lib.stream_ptr()), "f32_to_bf16")
            else:
                self.flat_twin.copy_(self.flat_param)
            if self.master is not None:
                self.flat_param.copy_(self.flat_twin)

    after_external_update = refresh_twins

    def watch(self, *modules):
        """keep the bf16 twin current across `module.load_state_dict(...)` (checkpoint resume, best-model reload before
        the test pass of the reference's main.py flow): a post-hook on each given module refreshes the twin.  In bf16 mode
        the forward / data-gradient GEMMs read their weights from the twin, so a stale twin would silently evaluate the old
        weights.  Parameter writes that bypass load_state_dict (a torch.optim step on these parameters, manual edits) still
        need an explicit after_external_update()."""
        for m in modules:
            m.register_load_state_dict_post_hook(lambda module, incompatible_keys: self.refresh_twins())
        return self

    def close(self):
        """drop the twin registration (the registry is keyed by the flat buffer's address)"""
        if getattr(self, "flat_twin", None) is not None:
            try:
                _lib.load().immtsf_bf16_twin_unregister(_lib.ptr(self.flat_param))
            except Exception:
                pass
            self.flat_twin = None

    def __del__(self):
        self.close()

    # ---------------------------------------------------------------------------------------------- step pieces
    def zero_grad(self):
        """ONE memset of the whole flat gradient buffer.  The HIP backward of the sink buckets then overwrites (or, for
        split-K weight gradients, atomically accumulates onto the zeros: `grads_prezeroed`) its ranges; autograd-owned
        parameters get `.grad = None` so that AccumulateGrad adopts the incoming tensor instead of launching one add
        kernel per parameter -- `_collect_autograd_grads` copies them into the flat buffer with one multi-tensor copy."""
        # zero_in_step (GraphedStep's one-graph mode turns it on): the previous step()'s Adam pass already left the buffer zero
        if not (getattr(self, "zero_in_step", False) and getattr(self, "_grad_zeroed_by_step", False)):
            self.flat_grad.zero_()
        self._grad_zeroed_by_step = False
        for p, _ in self._autograd_owned:
            p.grad = None
        self._reduced = [False] * len(self.buckets)
        self._collected = False

    def _collect_autograd_grads(self):
        if self._collected:
            return
        self._collected = True
        dst, src = [], []
        for p, v in self._autograd_owned:
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad.reshape(v.shape))
        if dst:
            torch._foreach_copy_(dst, src)

    def collect_grads(self):
        """copy the autograd-owned gradients (backbone) into the flat buffer NOW.  Call right after backward() when the
        step is being captured into a hipGraph, so that the copy is part of the graph and runs on every replay."""
        self._collected = False
        self._collect_autograd_grads()

    def _on_bucket_done(self, bi: int, burst=None):
        """backward hook of a sink bucket: start its all-reduce now -- only while overlapping is on.  (A caller that turns
        `overlap` off after construction, e.g. GraphedStep without captured collectives, must not get collectives issued
        from inside the backward: under graph capture they would be captured AND repeated eagerly afterwards.)"""
        cap = getattr(self, "_capture_hook", None)
        if cap is not None:          # FlagStep's capture of the data-parallel step: announce the bucket by a device flag instead
            cap(bi, burst)
            return
        if self.overlap:
            self._bucket_ready(bi)

    def _bucket_ready(self, bi: int):
        if not self.collective or self.sharded or self._reduced[bi]:
            return
        import torch.distributed as dist
        if bi not in self.sink_buckets:
            self._collect_autograd_grads()
        lo, hi = self.ranges[bi]
        if hi == lo:
            self._reduced[bi] = True
            return
        if self.overlap:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self._all_reduce(lo, hi)
        else:
            self._all_reduce(lo, hi)
        self._reduced[bi] = True

    def _all_reduce(self, lo: int, hi: int):
        """sum flat_grad[lo:hi] over the group on the current stream, through the chosen wire format"""
        import torch.distributed as dist
        g = self.flat_grad[lo:hi]
        if self.grad_wire == "fp32":
            dist.all_reduce(g, group=self.group)
            return
        w = self._wire[lo:hi]
        if g.is_cuda:                   # 16-byte conversion kernels (round to nearest even / exact widening)
            lib = _lib.load()
            _lib.check(lib.immtsf_f32_to_bf16(_lib.ptr(g), _lib.ptr(w), g.numel(), _lib.stream_ptr()), "f32_to_bf16")
            dist.all_reduce(w, group=self.group)
            _lib.check(lib.immtsf_bf16_to_f32(_lib.ptr(w), _lib.ptr(g), g.numel(), _lib.stream_ptr()), "bf16_to_f32")
        else:
            w.copy_(g)
            dist.all_reduce(w, group=self.group)
            g.copy_(w)

    def sync_grads(self):
        """all-reduce (sum) whatever has not been reduced yet and join the communication stream.  The loss is
        normalised over the GLOBAL batch (immtsf.ops.masked_mse with `group`), so the sum of the ranks' gradients
        is exactly the single-process full-batch gradient."""
        self._collect_autograd_grads()
        if self.sharded:
            if self.collective and not all(self._reduced):      # collective off (a rank-local measurement leg): no communication
                self._reduce_scatter()
                self._reduced = [True] * len(self.buckets)
            if self.comm_stream is not None:                    # (nothing is enqueued there in sharded mode; joined all the same)
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            return
        if self.collective:
            if not self.overlap and not any(self._reduced):
                import torch.distributed as dist
                self._all_reduce(0, self.flat_grad.numel())            # one collective for the whole flat buffer
                self._reduced = [True] * len(self.buckets)
            for bi in range(len(self.buckets)):
                self._bucket_ready(bi)
            if self.overlap:
                torch.cuda.current_stream().wait_stream(self.comm_stream)

    def _reduce_scatter(self):
        """flat_grad[shard] <- sum over ranks of flat_grad[shard] (RCCL reduce-scatter, through the wire format; gloo, which
        has no reduce-scatter, all-reduces the whole buffer: same values in the shard, test backend only)"""
        import torch.distributed as dist
        lo, hi = self.shard
        g = self.flat_grad
        native = dist.get_backend(self.group) != "gloo"
        if self.grad_wire == "bf16":
            w = self._wire
            if g.is_cuda:
                lib = _lib.load()
                _lib.check(lib.immtsf_f32_to_bf16(_lib.ptr(g), _lib.ptr(w), g.numel(), _lib.stream_ptr()), "f32_to_bf16")
            else:
                w.copy_(g)
            if native:
                dist.reduce_scatter_tensor(self._wire_shard, w, group=self.group)
                src = self._wire_shard
            else:
                dist.all_reduce(w, group=self.group)
                src = w[lo:hi]
            if g.is_cuda:
                lib = _lib.load()
                _lib.check(lib.immtsf_bf16_to_f32(_lib.ptr(src), _lib.ptr(g[lo:hi]), hi - lo, _lib.stream_ptr()), "bf16_to_f32")
            else:
                g[lo:hi].copy_(src)
        elif native:
            dist.reduce_scatter_tensor(self._grad_shard, g, group=self.group)
            g[lo:hi].copy_(self._grad_shard)
        else:
            dist.all_reduce(g, group=self.group)

    def _all_gather(self, full, shard):
        import torch.distributed as dist
        if dist.get_backend(self.group) != "gloo":
            dist.all_gather_into_tensor(full, shard, group=self.group)
        else:
            per = shard.numel()
            dist.all_gather([full[r * per:(r + 1) * per] for r in range(self.world)], shard.clone(), group=self.group)

    def _step_sharded(self):
        """clip + Adam on this rank's shard (global norm: 1024 partial sums of squares, all-reduced), then all-gather"""
        import torch.distributed as dist
        lo, hi = self.shard
        per = hi - lo
        g = self.flat_grad[lo:hi]
        p = self.master if self.master is not None else self.flat_param[lo:hi]
        if self.flat_param.is_cuda:
            lib = _lib.load()
            sd = _lib.ptr(self.step_dev) if self.device_step else None
            dd = _lib.ptr(self.drop_dev) if self.device_step else None
            _lib.check(lib.immtsf_adam_sqnorm(_lib.ptr(g), per, _lib.ptr(self.norm_scratch), sd, dd, _lib.stream_ptr()), "adam_sqnorm")
            if self.collective:
                dist.all_reduce(self.norm_scratch, group=self.group)
            tw = _lib.ptr(self.flat_twin[lo:hi]) if self.flat_twin is not None else None
            _lib.check(lib.immtsf_adam_apply(_lib.ptr(p), _lib.ptr(g), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), per, self.lr,
                                             self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, sd, self.max_norm,
                                             _lib.ptr(self.norm_scratch), tw, _lib.stream_ptr()), "adam_apply")
        else:
            sq = (g * g).sum().reshape(1)
            if self.collective:
                dist.all_reduce(sq, group=self.group)
            gg = g
            if self.max_norm and self.max_norm > 0:
                gg = g * torch.clamp(self.max_norm / (sq.sqrt() + 1e-6), max=1.0)
            if self.wd:
                gg = gg + self.wd * p
            b1, b2 = self.betas
            self.exp_avg.mul_(b1).add_(gg, alpha=1 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(gg, gg, value=1 - b2)
            bc1, bc2 = 1 - b1 ** self.step_count, 1 - b2 ** self.step_count
            p.addcdiv_(self.exp_avg, self.exp_avg_sq.sqrt() / (bc2 ** 0.5) + self.eps, value=-self.lr / bc1)
            if self.flat_twin is not None:
                self.flat_twin[lo:hi].copy_(p)
        if not self.collective:          # rank-local leg: the other ranks' shards simply keep their values
            if self.param_wire != "bf16":
                self.refresh_twins()
            return
        if self.param_wire == "bf16":
            self._all_gather(self.flat_twin, self.flat_twin[lo:hi])
            if self.flat_param.is_cuda:
                lib = _lib.load()
                _lib.check(lib.immtsf_bf16_to_f32(_lib.ptr(self.flat_twin), _lib.ptr(self.flat_param), self.flat_param.numel(),
                                                  _lib.stream_ptr()), "bf16_to_f32")
            else:
                self.flat_param.copy_(self.flat_twin)
        else:
            self._all_gather(self.flat_param, self.flat_param[lo:hi])
            self.refresh_twins()

    def step(self):
        self.step_count += 1
        if self.sharded:
            self._step_sharded()
            return
        if self.flat_param.is_cuda and self.device_step and getattr(self, "step_guard", None):
            # behind a guard word (FlagStep's time-out report): a non-zero word drops the step on the device -- no update from
            # gradients a missed hand-over may have left incomplete (csrc/tail.hip sqnorm_partial_kernel / adam_kernel)
            lib = _lib.load()
            zero = getattr(self, "zero_in_step", False)
            self._grad_zeroed_by_step = bool(zero)
            _lib.check(lib.immtsf_adam_step_guarded(_lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.exp_avg),
                                                    _lib.ptr(self.exp_avg_sq), self.flat_param.numel(), self.lr, self.betas[0],
                                                    self.betas[1], self.eps, self.wd, _lib.ptr(self.step_dev), self.max_norm,
                                                    _lib.ptr(self.norm_scratch), _lib.ptr(self.drop_dev), int(self.step_guard),
                                                    1 if zero else 0, _lib.stream_ptr()), "adam_step_guarded")
        elif self.flat_param.is_cuda and self.device_step:
            lib = _lib.load()
            zero = getattr(self, "zero_in_step", False)
            fn = lib.immtsf_adam_step_dev_zero if zero else lib.immtsf_adam_step_dev
            self._grad_zeroed_by_step = bool(zero)
            _lib.check(fn(_lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.exp_avg),
                                                _lib.ptr(self.exp_avg_sq), self.flat_param.numel(), self.lr, self.betas[0],
                                                self.betas[1], self.eps, self.wd, _lib.ptr(self.step_dev), self.max_norm,
                                                _lib.ptr(self.norm_scratch), _lib.ptr(self.drop_dev), _lib.stream_ptr()),
                       "adam_step_dev")
        elif self.flat_param.is_cuda:
            lib = _lib.load()
            _lib.check(lib.immtsf_adam_step(_lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), _lib.ptr(self.exp_avg),
                                            _lib.ptr(self.exp_avg_sq), self.flat_param.numel(), self.lr, self.betas[0],
                                            self.betas[1], self.eps, self.wd, self.step_count, self.max_norm,
                                            _lib.ptr(self.norm_scratch), _lib.stream_ptr()), "adam_step")
        else:   # CPU (gloo tests of the DP logic only): same arithmetic in torch ops
            g = self.flat_grad
            if self.max_norm and self.max_norm > 0:
                g = g * torch.clamp(self.max_norm / (g.norm() + 1e-6), max=1.0)
            if self.wd:
                g = g + self.wd * self.flat_param
            b1, b2 = self.betas
            self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** self.step_count, 1 - b2 ** self.step_count
            self.flat_param.addcdiv_(self.exp_avg, self.exp_avg_sq.sqrt() / (bc2 ** 0.5) + self.eps, value=-self.lr / bc1)

    def adam_prepare(self, pending=None, err=None, skip_out=None, from_wire=False, guard=False):
        """first half of clip + Adam as launches the caller places (include/immtsf.h immtsf_adam_prepare): the squared norm of the whole
        gradient (from_wire: of its reduced bf16 wire image) and the step decision into `skip_out`; addresses are raw device pointers"""
        lib = _lib.load()
        n = self.flat_param.numel()
        wire = self._wire if (from_wire and self._wire is not None) else None
        gh = gf = None
        if guard:
            if wire is not None:
                gh = self._wire_store[n:].data_ptr()
            else:
                gf = self._grad_store[n:].data_ptr()
        _lib.check(lib.immtsf_adam_prepare(_lib.ptr(self.flat_grad), None if wire is None else wire.data_ptr(), n, _lib.ptr(self.norm_scratch),
                                           _lib.ptr(self.step_dev), _lib.ptr(self.drop_dev), pending, err, gh, gf, skip_out,
                                           _lib.stream_ptr()), "adam_prepare")

    def adam_range(self, lo, hi, skip=None, from_wire=False):
        """second half: the update of flat elements [lo, hi) on the current stream, clipped by adam_prepare's norm; leaves the gradient zero"""
        lib = _lib.load()
        wire = self._wire if (from_wire and self._wire is not None) else None
        _lib.check(lib.immtsf_adam_range(_lib.ptr(self.flat_param), _lib.ptr(self.flat_grad), None if wire is None else wire.data_ptr(),
                                         _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), self.flat_param.numel(), lo, hi, self.lr,
                                         self.betas[0], self.betas[1], self.eps, self.wd, _lib.ptr(self.step_dev), self.max_norm,
                                         _lib.ptr(self.norm_scratch), 1, skip, _lib.stream_ptr()), "adam_range")

    def snapshot(self):
        """parameters, Adam moments and step / dropout counters (GraphedStep / PhasedStep restore them after warming up)"""
        s = {"param": self.flat_param.clone(), "m": self.exp_avg.clone(), "v": self.exp_avg_sq.clone(), "step": self.step_count}
        if self.master is not None:
            s["master"] = self.master.clone()
        if self.device_step:
            s["step_dev"], s["drop_dev"] = self.step_dev.clone(), self.drop_dev.clone()
        return s

    def restore(self, s):
        self.flat_param.copy_(s["param"])
        self.exp_avg.copy_(s["m"])
        self.exp_avg_sq.copy_(s["v"])
        self.step_count = s["step"]
        if self.master is not None:
            self.master.copy_(s["master"])
        if self.device_step:
            self.step_dev.copy_(s["step_dev"])
            self.drop_dev.copy_(s["drop_dev"])
        if self.flat_twin is not None:
            if self.flat_param.is_cuda:
                lib = _lib.load()
                _lib.check(lib.immtsf_f32_to_bf16(_lib.ptr(self.flat_param), _lib.ptr(self.flat_twin), self.flat_param.numel(),
                                                  _lib.stream_ptr()), "f32_to_bf16")
            else:
                self.flat_twin.copy_(self.flat_param)

    def flush(self):
        """apply an optimizer step a step engine still holds back (FlagStep runs clip + Adam of step k at the head of replay k + 1): a
        no-op otherwise.  state_dict() calls it; call it yourself before reading parameters that must include the last step."""
        cb = getattr(self, "_flush_cb", None)
        if cb is not None:
            cb()

    def state_dict(self):
        """checkpoint of the optimizer: Adam moments, step counters and -- with a bf16 parameter wire -- the fp32 master
        parameters, all as FULL flat tensors (a sharded trainer gathers its ranks' shards; every rank returns the same
        dict).  Together with the modules' own state_dict this resumes training exactly; `load_state_dict` takes it back on
        any world size with the same parameter layout."""
        self.flush()

        def full(shard):
            if not self.sharded:
                return shard.detach().clone()
            out = torch.empty(self.flat_param.numel(), dtype=shard.dtype, device=shard.device)
            self._all_gather(out, shard.contiguous())
            return out
        sd = {"exp_avg": full(self.exp_avg), "exp_avg_sq": full(self.exp_avg_sq), "step": self.step_count,
              "n": self.flat_param.numel(), "world": self.world}
        if self.master is not None:
            sd["master"] = full(self.master)
        if self.device_step:
            sd["step_dev"], sd["drop_dev"] = self.step_dev.clone(), self.drop_dev.clone()
        return sd

    def load_state_dict(self, sd):
        n = self.flat_param.numel()
        k = min(n, int(sd["n"]))            # (the flat length is padded to a multiple of 8 * world: the common prefix is the payload)
        lo, hi = self.shard
        def take(dst, src):
            dst.zero_()
            a, b = lo, min(hi, k)
            if b > a:
                dst[:b - a].copy_(src[a:b].to(dst.device))
        take(self.exp_avg, sd["exp_avg"])
        take(self.exp_avg_sq, sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        if "master" in sd:
            if self.master is not None:
                take(self.master, sd["master"])
            else:                           # resuming without a bf16 parameter wire: the master IS the parameter vector
                self.flat_param[:k].copy_(sd["master"][:k].to(self.flat_param.device))
        elif self.master is not None:
            self.master.copy_(self.flat_param[lo:hi])
        if self.device_step and "step_dev" in sd:
            self.step_dev.copy_(sd["step_dev"])
            self.drop_dev.copy_(sd["drop_dev"])
        if self.master is None:
            self.refresh_twins()
        elif self.flat_twin is not None:    # replicated parameters = widened bf16 image of the masters: rebuild from the checkpoint
            src = sd["master"][:k].to(self.flat_param.device) if "master" in sd else self.flat_param[:k]
            self.flat_twin[:k].copy_(src)
            self.flat_param[:k].copy_(self.flat_twin[:k])

    def grad_bytes(self) -> int:
        return self.flat_grad.numel() * 4


class GraphedStep:
    """One training step as two hipGraphs (HIP streams + graphs instead of a tracing compiler):
        graph A = zero-grad, forward, loss, backward, gradient collection into the flat buffer
                  [+ with capture_collectives: the RCCL gradient all-reduces, bucket by bucket on the communication
                  stream as the backward finishes each bucket -- a parallel branch of the graph]
        [otherwise, world > 1: one eager RCCL all-reduce of the flat gradient between the graphs]
        graph B = clip + Adam, which also bumps the device-side Adam step and dropout-key counters, so every replay
                  is a NEW training step (needs FlatTrainer(device_step=True)).
    `loss_fn()` runs the forward and returns the scalar loss; inputs must be static device tensors."""

    def __init__(self, trainer: FlatTrainer, loss_fn, warmup: int = 3, capture_collectives: bool = False,
                 restore_after_warmup: bool = True, sched_gate: bool = False):
        if not trainer.device_step:
            raise ValueError("GraphedStep needs FlatTrainer(device_step=True): host-side step counters would freeze in the graph")
        self.trainer, self.loss_fn = trainer, loss_fn
        self.sched = torch.zeros(4, dtype=torch.int32, device=trainer.flat_param.device) if sched_gate else None
        self.captured_comm = bool(capture_collectives and trainer.collective)
        if trainer.collective and not self.captured_comm:
            # the bucket hooks must not fire inside the captured backward: their all-reduces would be captured (pulling the
            # communication stream into the capture without a join) AND repeated by sync_grads() between the graphs
            trainer.overlap = False
        # no communication between backward and optimizer (one process): clip + Adam ride at the end of graph A -- one graph
        # launch per step instead of two -- and Adam's pass over the gradient leaves it zero for the next replay (the 32 MB
        # zero-fill in front of the forward was 12 us that nothing could overlap: both branches wait for it)
        self.single = not trainer.collective
        if self.single and not trainer.sharded:
            trainer.zero_in_step = True
        side = torch.cuda.Stream(device=trainer.flat_param.device)
        side.wait_stream(torch.cuda.current_stream())
        # the warm-up runs REAL steps (allocator pools, lazy inits, RCCL channels): parameters, moments and the step / dropout
        # counters are put back afterwards, so that the first replay is training step 1 from the caller's weights
        snap = trainer.snapshot() if restore_after_warmup else None
        with torch.cuda.stream(side):           # off the default stream
            for _ in range(warmup):
                self._fwd_bwd()
                trainer.sync_grads()
                trainer.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if snap is not None:
            trainer.restore(snap)
            torch.cuda.synchronize()
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a):
            self.loss = self._fwd_bwd()
            if self.captured_comm:
                trainer.sync_grads()
            if self.single:
                trainer.step()
        if not trainer.sharded and not self.single:          # a sharded optimizer's step holds two collectives: it stays eager (three kernels)
            with torch.cuda.graph(self.graph_b):
                trainer.step()

    def _fwd_bwd(self):
        # (the 32 MB gradient memset as a third parallel branch of graph A, beside the forward, was measured: the step
        # got 5 % SLOWER -- 1.18 vs 1.12 ms -- so it stays in front of the forward)
        from . import config
        self.trainer.zero_grad()
        if self.sched is not None:
            # scheduling gate (config.sched_gate): the patch encoder's backward -- parameter gradients only, but a kernel that fills
            # every CU's LDS for a millisecond at thousands of windows -- starts when the text side's row-bound backward kernels are
            # through, beside its small-launch tail, instead of beside them (4096 windows: a 19 us reduction took 0.7 ms in its shade)
            fp = self.sched.data_ptr()
            _lib.check(_lib.load().immtsf_flags_clear(fp, 2, torch.cuda.current_stream().cuda_stream), "flags_clear")
            config.sched_gate, config.sched_armed = (fp, fp + 4), None
        try:
            loss = self.loss_fn()
            ops.backward_unit(loss)
        finally:
            config.sched_gate = config.sched_armed = None
        self.trainer.collect_grads()
        return loss

    def __call__(self):
        t = self.trainer
        self.graph_a.replay()
        if self.single:
            return self.loss
        if not self.captured_comm:
            t._reduced = [False] * len(t.buckets)
            t.sync_grads()
        if t.sharded:
            t.step()
        else:
            self.graph_b.replay()
        return self.loss


class PhasedStep:
    """One training step as SIX single-stream hipGraphs replayed on TWO HIP streams, with HIP events between them.

    The step has two chains that only meet at the modality fusion: the text side (TTF + the key/value half of MMF) and
    the backbone.  Captured as parallel branches of ONE hipGraph they are at the mercy of the graph executor: on
    ROCm 7.2 the branch that is not the capturing stream's continuation started 200-400 us after its inputs were ready
    (r02 traces: the forward overlapped, the two backward branches ran one after the other although the captured
    dependencies were exactly fork -> join).  Here every graph is a plain chain, the parallelism is two real streams, and
    the dependencies are event waits the host enqueues between graph launches:

        stream T (text):      T1 zero-grad, TTF fwd, MMF k|v fwd ........ T2 MMF query half fwd, loss, its backward ... T3 k|v + TTF bwd .. O clip+Adam
        stream B (backbone):  B1 backbone fwd ............................(waits T2) B2 backbone bwd, gradient collection ..........^
                                          T2 waits B1                                                      O waits B2

    text_fn() -> tuple of tensors (those that require grad are cut: the head sees detached copies and their gradients
    are fed back into T3); backbone_fn() -> pred_y; head_fn(pred_y, *text_out) -> scalar loss.  Needs
    FlatTrainer(device_step=True).  world > 1: one eager all-reduce of the flat gradient in front of O (like
    GraphedStep without captured collectives)."""

    def __init__(self, trainer: FlatTrainer, text_fn, backbone_fn, head_fn, warmup: int = 3):
        if not trainer.device_step:
            raise ValueError("PhasedStep needs FlatTrainer(device_step=True): host-side step counters would freeze in the graphs")
        self.trainer = trainer
        self.text_fn, self.backbone_fn, self.head_fn = text_fn, backbone_fn, head_fn
        if trainer.collective:
            trainer.overlap = False
        dev = trainer.flat_param.device
        self.T, self.B = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream()
        self.T.wait_stream(cur)
        self.B.wait_stream(cur)
        snap = trainer.snapshot()
        for _ in range(warmup):       # eager REAL steps on the two streams (allocator pools, lazy inits, RCCL channels) ...
            self._eager_step()
        torch.cuda.synchronize()
        trainer.restore(snap)         # ... undone: the first replay is training step 1 from the caller's weights
        torch.cuda.synchronize()
        G = torch.cuda.CUDAGraph
        self.gT1, self.gB1, self.gT2, self.gB2, self.gT3, self.gO = G(), G(), G(), G(), G(), G()
        # one memory pool per stream: graphs that replay concurrently must never be handed each other's freed blocks
        poolT, poolB = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        with torch.cuda.graph(self.gT1, pool=poolT, stream=self.T):
            trainer.zero_grad()
            outs = text_fn()
        with torch.cuda.graph(self.gB1, pool=poolB, stream=self.B):
            pred = backbone_fn()
        with torch.cuda.graph(self.gT2, pool=poolT, stream=self.T):
            py, cuts, loss = self._head(pred, outs)
            dpy = py.grad
            dcuts = [c.grad if (torch.is_tensor(c) and c.requires_grad) else None for c in cuts]
        with torch.cuda.graph(self.gB2, pool=poolB, stream=self.B):
            torch.autograd.backward([pred], [dpy])
            trainer.collect_grads()
        with torch.cuda.graph(self.gT3, pool=poolT, stream=self.T):
            self._text_backward(outs, dcuts)
        # the optimizer as a graph -- unless it holds collectives (sharded: norm all-reduce + parameter all-gather), which stay eager
        # like GraphedStep's (capture_collectives is the only path that puts RCCL calls into a graph)
        self.opt_eager = bool(trainer.sharded and trainer.collective)
        if not self.opt_eager:
            with torch.cuda.graph(self.gO, pool=poolT, stream=self.T):
                trainer.step()
        self.loss = loss
        self._keep = (outs, pred, py, cuts, dpy, dcuts)          # boundary tensors live in the graphs' pools: keep them referenced
        self.eB1, self.eT2, self.eB2, self.eO = (torch.cuda.Event() for _ in range(4))
        self.eO.record(self.T)

    def _head(self, pred, outs):
        from . import config
        py = pred.detach().requires_grad_(True)
        cuts = [o.detach().requires_grad_(True) if (torch.is_tensor(o) and o.requires_grad) else o for o in outs]
        loss = self.head_fn(py, *cuts)
        config.defer_param_grads = True       # the head's parameter gradients are enqueued behind the text-side backward (T3)
        try:
            ops.backward_unit(loss)
        finally:
            config.defer_param_grads = False
        return py, cuts, loss

    @staticmethod
    def _text_backward(outs, dcuts):
        ts = [o for o, g in zip(outs, dcuts) if g is not None]
        gs = [g for g in dcuts if g is not None]
        if ts:
            torch.autograd.backward(ts, gs)
        ops.run_deferred()

    def _eager_step(self):
        t = self.trainer
        with torch.cuda.stream(self.T):
            t.zero_grad()
            outs = self.text_fn()
        self.B.wait_stream(self.T)       # (the zero-fill precedes every gradient write)
        with torch.cuda.stream(self.B):
            pred = self.backbone_fn()
        self.T.wait_stream(self.B)
        with torch.cuda.stream(self.T):
            py, cuts, loss = self._head(pred, outs)
            dcuts = [c.grad if (torch.is_tensor(c) and c.requires_grad) else None for c in cuts]
        self.B.wait_stream(self.T)
        with torch.cuda.stream(self.B):
            torch.autograd.backward([pred], [py.grad])
            t.collect_grads()
        with torch.cuda.stream(self.T):
            self._text_backward(outs, dcuts)
        self.T.wait_stream(self.B)
        with torch.cuda.stream(self.T):
            t._reduced = [False] * len(t.buckets)
            t.sync_grads()
            t.step()
        return loss

    def __call__(self):
        t, T, B = self.trainer, self.T, self.B
        cur = torch.cuda.current_stream()
        T.wait_stream(cur)
        B.wait_stream(cur)          # the caller may have refreshed the static batch / the parameters on its stream
        with torch.cuda.stream(T):
            self.gT1.replay()
        with torch.cuda.stream(B):
            B.wait_event(self.eO)            # the previous step's parameter update
            self.gB1.replay()
            self.eB1.record(B)
        with torch.cuda.stream(T):
            T.wait_event(self.eB1)
            self.gT2.replay()
            self.eT2.record(T)
        with torch.cuda.stream(B):
            B.wait_event(self.eT2)
            self.gB2.replay()
            self.eB2.record(B)
        with torch.cuda.stream(T):
            self.gT3.replay()
            T.wait_event(self.eB2)
            if t.collective:
                t._reduced = [False] * len(t.buckets)
                t._collected = True
                t.sync_grads()
            if self.opt_eager:
                t.step()
            else:
                self.gO.replay()
            self.eO.record(T)
        cur.wait_stream(T)
        return self.loss


class FlagStep(PhasedStep):
    """PhasedStep's decomposition (text / backbone / head) as ONE hipGraph per step whose branches have NO edges between the fork at the
    start of the step and the join at its end: where a branch needs another's result it spins on a device flag (csrc/sync.hip) instead
    of waiting on an event.

    Why: on ROCm 7.2 a graph branch (or a stream) that reaches a dependency BEFORE it is satisfied resumes 110 - 175 us after the
    producer has finished (the text-side backward of the cfg2 step: profiles/r03_step_kernel_sequence.txt), whereas a wait that is
    already satisfied costs nothing.  A one-lane spin kernel resumes within a microsecond and occupies one wave slot.

        stream T:  norm + step decision | Adam(T's buckets), text fwd .. wait(B1) head fwd + bwd, set(T2) .. text bwd ... | join(P, B): one barrier | clear
        stream B:  (forked behind the norm)  Adam(B's buckets), backbone fwd, set(B1) .. wait(T2) backbone bwd, collect, set(B2: for the timeline tool)
        stream P:  (forked behind T's Adam)  Adam(P's buckets), fold, set .......... wait(tail inputs) parameter-gradient chain, set(P2)

    THE OPTIMIZER SITS AT THE HEAD OF THE NEXT REPLAY (round 5).  clip + Adam of step k run at the head of replay k + 1 (`flush()`
    applies the last one; FlatTrainer.state_dict() flushes): the squared norm of the whole gradient and the step decision on T
    (immtsf_adam_prepare), then the update of every bucket on the branch that reads its parameters FIRST (`adam_split`) -- the backbone
    starts behind its own 0.3 MB instead of behind all 277 MB of optimizer traffic, the text side behind its 150 MB, and the parameter
    branch updates MMF's parameters beside the text side's first kernels.  And because nothing follows the backward inside the graph,
    a data-parallel step is the SAME single graph launch:

        stream S (caller):  [wait_ge(comm_done, k-1)] graph k
        comm stream:        wait_ge(bucket i, k) all-reduce(bucket i) ... guard slot + last bucket ... bump(comm_done)     (eager)

    Inside the graph a bucket whose gradients are final -- a block's backward hook; TTF_T2V_XAttn's backward runs in three phases and
    fires a hook per phase (ops._t2v_phase_hooks); `backbone_buckets` complete with the backbone's backward; what nobody announces
    completes with the join -- is rounded to the bf16 wire image in place of a conversion kernel around the collective (bf16 wire) and
    bumps a COUNTING flag; the communication stream, not ordered behind S at all, spins for the k-th bump in front of the k-th
    replay's all-reduce of that bucket, in place, on the wire image; Adam at the head of replay k + 1 reads the reduced wire image
    directly.  Ordering argument (no spin and no collective ever waits on the other kind in a cycle): a spin on the communication
    stream waits only for kernels of ITS OWN rank's graph k, which wait for nothing outside that graph; a collective waits for the
    peers' same collective, which sits behind the peers' own spins; graph k + 1 waits (one eager spin on S) for the communication
    stream's last bump of step k.

    Fail-safe.  Every spin gives up after `timeout_ms` and sets the guard word flags[8].  The step decision reads it -- and, data
    parallel, the ranks' guard words SUMMED by the step's last collective (a slot behind the last bucket), so EVERY rank drops the same
    step (no update, step not counted, gradient zeroed) and the replicas stay identical; the word is sticky until clear_error().
    `check()` raises on every rank; `check_every` > 0 makes __call__ do that every so many steps."""

    # flag words (int32 offsets into self.flags)
    _B1, _T2, _B2, _FOLD, _TAIL, _P2, _SCHED, _SCHED_TO, _TTF = range(9)        # hand-over flags: cleared at the end of every replay
    _COUNT0 = 16                            # one counting flag per announced bucket
    _ERR, _PENDING, _SKIP, _COMM_DONE = 40, 41, 42, 43      # guard word, gradient pending, step decision, collectives done

    def __init__(self, trainer: FlatTrainer, text_fn, backbone_fn, head_fn, warmup: int = 3, param_tail: Optional[int] = None,
                 fold_by_flag: bool = True, head_flag: bool = True, param_branch: bool = True, sched_gate: bool = True,
                 adam_split=None, backbone_buckets: Sequence[int] = (), timeout_ms: int = 50, comm_timeout_ms: int = 5000,
                 check_every: int = 0, ttf_wgrad_tail: bool = True, merge_adjacent: bool = True, backbone_wgrad_tail: bool = True):
        """adam_split: (T's buckets, B's buckets, P's buckets) -- bucket indices whose clip + Adam update runs at the head of that
        branch; every bucket must be listed once, and a bucket belongs to the branch that reads its parameters FIRST in the step
        (bench.py: TTF -> T, the backbone -> B, MMF_XAttn_Add + the proj_out it folds -> P).  None: all on T, in front of the fork.
        backbone_buckets: buckets whose gradients are final when backbone_fn's backward (and the gradient collection) has run:
        announced on B instead of waiting for the join.
        merge_adjacent (data parallel): buckets announced in one burst (they complete at the same moment) that are neighbours in the flat
        buffer get one wire image, one flag and one collective.
        ttf_wgrad_tail: TTF_T2V_XAttn's early weight gradients (out_proj, attn.in_proj: inputs ready long before the text side's
        backward ends) leave the text side's dependent chain for the parameter branch, behind MMF_XAttn_Add's chain
        (ops.TTFT2VXAttnFn.backward; needs param_branch and gradient sinks).
        backbone_wgrad_tail: the weight gradients of a backbone's large linear layers (ops.LinearBf16Fn: PatchTST's q | k | v, output and
        feed-forward projections) leave the backbone's dependent chain the same way: the data gradient stays, the grouped weight-gradient
        launch runs on the parameter branch behind a flag of its own (up to seven layers per step; needs param_branch and gradient
        sinks).  The backbone's buckets are then announced on that branch."""
        if not trainer.device_step:
            raise ValueError("FlagStep needs FlatTrainer(device_step=True)")
        if trainer.sharded:
            raise ValueError("FlagStep reduces whole buckets: use GraphedStep / PhasedStep with a sharded optimizer")
        self.trainer = trainer
        self.text_fn, self.backbone_fn, self.head_fn = text_fn, backbone_fn, head_fn
        dev = trainer.flat_param.device
        lib = _lib.load()
        self.dist = bool(trainer.collective)
        self.timeout_ms, self.comm_timeout_ms, self.check_every = int(timeout_ms), int(comm_timeout_ms), int(check_every)
        self.T, self.B = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        # parameter-only work (MMF_XAttn_Add's fold in front, its parameter-gradient chain behind) on a THIRD branch of the graph
        self.P = torch.cuda.Stream(device=dev) if param_branch else self.B
        # launches of MMF_XAttn_Add's parameter chain left to the parameter branch (without one, data parallel: none -- the bucket's
        # hook must fire behind its LAST gradient write, on the branch that announces it)
        self._defer = (3 if param_branch else (0 if self.dist else 1)) if param_tail is None else int(param_tail)
        if self.dist:
            trainer.overlap = False           # no collectives from inside the captured backward (the hooks bump flags instead)
            if not param_branch:
                self._defer = 0
        nb = len(trainer.buckets)
        if adam_split is not None:
            listed = sorted(b for grp in adam_split for b in grp)
            if len(adam_split) != 3 or listed != list(range(nb)):
                raise ValueError("adam_split must be three lists that hold every bucket index exactly once")
            if not param_branch and adam_split[2]:
                adam_split = (list(adam_split[0]), list(adam_split[1]) + list(adam_split[2]), [])
        cur = torch.cuda.current_stream()
        self.T.wait_stream(cur)
        self.B.wait_stream(cur)
        self.P.wait_stream(cur)
        # (before the warm-up: its last Adam pass then leaves the gradient buffer zero, and the captured zero_grad() holds no fill)
        trainer.zero_in_step = True
        snap = trainer.snapshot()
        for _ in range(warmup):
            self._eager_step()
        torch.cuda.synchronize()
        trainer.restore(snap)
        trainer.flat_grad.zero_()
        torch.cuda.synchronize()
        self.flags = torch.zeros(48, dtype=torch.int32, device=dev)
        fp = self.flags.data_ptr()
        W = lambda i: fp + 4 * i        # noqa: E731
        F_B1, F_T2, F_B2, F_P2, F_ERR = W(self._B1), W(self._T2), W(self._B2), W(self._P2), W(self._ERR)
        self._f_err, self._f_pending, self._f_skip, self._f_comm = F_ERR, W(self._PENDING), W(self._SKIP), W(self._COMM_DONE)
        sp = lambda st: st.cuda_stream        # noqa: E731
        tmo = self.timeout_ms

        def fset(flag, st):
            _lib.check(lib.immtsf_flag_set(flag, sp(st)), "flag_set")

        def fwait(flag, st):
            _lib.check(lib.immtsf_flag_wait(flag, F_ERR, tmo, sp(st)), "flag_wait")

        bf16_wire = self.dist and trainer._wire is not None
        self._from_wire = bf16_wire
        # data parallel: counting flags (never cleared), one per announced bucket
        self.segments = []                    # [dict(flag, lo, hi, buckets, branch)]
        announced = set()
        branch_now = ["T"]

        def announce_range(lo, hi, buckets, burst=None):
            if hi == lo:
                return
            k = len(self.segments)
            if self._COUNT0 + k >= self._ERR:
                raise RuntimeError("FlagStep: more than 24 announced buckets")
            flag = W(self._COUNT0 + k)
            st = torch.cuda.current_stream().cuda_stream
            if bf16_wire:       # the wire image, written where the bucket completes: no conversion kernel around the collective.  (As ONE
                # launch with the announcement -- the last workgroup to finish bumps the flag -- it was slower: every workgroup's
                # release fence is an L2 write-back on this part, 0.63 vs 0.49 ms per step; the kernel boundary does it once.)
                _lib.check(lib.immtsf_f32_to_bf16(_lib.ptr(trainer.flat_grad[lo:hi]), _lib.ptr(trainer._wire[lo:hi]), hi - lo, st), "f32_to_bf16")
            _lib.check(lib.immtsf_flag_bump(flag, st), "flag_bump")
            self.segments.append({"flag": flag, "flags": [flag], "lo": lo, "hi": hi, "buckets": tuple(buckets), "branch": branch_now[0] * len(buckets)})

        bursts = {}

        def announce(bi, burst=None):
            if bi in announced:
                return
            announced.add(bi)
            if burst is None or not merge_adjacent:
                announce_range(*trainer.ranges[bi], (bi,))
                return
            # a burst: hooks fired back to back (the buckets complete at the same moment): its members that are neighbours in the flat
            # buffer get ONE wire image, ONE flag and ONE collective -- emitted when the burst's last member has reported
            token, i, cnt = burst
            bursts.setdefault(id(token), []).append(bi)
            if i + 1 < cnt:
                return
            members = sorted(bursts.pop(id(token)), key=lambda b_: trainer.ranges[b_][0])
            for lo, hi in _runs([trainer.ranges[b_] for b_ in members]):
                announce_range(lo, hi, [b_ for b_ in members if lo <= trainer.ranges[b_][0] and trainer.ranges[b_][1] <= hi and
                                        trainer.ranges[b_][1] > trainer.ranges[b_][0]])

        def adam(buckets):
            for lo, hi in _runs([trainer.ranges[b] for b in sorted(buckets)]):
                trainer.adam_range(lo, hi, skip=self._f_skip, from_wire=bf16_wire)

        self.graph = torch.cuda.CUDAGraph()
        B, P = self.B, self.P
        from . import config
        try:
            with torch.cuda.graph(self.graph):
                T = torch.cuda.current_stream()
                trainer._grad_zeroed_by_step = True       # (the Adam passes below leave every range zero)
                trainer.zero_grad()
                # ---- the previous replay's optimizer step: norm + decision, then the buckets on the branches that read them first
                trainer.adam_prepare(pending=self._f_pending, err=F_ERR, skip_out=self._f_skip, from_wire=bf16_wire, guard=self.dist)
                if adam_split is None:
                    adam(range(nb))
                    B.wait_stream(T)                  # fork (satisfied when B gets there: nothing runs on B before it)
                    if P is not B:
                        P.wait_stream(T)
                else:
                    B.wait_stream(T)
                    with torch.cuda.stream(B):
                        adam(adam_split[1])
                    adam(adam_split[0])
                    if P is not B:
                        P.wait_stream(T)              # (behind T's own update: the two large updates do not share the HBM)
                        with torch.cuda.stream(P):
                            adam(adam_split[2])
                config.fold_stream = P                # parameter-only work of the text side: on its own branch (or at the head of the backbone's)
                config.fold_flag = (W(self._FOLD), F_ERR) if fold_by_flag else None
                # scheduling hint (GraphedStep._fwd_bwd has the why): its time-out goes to its own word -- NOT the guard word, a hint
                # that expires costs nothing but the overlap it was after
                config.sched_gate, config.sched_armed = ((W(self._SCHED), W(self._SCHED_TO)) if sched_gate else None), None
                try:
                    outs = text_fn()
                finally:
                    config.fold_stream = None
                    config.fold_flag = None
                with torch.cuda.stream(B):
                    pred = backbone_fn()
                    fset(F_B1, B)
                fwait(F_B1, T)
                config.head_done_flag = F_T2 if head_flag else None
                try:
                    py, cuts, loss = self._head(pred, outs)
                    taken = config.head_done_flag is None and head_flag
                finally:
                    config.head_done_flag = None      # (never leave the address of this step's flag behind for an unrelated call)
                dpy = py.grad
                dcuts = [c.grad if (torch.is_tensor(c) and c.requires_grad) else None for c in cuts]
                if not taken:
                    fset(F_T2, T)                     # (a head that publishes the flag itself -- MMFXRankQLossFn -- has consumed it)
                elif dpy is None or dpy.data_ptr() != config.head_dy_ptr:
                    raise RuntimeError("FlagStep: the head published its dY flag early, but autograd did not hand that buffer on as the "
                                       "backbone's output gradient (head_flag=False disables the early flag)")
                # parameter-gradient tails (work only the optimizer waits for) go to the parameter branch: the text side's behind the TAIL
                # flag (their inputs exist), a backbone's large linear layers' weight gradients (immtsf.ops.LinearBf16Fn: PatchTST's
                # projections and feed-forward products) each behind a flag of its own from the spare words 9..15
                tail = {"flag": (W(self._TAIL), F_ERR), "jobs": [], "jobs_b": [], "defer": self._defer,
                        "ttf_flag": (W(self._TTF), F_ERR) if (ttf_wgrad_tail and P is not B) else None,
                        "wgrad_flags": [(W(i), F_ERR) for i in range(15, 8, -1)] if (backbone_wgrad_tail and P is not B) else []}
                with torch.cuda.stream(B):
                    fwait(F_T2, B)
                    config.param_tail = {"jobs_b": tail["jobs_b"], "wgrad_flags": tail["wgrad_flags"]} if tail["wgrad_flags"] else None
                    try:
                        torch.autograd.backward([pred], [dpy])
                    finally:
                        config.param_tail = None
                    trainer.collect_grads()
                    if self.dist and not tail["jobs_b"]:
                        branch_now[0] = "B"
                        for bi in backbone_buckets:
                            announce(bi)
                tail["wgrad_flags"] = []              # (the text side's own products stay where they are)
                config.param_tail = tail if self._defer > 0 else None
                trainer._capture_hook = announce if self.dist else None
                branch_now[0] = "T"
                try:
                    self._text_backward(outs, dcuts)
                finally:
                    config.param_tail = None
                    trainer._capture_hook = None
                trainer._capture_hook = announce if self.dist else None
                branch_now[0] = "P"
                try:
                    with torch.cuda.stream(P):
                        for job in tail["jobs_b"]:            # the backbone's deferred weight gradients, each behind its own flag
                            job(sp(P))
                        if self.dist and tail["jobs_b"]:      # (the backbone's buckets are complete here, not at the end of its branch)
                            for bi in backbone_buckets:
                                announce(bi)
                        if tail["jobs"]:
                            if tail.get("flag_set"):
                                fwait(W(self._TAIL), P)
                            for job in tail["jobs"]:
                                job(sp(P))
                finally:
                    trainer._capture_hook = None
                with torch.cuda.stream(P):
                    if P is not B:
                        fset(F_P2, P)
                with torch.cuda.stream(B):
                    fset(F_B2, B)
                # join: both branches as dependencies of ONE node (the flags_clear_set below).  A stream join costs 7 - 10 us on this stack
                # even when the joined branch ended long ago (profiles/r05_flag_timeline.txt); with a flag wait in front of each -- the
                # round-4 form: "the flag says so" -- the two joins were two barriers in a row behind the last branch: 18 us of a 440 us
                # step.  Back to back they become one barrier with two signals: same box, three alternating pairs of 40-step timelines,
                # median 425.6 us against 442.0.  (The two flags are still set: the timeline tool reads them.)
                if P is not B:
                    T.wait_stream(P)
                T.wait_stream(B)
                if self.dist:
                    # what nobody announced is complete now: contiguous runs of the remaining buckets, announced behind the join
                    branch_now[0] = "J"
                    rest = [bi for bi in range(nb) if bi not in announced]
                    for lo, hi in _runs([trainer.ranges[b] for b in rest]):
                        announce_range(lo, hi, [b for b in rest if lo <= trainer.ranges[b][0] and trainer.ranges[b][1] <= hi])
                    announced.update(rest)
                _lib.check(lib.immtsf_flags_clear_set(fp, 16, self._f_pending, sp(T)), "flags_clear_set")
        finally:
            config.sched_gate = config.sched_armed = None
        self._order_segments()
        self.comm = None
        if self.dist:
            # A stream of its own at DEFAULT priority.  HIP deals the streams of one priority round-robin onto a handful of hardware
            # queues, so this stream may share one with a branch of the graph and then run its spins behind that branch instead of
            # beside it (seen once: every bucket's wait entered 3 us after the step's last kernel, the collectives exposed in front of
            # the next step) -- slower, never wrong: the graph does not wait for this stream.  A HIGH-priority stream gets queues of
            # its own, but a spin kernel on a high-priority queue slowed the graph's text branch THREEFOLD (1.36 vs 0.51 ms per step:
            # profiles/r05_dist_ab.txt), so that is not the default.
            self.comm = torch.cuda.Stream(device=dev, priority=int(os.environ.get("IMMTSF_COMM_PRIO", "0")))
        self._epoch = 0
        self.loss = loss
        self._keep = (outs, pred, py, cuts, dpy, dcuts)
        trainer._flush_cb = self.flush

    # ------------------------------------------------------------------------------------------------------------------------------
    def timed_out(self) -> bool:
        """this rank's guard word (synchronises)"""
        return bool(int(self.flags[self._ERR].item()) != 0)

    def check(self):
        """raise, on EVERY rank, if a spin timed out on any of them since the last clear_error(): the steps since then were dropped --
        on every rank alike (a collective call when the trainer has a process group; synchronises)"""
        bad = self.flags[self._ERR:self._ERR + 1].to(torch.float32)
        if self.dist:
            import torch.distributed as dist
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.trainer.group)
        if float(bad.item()) != 0.0:
            raise _lib.ImmtsfError("FlagStep: a device-flag wait timed out (the graph's branches, or the communication stream, did not "
                                   "run concurrently): the guarded optimizer dropped the affected steps on every rank; clear_error() "
                                   "and go on, or restore a snapshot and use GraphedStep")

    def clear_error(self):
        self.flags[self._ERR:self._ERR + 1].zero_()

    def reset(self):
        """forget a gradient that is waiting for its optimizer step (after trial replays whose effect the caller undoes with
        trainer.restore): the next replay's head then applies nothing"""
        self._wait_comm(torch.cuda.current_stream())
        self.flags[self._PENDING:self._PENDING + 1].zero_()
        self.trainer.flat_grad.zero_()

    def _order_segments(self):
        """the communication order before any measurement: the text side's buckets but its last, the parameter branch's, the text side's
        last, the backbone's, the join's; the SAME on every rank (calibrate_comm_order replaces it by the measured completion order)"""
        tseg = [g for g in self.segments if g["branch"][0] == "T"]
        self.segments = (tseg[:-1] + [g for g in self.segments if g["branch"][0] == "P"] + tseg[-1:] +
                         [g for g in self.segments if g["branch"][0] == "B"] + [g for g in self.segments if g["branch"][0] == "J"])

    def calibrate_comm_order(self, replays: int = 3, merge_tail_us: float = 60.0):
        """order the communication stream's collectives by when their buckets ACTUALLY complete: `replays` replays under the flag
        kernels' own trace (immtsf_flag_trace: device wall clock, nothing serialised), the completion time of every segment (its last
        flag, relative to the previous step's clear) averaged over the replays and -- so that every rank ends up with the SAME order --
        over the ranks.  The static order of _order_segments is a guess from the program's structure; a collective that waits for a
        late bucket while earlier-finished ones queue behind it leaves them all exposed at the end of the step (measured: three
        collectives behind the last flag, 37 us in front of the next step; ordered by completion: one).  The replays are real training
        steps: callers undo them (bench.flag_step restores its snapshot).  A collective call when the trainer has a process group."""
        if not self.dist or len(self.segments) < 2:
            return
        lib = _lib.load()
        torch.cuda.synchronize()
        _lib.check(lib.immtsf_flag_trace(1), "flag_trace")
        for _ in range(replays):
            self()
        torch.cuda.synchronize()
        buf = (C.c_int64 * (3 * 1024))()
        n = lib.immtsf_flag_trace_read(buf, 1024)
        lib.immtsf_flag_trace(0)
        base = self.flags.data_ptr()
        ev = sorted((buf[3 * i + 2], buf[3 * i] - base, buf[3 * i + 1]) for i in range(max(n, 0)))
        t_clear, acc = None, {}
        for t, off, kind in ev:
            if kind == 3 and off == 0:
                t_clear = t
            elif kind == 0 and t_clear is not None:
                acc.setdefault(off, []).append((t - t_clear) / 100.0)
        times = []
        for g in self.segments:
            ts = acc.get(g["flag"] - base)
            times.append(sum(ts) / len(ts) if ts else float("inf"))
        tt = torch.tensor(times, dtype=torch.float64, device=self.flags.device)
        tt = torch.where(torch.isfinite(tt), tt, torch.full_like(tt, 1e9))
        import torch.distributed as dist
        dist.all_reduce(tt, group=self.trainer.group)
        order = sorted(range(len(self.segments)), key=lambda i: (float(tt[i]), i))
        segs = [dict(self.segments[i], done_us=float(tt[i]) / self.trainer.world) for i in order]
        # the buckets that complete within `merge_tail_us` of the LAST one, when they are one contiguous range of the flat buffer, go out
        # as ONE collective behind one wait on all their flags: every collective in the exposed tail costs its full latency (three
        # collectives behind the last flag: +37 us at one rank, ~3 x the RCCL latency at eight); an earlier bucket keeps its own
        tail = [g for g in segs if segs[-1]["done_us"] - g["done_us"] <= merge_tail_us]
        if len(tail) > 1:
            lo, hi = min(g["lo"] for g in tail), max(g["hi"] for g in tail)
            if sum(g["hi"] - g["lo"] for g in tail) == hi - lo:
                merged = {"flag": tail[-1]["flag"], "flags": [f for g in tail for f in g["flags"]], "lo": lo, "hi": hi,
                          "buckets": tuple(b for g in sorted(tail, key=lambda g_: g_["lo"]) for b in g["buckets"]),
                          "branch": "".join(g["branch"] for g in sorted(tail, key=lambda g_: g_["lo"])), "done_us": tail[-1]["done_us"]}
                segs = segs[:len(segs) - len(tail)] + [merged]
        self.segments = segs
        self.completion_us = [round(g["done_us"], 1) for g in segs]

    def _enqueue_collectives(self, cs, k):
        """the step's collectives on the current stream (raw handle `cs`): the seeds, then every segment behind its buckets' flags, this
        rank's guard word riding with the last one.  k: the replay number the counting flags are compared with."""
        import torch.distributed as dist
        t, lib = self.trainer, _lib.load()
        n = t.flat_param.numel()
        store = t._wire_store if self._from_wire else t._grad_store

        last = len(self.segments) - 1
        for i, g in enumerate(self.segments):
            hi = g["hi"]
            slot = None
            if i == last:
                # this rank's guard word rides with the last collective (written by the wait kernel itself): into the slot behind the
                # payload when the last range ends there, else as eight elements of its own
                slot = store[n:].data_ptr()
                if hi == n:
                    hi = n + 8
            fl = g["flags"]
            for j in range(0, len(fl), 4):       # ONE wait launch per (up to four) flags of a merged segment
                part = fl[j:j + 4]
                arr = (C.c_void_p * len(part))(*part)
                _lib.check(lib.immtsf_flag_wait_ge_multi(len(part), arr, k, self._f_err, self.timeout_ms, slot if j + 4 >= len(fl) else None,
                                                         1 if self._from_wire else 0, cs), "flag_wait_ge_multi")
            dist.all_reduce(store[g["lo"]:hi], group=t.group)
            if i == last and hi != n + 8:
                dist.all_reduce(store[n:n + 8], group=t.group)

    def _wait_comm(self, stream):
        """`stream` waits (one spin kernel) for the collectives of the last replay"""
        if self.dist and self._epoch > 0:
            _lib.check(_lib.load().immtsf_flag_wait_ge(self._f_comm, self._epoch & 0x7FFFFFFF, self._f_err, self.comm_timeout_ms,
                                                       stream.cuda_stream), "flag_wait_ge")

    def flush(self):
        """apply the optimizer step of the last replay now (eager launches on the current stream); the next replay's head then finds
        nothing pending.  Idempotent."""
        S = torch.cuda.current_stream()
        self._wait_comm(S)
        t = self.trainer
        t.adam_prepare(pending=self._f_pending, err=self._f_err, skip_out=self._f_skip, from_wire=self._from_wire, guard=self.dist)
        t.adam_range(0, t.flat_param.numel(), skip=self._f_skip, from_wire=self._from_wire)
        # (the dropout counter advanced with the decision: harmless -- every replay draws a fresh key anyway)

    def __call__(self):
        S = torch.cuda.current_stream()
        self._wait_comm(S)
        self.graph.replay()
        self._epoch += 1
        if self.dist:
            with torch.cuda.stream(self.comm):
                self._enqueue_collectives(self.comm.cuda_stream, self._epoch & 0x7FFFFFFF)
                _lib.check(_lib.load().immtsf_flag_bump(self._f_comm, self.comm.cuda_stream), "flag_bump")
        if self.check_every and self._epoch % self.check_every == 0:
            self.check()
        return self.loss


def _runs(ranges):
    """contiguous runs of (lo, hi) ranges (sorted by lo; empty ones dropped)"""
    out = []
    for lo, hi in sorted(r for r in ranges if r[1] > r[0]):
        if out and out[-1][1] == lo:
            out[-1] = (out[-1][0], hi)
        else:
            out.append((lo, hi))
    return out
