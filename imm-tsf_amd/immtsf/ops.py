"""torch.autograd.Function wrappers around the block-level C ABI (include/immtsf.h).

Every Function enqueues on torch's current HIP stream; the forward workspace (which holds the saved-for-backward
state) is a torch uint8 tensor kept on `ctx`; gradients are written by the HIP backward into one flat buffer
and returned as views.  Inputs must be fp32, contiguous and on the GPU: there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, config
from ._lib import DecoderParams, FusionCfg, GCNParams, GRParams, NoteIndex, RecAvgParams, T2VParams, TTCNParams, XAddParams, check, ptr, stream_ptr


def _need_gpu(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.ImmtsfError("immtsf ops need tensors on the GPU (HIP); there is no CPU fallback")
        if t.dtype not in (torch.float32, torch.uint8, torch.bool, torch.int32):
            raise _lib.ImmtsfError(f"unsupported dtype {t.dtype}; the HIP path takes fp32 tensors")


def _c(t):
    return None if t is None else t.contiguous()


def _struct(cls, tensors):
    s = cls()
    for name, t in zip(cls.FIELDS, tensors):
        setattr(s, name, None if t is None else t.data_ptr())
    return s


def _sinks_of(params):
    """Gradient sinks registered by immtsf.train.FlatTrainer: a parameter carrying `_immtsf_grad_sink` (a view into
    the trainer's flat gradient buffer) gets its gradient written there directly by the HIP backward, and autograd
    is handed None for it (no AccumulateGrad add kernel, no per-parameter allocation)."""
    return [None if p is None else getattr(p, "_immtsf_grad_sink", None) for p in params]


def _shared(params, sinks):
    """True when every given parameter writes to a sink that several HIP backward ops ADD into (FlatTrainer sink_shared:
    the buffers are zero-filled once per step and every writer accumulates)."""
    return all(s is not None and getattr(p, "_immtsf_grad_shared", False) for p, s in zip(params, sinks))


def _claim_sinks(params, sinks, who):
    """two HIP ops that write the SAME parameters' sinks in one step (tPatchGNN's LearnableTE: the patch encoder and the decoder) must
    both accumulate -- which they do only when the trainer declared the parameters `sink_shared`.  Each op names itself here in its
    forward; a second writer of an undeclared sink raises instead of silently overwriting the first one's gradient."""
    for p, s in zip(params, sinks):
        if p is None or s is None or getattr(p, "_immtsf_grad_shared", False):
            continue
        seen = getattr(p, "_immtsf_sink_writers", None)
        if seen is None:
            p._immtsf_sink_writers = {who}
        elif who not in seen:
            raise _lib.ImmtsfError("a parameter's gradient sink has two writers (%s and %s) but was not declared shared: pass it in "
                                   "FlatTrainer(sink_shared=[...])" % (sorted(seen)[0], who))


def _prezeroed(params, sinks):
    """1 when every gradient of the block goes to a sink that its owner zero-fills each step (FlatTrainer)."""
    ok = all(p is None or (s is not None and getattr(p, "_immtsf_grad_prezeroed", False)) for p, s in zip(params, sinks))
    return 1 if ok else 0


def _grad_buffers(params, sinks=None):
    """(buffers the HIP backward writes, gradients to return to autograd).  One flat allocation for the
    parameters without a sink."""
    sinks = sinks or [None] * len(params)
    n = sum(p.numel() for p, s in zip(params, sinks) if p is not None and s is None)
    dev = next(p.device for p in params if p is not None)
    flat = torch.empty(n, dtype=torch.float32, device=dev) if n else None
    bufs, rets, off = [], [], 0
    for p, s in zip(params, sinks):
        if p is None:
            bufs.append(None)
            rets.append(None)
        elif s is not None:
            bufs.append(s)
            rets.append(None)
        else:
            v = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
            bufs.append(v)
            rets.append(v)
    return bufs, rets


def _zeroed_grad_buffers(params, sinks):
    """like _grad_buffers for backward kernels that ACCUMULATE parameter gradients by atomics: the buffers must read zero.
    Sinks whose owner zero-fills them every step (FlatTrainer) are used as they are; everything else comes out of one
    zero-filled allocation."""
    pre = [s is not None and getattr(p, "_immtsf_grad_prezeroed", False) for p, s in zip(params, sinks)]
    n = sum(p.numel() for p, ok in zip(params, pre) if not ok)
    flat = torch.zeros(n, dtype=torch.float32, device=params[0].device) if n else None
    bufs, rets, off = [], [], 0
    for p, s, ok in zip(params, sinks, pre):
        if ok:
            bufs.append(s)
            rets.append(None)
        else:
            v = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
            bufs.append(v)
            rets.append(v)
    return bufs, rets


def _fire(hook, burst=None):
    """run a bucket hook (immtsf.train.FlatTrainer); burst: (token, index, count) of hooks fired back to back -- buckets that complete at
    the same moment -- which lets a data-parallel step pack, announce and send neighbours in the flat buffer as one"""
    if hook is not None:
        if burst is not None and getattr(hook, "_immtsf_bucket_index", None) is not None:
            hook(burst)
        else:
            hook()


def _bytes(n, dev):
    return torch.empty(max(int(n), 1), dtype=torch.uint8, device=dev)


# ---- bf16 images of activations that cross a block boundary (bf16 mode).  The producer registers (fp32 tensor, bf16
# image); a consumer that is handed the SAME, unmodified fp32 data finds the image and skips its cast kernel.  The registry
# holds the fp32 tensor's STORAGE (so the address cannot be recycled under a stale entry) but not the tensor: a gradient that a
# backward returns must stay un-referenced, or autograd's AccumulateGrad clones it instead of taking it (a copy launch on the
# serial section in front of the backbone's backward, and the clone then misses its image: a cast launch more).
_shadows = []          # [(storage, data_ptr, shape, version, bf16 tensor)], newest last, at most _SHADOW_CAP entries
_SHADOW_CAP = 6


def _shadow_put(t, h):
    _shadows.append((t.untyped_storage(), t.data_ptr(), tuple(t.shape), t._version, h))
    if len(_shadows) > _SHADOW_CAP:
        del _shadows[0]


def _shadow_get(x):
    if not x.is_contiguous():
        return None
    for _st, p, shape, ver, h in reversed(_shadows):
        if p == x.data_ptr() and shape == tuple(x.shape) and x._version == ver:
            return h
    return None


# ---- hand-over of the fused tail (config.z_handover).  Forward: the producer (TTFT2VXAttnFn) registers its output Z with the cfg of its
# call; the consumer (MMFXRankPFn) that is handed the same data finds it and may ask the library whether the producer's backward takes a
# low-rank gradient.  Backward: the consumer registers (dZ placeholder -> dP, Wc); the producer's backward looks its incoming gradient up.
_handover = []         # [(storage, data_ptr, shape, version, FusionCfg of the producer's call, half_only)]
_lowrank = []          # [(storage, data_ptr, shape, version, (LowRankGrad struct, tensors it points into))]


def _reg_put(reg, t, payload):
    reg.append((t.untyped_storage(), t.data_ptr(), tuple(t.shape), t._version, payload))
    if len(reg) > _SHADOW_CAP:
        del reg[0]


def _reg_take(reg, x, pop=False):
    if x is None or not x.is_contiguous():
        return None
    for i in range(len(reg) - 1, -1, -1):
        _st, p, shape, ver, payload = reg[i]
        if p == x.data_ptr() and shape == tuple(x.shape) and x._version == ver:
            if pop:
                del reg[i]
            return payload
    return None


def _bf16_dataflow(precision, d):
    return precision == 1 and d >= 16 and d % 16 == 0


def make_cfg(B, N, T, Cc, d_m, d, H, precision, training, p_drop, kappa, seed, device=None):
    return FusionCfg(B, N, T, Cc, d_m, d, H, precision, 1 if training else 0, float(p_drop), float(kappa),
                     int(seed) & 0xFFFFFFFFFFFFFFFF, None if device is None else config.dropout_counter_ptr(device))


# ------------------------------------------------------------------------------------------------ TTF_T2V_XAttn
class PackedNotes:
    """A batch's notes in packed-ragged form over a resident embedding matrix (immtsf.data.ResidentStore.collate):
    emb [rows, d_m] fp32, src_rows [sum_n] int32 (row of each note, window-major), lengths [B] int32, N = padded width
    of the accompanying tau (B,N).  Accepted by TTF_T2V_XAttn.forward in place of the zero-padded (B,N,d_m) tensor."""

    def __init__(self, emb, src_rows, lengths, N):
        if emb.numel() >= 2 ** 32:
            raise ValueError("resident embedding matrix must stay below 2^32 elements (32-bit gather offsets)")
        self.emb, self.src_rows, self.lengths, self.N = emb, src_rows, lengths, int(N)
        self.shape = (lengths.shape[0], self.N, emb.shape[1])
        self.device = emb.device
        self._index = None

    def index(self):
        """the batch's ragged index (immtsf_note_index: mask, M_txt, lengths, offsets, rowmap, seg), built ONCE per batch -- the batch
        builder knows every window's note count, so the index is part of the hand-over (offsets authoritative) instead
        of two launches at the head of every forward.  Returns (ctypes struct, tensors it points into)."""
        if self._index is None:
            B, N = self.lengths.shape[0], self.N
            dev = self.device
            u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)        # noqa: E731
            i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)       # noqa: E731
            t = {"mask": u8(B * N), "mtxt": u8(B), "lengths": i32(B), "offsets": i32(B + 1), "rowmap": i32(B * N), "seg": i32(B * N)}
            ix = NoteIndex(*[t[f].data_ptr() for f in NoteIndex.FIELDS])
            check(_lib.load().immtsf_note_index_build(ptr(self.lengths), B, N, C.byref(ix), stream_ptr()), "note_index_build")
            self._index = (ix, t)
        return self._index


class TTFT2VXAttnFn(torch.autograd.Function):
    """(notes (B,N,d_m) | resident emb + src_rows + lengths, tau (B,N)) -> (E_txt (B,T,d), M_txt uint8 (B)).
    params in immtsf_t2v_params order."""

    @staticmethod
    def forward(ctx, notes, tau, T, H, p_drop, training, precision, seed, nan_flag, src_rows, lengths, no_proj, *params):
        # no_proj: False | True | "handover" (True + config.z_handover's promises: the output's only consumer reads the bf16 image when there
        # is one, and may send its gradient back in low-rank form)
        lib = _lib.load()
        handover = no_proj == "handover" and config.z_handover
        no_proj = bool(no_proj)
        notes, tau = _c(notes), _c(tau)
        params = tuple(_c(p) for p in params)
        _need_gpu(notes, tau, *params)
        # (src_rows may come as the PackedNotes itself: then the batch's prebuilt ragged index is used instead of deriving it in the call)
        index = None
        if isinstance(src_rows, PackedNotes):
            index = src_rows.index() if config.note_index else None
            src_rows = src_rows.src_rows
        packed = src_rows is not None
        B, N = tau.shape
        d_m = notes.shape[-1]
        d = params[0].numel()
        cfg = make_cfg(B, N, T, 0, d_m, d, H, precision, training, p_drop, 0.0, seed, notes.device)
        cfg.form = {"auto": 0, "chain": 1, "fold": 2, "mix": 3}[config.t2v_form]        # (the backward reads the same cfg: one form per call pair)
        if no_proj:             # proj_out is left to the consumer (MMFXRankPFn's "_z" form): the output is Z, the LayerNorm + dropout result
            cfg.form |= _lib.FORM_NO_PROJ
        ctx.no_proj = bool(no_proj)
        ws = _bytes(lib.immtsf_ttf_t2v_xattn_workspace_bytes(C.byref(cfg)), notes.device)
        E = torch.empty(B, T, d, dtype=torch.float32, device=notes.device)
        M = torch.empty(B, dtype=torch.uint8, device=notes.device)
        ps = _struct(T2VParams, params)
        E_h = None
        if _bf16_dataflow(precision, d) and d_m % 8 == 0:
            E_h = torch.empty(B, T, d, dtype=torch.bfloat16, device=notes.device)
            cfg.out_h = E_h.data_ptr()
        if handover and E_h is not None and config.nan_check != "sync":
            cfg.form |= _lib.FORM_HALF_OUT        # (E stays unwritten: whoever reads it must take the image -- MMFXRankPFn checks)
        if index is not None:
            cfg.note_index = C.addressof(index[0])
            M = index[1]["mtxt"].view(-1)        # (a fresh tensor object over the batch's M_txt: no copy, no launch)
        ctx.index = index                        # (keeps the struct and its tensors alive for the backward)
        if packed:
            check(lib.immtsf_ttf_t2v_xattn_forward_packed(C.byref(cfg), C.byref(ps), ptr(notes), ptr(src_rows), ptr(lengths),
                                                          ptr(tau), ptr(E), ptr(M), ptr(ws), ws.numel(), stream_ptr()),
                  "ttf_t2v_xattn_forward_packed")
        else:
            check(lib.immtsf_ttf_t2v_xattn_forward(C.byref(cfg), C.byref(ps), ptr(notes), ptr(tau), ptr(E), ptr(M), ptr(ws),
                                                   ws.numel(), ptr(nan_flag), stream_ptr()), "ttf_t2v_xattn_forward")
        cfg.out_h = None
        if E_h is not None:
            _shadow_put(E, E_h)
        if handover:
            _reg_put(_handover, E, (cfg, bool(cfg.form & _lib.FORM_HALF_OUT)))
        ctx.cfg, ctx.ws, ctx.src_rows = cfg, ws, src_rows
        # scheduling hint of a two-stream captured step (immtsf.train.GraphedStep): this call's backward will set the gate behind its
        # row-bound kernels; remember who promised, so that nobody waits for a flag nobody sets
        ctx.gate = None
        if config.sched_gate is not None and training:
            ctx.gate = config.sched_gate[0]
            config.sched_armed = torch.cuda.current_stream().cuda_stream
        ctx.save_for_backward(notes, tau, *[p if p is not None else notes.new_empty(0) for p in params])
        ctx.none_mask = [p is None for p in params]
        ctx.sinks = _sinks_of(params)
        ctx.cfg.grads_prezeroed = _prezeroed(params, ctx.sinks)
        # bucket hooks (immtsf.train.FlatTrainer): a hook fires behind the backward PHASE that completes its bucket -- out_proj /
        # LayerNorm (/ proj_out) gradients are final long before input_proj's, and a data-parallel step hands them to the all-reduce then
        ctx.folded = bool(lib.immtsf_ttf_t2v_xattn_folded(C.byref(cfg)))
        ctx.phase_hooks = _t2v_phase_hooks(params, no_proj, ctx.folded)
        ctx.mark_non_differentiable(M)
        ctx.set_materialize_grads(False)      # no zero-filled uint8 'gradient' of M (one fill kernel per backward)
        return E, M

    @staticmethod
    def backward(ctx, dE, _dM):
        lib = _lib.load()
        saved = ctx.saved_tensors
        notes, tau = saved[0], saved[1]
        params = [None if isnone else p for p, isnone in zip(saved[2:], ctx.none_mask)]
        if dE is None:       # E_txt unused downstream (grads are not materialised, see forward)
            dE = torch.zeros(ctx.cfg.B, ctx.cfg.T, ctx.cfg.d, dtype=torch.float32, device=notes.device)
        dE = dE.contiguous()
        grads, rets = _grad_buffers(params, ctx.sinks)
        sc = _bytes(lib.immtsf_ttf_t2v_xattn_scratch_bytes(C.byref(ctx.cfg)), notes.device)
        ps, gs = _struct(T2VParams, params), _struct(T2VParams, grads)
        ctx.cfg.sched_flag = ctx.gate
        dE_h = _shadow_get(dE) if _bf16_dataflow(ctx.cfg.precision, ctx.cfg.d) else None
        ctx.cfg.in_h = None if dE_h is None else dE_h.data_ptr()
        lowrank = _reg_take(_lowrank, dE, pop=True) if ctx.no_proj else None      # (dE is then an unwritten placeholder: dZ = coef basis)
        ctx.cfg.lr_grad = None if lowrank is None else C.addressof(lowrank[0])
        cfg, hooks = ctx.cfg, ctx.phase_hooks
        packed = ctx.src_rows is not None

        def call(mask, stream=None):
            cfg.bwd_phase = mask
            st = stream_ptr() if stream is None else stream
            if packed:
                check(lib.immtsf_ttf_t2v_xattn_backward_packed(C.byref(cfg), C.byref(ps), ptr(notes), ptr(ctx.src_rows), ptr(tau),
                                                               ptr(dE), ptr(ctx.ws), ctx.ws.numel(), ptr(sc), sc.numel(),
                                                               C.byref(gs), st), "ttf_t2v_xattn_backward_packed")
            else:
                check(lib.immtsf_ttf_t2v_xattn_backward(C.byref(cfg), C.byref(ps), ptr(notes), ptr(tau), ptr(dE), ptr(ctx.ws),
                                                        ctx.ws.numel(), ptr(sc), sc.numel(), C.byref(gs), st),
                      "ttf_t2v_xattn_backward")
            cfg.bwd_phase = 0

        tail = config.param_tail
        split = (tail is not None and tail.get("ttf_flag") is not None and tail.get("defer", 0) > 0 and not ctx.folded and
                 all(r is None for r in rets))
        if split:
            # a step with a parameter-only branch (immtsf.train.FlagStep): this stream runs the data path and the LAST phase's weight
            # gradients; the weight gradients of phases A and B -- whose inputs exist long before the end -- leave as one grouped launch
            # on that branch, behind a flag that says phase B's data path has run.  Their buckets' hooks fire there.
            call(_lib.BWD_PHASE_A | _lib.BWD_PHASE_B)
            flag, err = tail["ttf_flag"]
            check(lib.immtsf_flag_set(flag, stream_ptr()), "flag_set")
            call(_lib.BWD_PHASE_C | _lib.BWD_WGRAD_C)
            early = [h for ph, h in hooks if ph < 2]
            keep = (notes, tau, dE, sc, ps, gs, params, grads, dE_h, lowrank)   # the job runs later, on another stream: everything it touches stays alive

            def job(stream, keep=keep):
                check(lib.immtsf_flag_wait(flag, err, 50, stream), "flag_wait")
                call(_lib.BWD_WGRAD_A | _lib.BWD_WGRAD_B, stream)
                token = object()
                for i_, h in enumerate(early):
                    _fire(h, (token, i_, len(early)))
            tail["jobs"].append(job)
            for ph, h in hooks:
                if ph == 2:
                    _fire(h)
        else:
            # one call (every weight gradient in one grouped launch) unless a bucket completes before the last phase: then one call per
            # run of phases up to the next hook (immtsf_fusion_cfg.bwd_phase), data path and weight gradients together
            cuts = sorted({ph for ph, _ in hooks if ph < 2})
            lo = 0
            for c in cuts + [2]:
                call(sum(0x11 << i for i in range(lo, c + 1)) if cuts else 0)
                lo = c + 1
                for ph, h in hooks:
                    if ph == c:
                        _fire(h)
        cfg.lr_grad = None
        if ctx.no_proj:         # (proj_out's gradients come out of the consumer's backward)
            rets = list(rets)
            rets[15] = rets[16] = None
        return (None,) * 12 + tuple(rets)


_T2V_PHASE = (1, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 0, 0, 0, 0, 0, 0)      # immtsf_t2v_params index -> backward phase that completes its gradient


def _t2v_phase_hooks(params, no_proj, folded):
    """[(phase, hook)] for TTFT2VXAttnFn.backward: every bucket hook found on the block's parameters, with the phase (0 A, 1 B, 2 C:
    include/immtsf.h IMMTSF_BWD_PHASE_*) behind which ALL of the block's parameters in that hook's bucket have their final gradient.  The
    folded form's gradients all leave its chain rule at the end (phase C); so does a hook whose parameter carries no bucket stamp."""
    own = [(i, p) for i, p in enumerate(params) if p is not None and not (no_proj and i >= 15)]
    out, seen = [], set()
    for i, p in own:
        hook = getattr(p, "_immtsf_bwd_hook", None)
        if hook is None or id(hook) in seen:
            continue
        seen.add(id(hook))
        bucket = getattr(p, "_immtsf_bucket", None)
        if folded or bucket is None:
            out.append((2, hook))
            continue
        out.append((max(_T2V_PHASE[j] for j, q in own if getattr(q, "_immtsf_bucket", None) == bucket), hook))
    return out


# ------------------------------------------------------------------------------------------------ TTF_RecAvg
class TTFRecAvgFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, notes, tau, t_hat, p_drop, training, precision, seed, nan_flag, *params):
        lib = _lib.load()
        notes, tau, t_hat = _c(notes), _c(tau), _c(t_hat)
        params = tuple(_c(p) for p in params)
        _need_gpu(notes, tau, t_hat, *params)
        B, N, d_m = notes.shape
        T = t_hat.shape[1]
        d = params[3].numel()      # ln_w
        cfg = make_cfg(B, N, T, 0, d_m, d, 1, precision, training, p_drop, 0.0, seed, notes.device)
        ws = _bytes(lib.immtsf_ttf_recavg_workspace_bytes(C.byref(cfg)), notes.device)
        E = torch.empty(B, T, d, dtype=torch.float32, device=notes.device)
        M = torch.empty(B, dtype=torch.uint8, device=notes.device)
        ps = _struct(RecAvgParams, params)
        check(lib.immtsf_ttf_recavg_forward(C.byref(cfg), C.byref(ps), ptr(notes), ptr(tau), ptr(t_hat), ptr(E), ptr(M),
                                            ptr(ws), ws.numel(), ptr(nan_flag), stream_ptr()), "ttf_recavg_forward")
        ctx.cfg, ctx.ws = cfg, ws
        ctx.save_for_backward(notes, tau, t_hat, *[p if p is not None else notes.new_empty(0) for p in params])
        ctx.none_mask = [p is None for p in params]
        ctx.sinks = _sinks_of(params)
        ctx.cfg.grads_prezeroed = _prezeroed(params, ctx.sinks)
        ctx.done_hook = getattr(params[0], "_immtsf_bwd_hook", None)
        ctx.mark_non_differentiable(M)
        ctx.set_materialize_grads(False)      # no zero-filled uint8 'gradient' of M (one fill kernel per backward)
        return E, M

    @staticmethod
    def backward(ctx, dE, _dM):
        lib = _lib.load()
        saved = ctx.saved_tensors
        notes, tau, t_hat = saved[:3]
        params = [None if isnone else p for p, isnone in zip(saved[3:], ctx.none_mask)]
        if dE is None:
            dE = torch.zeros(ctx.cfg.B, ctx.cfg.T, ctx.cfg.d, dtype=torch.float32, device=notes.device)
        dE = dE.contiguous()
        grads, rets = _grad_buffers(params, ctx.sinks)
        sc = _bytes(lib.immtsf_ttf_recavg_scratch_bytes(C.byref(ctx.cfg)), notes.device)
        ps, gs = _struct(RecAvgParams, params), _struct(RecAvgParams, grads)
        check(lib.immtsf_ttf_recavg_backward(C.byref(ctx.cfg), C.byref(ps), ptr(notes), ptr(tau), ptr(t_hat), ptr(dE),
                                             ptr(ctx.ws), ctx.ws.numel(), ptr(sc), sc.numel(), C.byref(gs), stream_ptr()),
              "ttf_recavg_backward")
        _fire(ctx.done_hook)
        return (None,) * 8 + tuple(rets)


# ------------------------------------------------------------------------------------------------ MMF_XAttn_Add
class MMFXAttnAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Y, E, M_u8, H, kappa, p_drop, training, precision, seed, *params):
        lib = _lib.load()
        Y, E, M_u8 = _c(Y), _c(E), _c(M_u8)
        params = tuple(_c(p) for p in params)
        _need_gpu(Y, E, M_u8, *params)
        B, T, Cc = Y.shape
        d = E.shape[2]
        cfg = make_cfg(B, 0, T, Cc, 0, d, H, precision, training, p_drop, kappa, seed, Y.device)
        ws = _bytes(lib.immtsf_mmf_xattn_add_workspace_bytes(C.byref(cfg)), Y.device)
        out = torch.empty_like(Y)
        ps = _struct(XAddParams, params)
        check(lib.immtsf_mmf_xattn_add_forward(C.byref(cfg), C.byref(ps), ptr(Y), ptr(E), ptr(M_u8), ptr(out), ptr(ws),
                                               ws.numel(), stream_ptr()), "mmf_xattn_add_forward")
        ctx.cfg, ctx.ws = cfg, ws
        ctx.sinks = _sinks_of(params)
        ctx.cfg.grads_prezeroed = _prezeroed(params, ctx.sinks)
        ctx.done_hook = getattr(params[0], "_immtsf_bwd_hook", None)
        ctx.save_for_backward(Y, E, M_u8, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        Y, E, M_u8, *params = ctx.saved_tensors
        dout = dout.contiguous()
        grads, rets = _grad_buffers(params, ctx.sinks)
        dY, dE = torch.empty_like(Y), torch.empty_like(E)
        sc = _bytes(lib.immtsf_mmf_xattn_add_scratch_bytes(C.byref(ctx.cfg)), Y.device)
        ps, gs = _struct(XAddParams, params), _struct(XAddParams, grads)
        check(lib.immtsf_mmf_xattn_add_backward(C.byref(ctx.cfg), C.byref(ps), ptr(Y), ptr(E), ptr(M_u8), ptr(dout), ptr(dY),
                                                ptr(dE), ptr(ctx.ws), ctx.ws.numel(), ptr(sc), sc.numel(), C.byref(gs),
                                                stream_ptr()), "mmf_xattn_add_backward")
        _fire(ctx.done_hook)
        return (dY, dE, None, None, None, None, None, None, None) + tuple(rets)


def _half_grads(params, owned_full, owned_part, sinks):
    """gradient buffers for one half of MMF_XAttn_Add: `params` in XAddParams order with None for parameters the half
    does not touch.  owned_part = indices (attn.in_proj_weight / _bias) of which the half writes only some rows: without a
    sink the rest of such a buffer must read as zero, because autograd adds the two halves' tensors."""
    grads, rets = _grad_buffers(params, sinks)
    for i in owned_part:
        if params[i] is not None and sinks[i] is None:
            grads[i].zero_()
    return grads, rets


class MMFXAttnKVFn(torch.autograd.Function):
    """key/value half of MMF_XAttn_Add: E_txt -> KV (B,T,2d) = (k | v) = in_proj_{k,v}(proj_{k,v}(E_txt)), computed with the
    per-step product weights.  Depends only on the text side, so it can run beside the backbone; params: (proj_k_w,
    proj_v_w, attn_in_w, attn_in_b)."""

    @staticmethod
    def forward(ctx, E, H, precision, done_hook, proj_k_w, proj_v_w, attn_in_w, attn_in_b):
        lib = _lib.load()
        E = _c(E)
        params = (None, _c(proj_k_w), _c(proj_v_w), _c(attn_in_w), _c(attn_in_b), None, None, None, None, None, None)
        _need_gpu(E, *params)
        B, T, d = E.shape
        cfg = make_cfg(B, 0, T, 1, 0, d, H, precision, False, 0.0, 0.0, 0, E.device)
        ws = _bytes(lib.immtsf_mmf_xattn_kv_workspace_bytes(C.byref(cfg)), E.device)
        KV = torch.empty(B, T, 2 * d, dtype=torch.float32, device=E.device)
        ps = _struct(XAddParams, params)
        E_h = _shadow_get(E) if _bf16_dataflow(precision, d) else None
        cfg.in_h = None if E_h is None else E_h.data_ptr()
        ctx.E_h = E_h
        check(lib.immtsf_mmf_xattn_kv_forward(C.byref(cfg), C.byref(ps), ptr(E), ptr(KV), ptr(ws), ws.numel(), stream_ptr()),
              "mmf_xattn_kv_forward")
        ctx.cfg, ctx.ws = cfg, ws
        ctx.sinks = _sinks_of(params)
        ctx.cfg.grads_prezeroed = _prezeroed(params, ctx.sinks)
        ctx.done_hook = done_hook          # the block's gradients are final once THIS half's backward has run
        ctx.save_for_backward(E, *params[1:5])
        return KV

    @staticmethod
    def backward(ctx, dKV):
        lib = _lib.load()
        E, wk, wv, win, bin_ = ctx.saved_tensors
        params = [None, wk, wv, win, bin_, None, None, None, None, None, None]
        grads, rets = _half_grads(params, (1, 2), (3, 4), ctx.sinks)
        dE = torch.empty_like(E)
        sc = _bytes(lib.immtsf_mmf_xattn_kv_scratch_bytes(C.byref(ctx.cfg)), E.device)
        ps, gs = _struct(XAddParams, params), _struct(XAddParams, grads)
        dKV = dKV.contiguous()
        cfg, dE_h = ctx.cfg, None
        if _bf16_dataflow(cfg.precision, cfg.d):
            dKV_h = _shadow_get(dKV)
            cfg.in_h = None if dKV_h is None else dKV_h.data_ptr()
            cfg.aux_h = None if ctx.E_h is None else ctx.E_h.data_ptr()
            dE_h = torch.empty(dE.shape, dtype=torch.bfloat16, device=dE.device)
            cfg.out_h = dE_h.data_ptr()
        check(lib.immtsf_mmf_xattn_kv_backward(C.byref(cfg), C.byref(ps), ptr(E), ptr(dKV), ptr(dE), ptr(ctx.ws),
                                               ctx.ws.numel(), ptr(sc), sc.numel(), C.byref(gs), stream_ptr()),
              "mmf_xattn_kv_backward")
        if dE_h is not None:
            _shadow_put(dE, dE_h)
        _fire(ctx.done_hook)
        return (dE, None, None, None) + tuple(rets[1:5])


_deferred = []          # parameter-gradient work postponed by a backward (config.defer_param_grads): closures


def run_deferred():
    """enqueue (on the current stream) the parameter-gradient work that backward passes postponed while
    `immtsf.config.defer_param_grads` was set (immtsf.train.PhasedStep moves it off the path to the backbone's backward)"""
    work = list(_deferred)
    del _deferred[:]
    for fn in work:
        fn()


def mmf_xattn_q_fold(Y_C, d, H, precision, params):
    """the query half's per-step product weights (see immtsf_mmf_xattn_q_fold): a flat fp32 tensor; no autograd -- the
    parameters' gradients come out of MMFXAttnQFn.backward by the chain rule"""
    lib = _lib.load()
    dev = params[0].device
    cfg = make_cfg(1, 0, 1, Y_C, 0, d, H, precision, False, 0.0, 0.0, 0, dev)
    fold = torch.empty(lib.immtsf_mmf_xattn_q_fold_floats(C.byref(cfg)), dtype=torch.float32, device=dev)
    ps = _struct(XAddParams, tuple(None if p is None else _c(p.detach()) for p in params))
    check(lib.immtsf_mmf_xattn_q_fold(C.byref(cfg), C.byref(ps), ptr(fold), stream_ptr()), "mmf_xattn_q_fold")
    return fold


class MMFXAttnQFn(torch.autograd.Function):
    """query half of MMF_XAttn_Add: (Y_ts, KV, M_txt) -> Y_out.  params in XAddParams order without proj_k/proj_v.
    fold: the result of mmf_xattn_q_fold (or None: formed inside the forward)."""

    @staticmethod
    def forward(ctx, Y, KV, M_u8, fold, H, kappa, p_drop, training, precision, seed, proj_q_w, attn_in_w, attn_in_b, *rest):
        lib = _lib.load()
        Y, KV, M_u8 = _c(Y), _c(KV), _c(M_u8)
        params = (_c(proj_q_w), None, None, _c(attn_in_w), _c(attn_in_b)) + tuple(_c(p) for p in rest)
        _need_gpu(Y, KV, M_u8, *params)
        B, T, Cc = Y.shape
        d = KV.shape[2] // 2
        cfg = make_cfg(B, 0, T, Cc, 0, d, H, precision, training, p_drop, kappa, seed, Y.device)
        ws = _bytes(lib.immtsf_mmf_xattn_q_workspace_bytes(C.byref(cfg)), Y.device)
        out = torch.empty_like(Y)
        ps = _struct(XAddParams, params)
        check(lib.immtsf_mmf_xattn_q_forward(C.byref(cfg), C.byref(ps), ptr(Y), ptr(KV), ptr(M_u8), ptr(fold), ptr(out), ptr(ws),
                                             ws.numel(), stream_ptr()), "mmf_xattn_q_forward")
        ctx.cfg, ctx.ws, ctx.fold = cfg, ws, fold
        ctx.sinks = _sinks_of(params)
        ctx.cfg.grads_prezeroed = _prezeroed(params, ctx.sinks)
        ctx.save_for_backward(Y, KV, M_u8, params[0], *params[3:])
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        Y, KV, M_u8, wq, *rest = ctx.saved_tensors
        params = [wq, None, None] + list(rest)
        grads, rets = _half_grads(params, (0, 5, 6, 7, 8, 9, 10), (3, 4), ctx.sinks)
        dY, dKV = torch.empty_like(Y), torch.empty_like(KV)
        sc = _bytes(lib.immtsf_mmf_xattn_q_scratch_bytes(C.byref(ctx.cfg)), Y.device)
        ps, gs = _struct(XAddParams, params), _struct(XAddParams, grads)
        # the parameter gradients can only be postponed when they go to sinks (autograd gets None for them either way)
        defer = bool(config.defer_param_grads) and all(r is None for r in rets)
        cfg, ws, fold = ctx.cfg, ctx.ws, ctx.fold
        dKV_h = None
        if _bf16_dataflow(cfg.precision, cfg.d):
            dKV_h = torch.empty(dKV.shape, dtype=torch.bfloat16, device=dKV.device)
            cfg.out_h = dKV_h.data_ptr()
        check(lib.immtsf_mmf_xattn_q_backward(C.byref(cfg), C.byref(ps), ptr(Y), ptr(KV), ptr(M_u8), ptr(fold), ptr(dout.contiguous()),
                                              ptr(dY), ptr(dKV), ptr(ws), ws.numel(), ptr(sc), sc.numel(), C.byref(gs),
                                              1 if defer else 0, stream_ptr()), "mmf_xattn_q_backward")
        cfg.out_h = None
        if dKV_h is not None:
            _shadow_put(dKV, dKV_h)
        if defer:
            keep = (Y, M_u8, params, grads)         # the closure keeps every buffer it touches alive

            def finish():
                check(lib.immtsf_mmf_xattn_q_backward_params(C.byref(cfg), C.byref(ps), ptr(keep[0]), ptr(keep[1]), ptr(fold), ptr(ws),
                                                             ws.numel(), ptr(sc), sc.numel(), C.byref(gs), stream_ptr()),
                      "mmf_xattn_q_backward_params")
            _deferred.append(finish)
        return (dY, dKV, None, None, None, None, None, None, None, None, rets[0]) + tuple(rets[3:])


# ---- MMF_XAttn_Add, low-rank form (csrc/xrank.hip): the text side is projected onto (2C+1) H columns, the attention + head is one
# kernel per direction on those columns
def mmf_xrank_pw(T, Cc, d, H):
    """row pitch of P for these dimensions, or 0 when the low-rank form does not take them (config.xattn_rank off, or limits)"""
    if not config.xattn_rank:
        return 0
    lib = _lib.load()
    cfg = make_cfg(1, 0, T, Cc, 0, d, H, 0, False, 0.0, 0.0, 0, None)
    return int(lib.immtsf_mmf_xrank_pw(C.byref(cfg)))


class MMFXRankPFn(torch.autograd.Function):
    """P half of MMF_XAttn_Add's low-rank form: E_txt -> (P (B,T,pw), b_HO (C)).  Depends only on the text side; its backward yields
    dE_txt and the gradients of every parameter except LayerNorm's.  params: XAddParams order without ln_w / ln_b."""

    @staticmethod
    def forward(ctx, E, Cc, H, precision, done_hook, *params9):
        lib = _lib.load()
        E = _c(E)
        # "_z" form: two more parameters, the producer's last linear map (proj_out.weight, .bias); E is then Z, its input
        proj = tuple(_c(p) for p in params9[9:11]) if len(params9) == 11 else None
        params9 = params9[:9]
        params = tuple(_c(p) for p in params9) + (None, None)
        _need_gpu(E, *params)
        B, T, d = E.shape
        cfg = make_cfg(B, 0, T, Cc, 0, d, H, precision, False, 0.0, 0.0, 0, E.device)
        pw = int(lib.immtsf_mmf_xrank_pw(C.byref(cfg)))
        if pw <= 0:
            raise _lib.ImmtsfError("MMF_XAttn_Add low-rank form: shape outside its limits")
        ws = _bytes(lib.immtsf_mmf_xrank_p_workspace_bytes(C.byref(cfg)), E.device)
        P = torch.empty(B, T, pw, dtype=torch.float32, device=E.device)
        bHO = torch.empty(Cc, dtype=torch.float32, device=E.device)
        ps = _struct(XAddParams, params)
        E_h = _shadow_get(E) if _bf16_dataflow(precision, d) else None
        cfg.in_h = None if E_h is None else E_h.data_ptr()
        ctx.E_h = E_h
        # the fused tail's hand-over (config.z_handover): did the producer promise to take a low-rank gradient for this Z?
        ho = _reg_take(_handover, E, pop=True) if proj is not None else None      # (one consumer: the entry has done its work)
        ctx.lowrank = False
        if ho is not None:
            if ho[1] and E_h is None:
                raise _lib.ImmtsfError("Z was handed over as its bf16 image alone (config.z_handover) but the image is gone")
            ctx.lowrank = bool(lib.immtsf_ttf_t2v_xattn_accepts_lowrank(C.byref(ho[0]), pw))
        # the fold depends on parameters only: with config.fold_stream (immtsf.train.FlagStep) it runs on that stream, beside the
        # text side's own forward, and this stream picks it up just before the projection
        L = config.fold_stream
        if L is not None:
            cur = torch.cuda.current_stream()
            ws.record_stream(L)
            bHO.record_stream(L)
            if proj is not None:
                check(lib.immtsf_mmf_xrank_fold_z(C.byref(cfg), C.byref(ps), ptr(proj[0]), ptr(proj[1]), ptr(bHO), ptr(ws), ws.numel(),
                                                  L.cuda_stream), "mmf_xrank_fold_z")
            else:
                check(lib.immtsf_mmf_xrank_fold(C.byref(cfg), C.byref(ps), ptr(bHO), ptr(ws), ws.numel(), L.cuda_stream), "mmf_xrank_fold")
            if config.fold_flag is not None:      # hand-over through a device flag (no graph edge): (flag address, time-out report address)
                check(lib.immtsf_flag_set(config.fold_flag[0], L.cuda_stream), "flag_set")
                check(lib.immtsf_flag_wait(config.fold_flag[0], config.fold_flag[1], 50, cur.cuda_stream), "flag_wait")
            else:
                cur.wait_stream(L)
        if proj is not None:
            _need_gpu(*proj)
            check(lib.immtsf_mmf_xrank_p_forward_z(C.byref(cfg), C.byref(ps), ptr(proj[0]), ptr(proj[1]), ptr(E), ptr(P), ptr(bHO), ptr(ws),
                                                   ws.numel(), 0 if L is None else 1, stream_ptr()), "mmf_xrank_p_forward_z")
        else:
            check(lib.immtsf_mmf_xrank_p_forward(C.byref(cfg), C.byref(ps), ptr(E), ptr(P), ptr(bHO), ptr(ws), ws.numel(),
                                                 0 if L is None else 1, stream_ptr()), "mmf_xrank_p_forward")
        ctx.cfg, ctx.ws = cfg, ws
        ctx.sinks = _sinks_of(params)
        ctx.done_hook = done_hook          # the block's gradients are final once THIS half's backward has run
        ctx.has_proj = proj is not None
        ctx.proj_sinks = _sinks_of(proj) if proj is not None else None
        ctx.save_for_backward(E, *params[:9], *(proj or ()))
        return P, bHO

    @staticmethod
    def backward(ctx, dP, dbHO):
        lib = _lib.load()
        E, *params9 = ctx.saved_tensors
        proj = None
        if ctx.has_proj:
            proj, params9 = params9[9:11], params9[:9]
            pgrads, prets = _grad_buffers(proj, ctx.proj_sinks)
        params = list(params9) + [None, None]
        grads, rets = _grad_buffers(params, ctx.sinks)
        dE = torch.empty_like(E)          # (low-rank hand-over: an unwritten placeholder that carries the registration below)
        cfg = ctx.cfg
        lowrank = ctx.lowrank and proj is not None and config.z_handover
        cfg.form = (cfg.form | _lib.FORM_LOWRANK_OUT) if lowrank else (cfg.form & ~_lib.FORM_LOWRANK_OUT)
        sc = _bytes(lib.immtsf_mmf_xrank_p_scratch_bytes(C.byref(cfg)), E.device)
        ps, gs = _struct(XAddParams, params), _struct(XAddParams, grads)
        dP = dP.contiguous()
        dbHO = torch.zeros(cfg.C, dtype=torch.float32, device=E.device) if dbHO is None else dbHO.contiguous()
        dE_h = None
        if _bf16_dataflow(cfg.precision, cfg.d):
            dP_h = _shadow_get(dP)
            cfg.in_h = None if dP_h is None else dP_h.data_ptr()
            cfg.aux_h = None if ctx.E_h is None else ctx.E_h.data_ptr()
            if proj is None:        # ("_z" form: the producer's backward reads dZ as fp32 rows only)
                dE_h = torch.empty(dE.shape, dtype=torch.bfloat16, device=dE.device)
                cfg.out_h = dE_h.data_ptr()
            else:
                cfg.out_h = None

        ws0 = ctx.ws

        def pre(stream):           # "_z" form: the parameter-only step in front of the chain (dW_fold from dWc; proj_out's gradients)
            if proj is not None:
                check(lib.immtsf_mmf_xrank_p_backward_pre_z(C.byref(cfg), ptr(proj[0]), ptr(proj[1]), ptr(ws0), ws0.numel(), ptr(sc), sc.numel(),
                                                            ptr(pgrads[0]), ptr(pgrads[1]), stream), "mmf_xrank_p_backward_pre_z")

        def data_half(part=0, stream=None):
            # part 0: dZ and the chain's seeds; "_z" form only: BWD_PHASE_A = dZ alone, BWD_WGRAD_A = the seeds (dWc, dbc) alone
            if proj is not None:
                cfg.bwd_phase = part
                check(lib.immtsf_mmf_xrank_p_backward_data_z(C.byref(cfg), C.byref(ps), ptr(proj[0]), ptr(proj[1]), ptr(E), ptr(dP), ptr(dE),
                                                             ptr(ctx.ws), ctx.ws.numel(), ptr(sc), sc.numel(),
                                                             stream_ptr() if stream is None else stream),
                      "mmf_xrank_p_backward_data_z")
                cfg.bwd_phase = 0
            else:
                check(lib.immtsf_mmf_xrank_p_backward_data(C.byref(cfg), C.byref(ps), ptr(E), ptr(dP), ptr(dE), ptr(ctx.ws), ctx.ws.numel(), ptr(sc),
                                                           sc.numel(), stream_ptr()), "mmf_xrank_p_backward_data")
        tail = config.param_tail
        if tail is not None and all(r is None for r in rets[:9]) and tail["defer"] > 0:
            # every gradient goes straight to its sink: the last `defer` launches of the parameter chain are left to the other branch
            k = 3 - min(3, int(tail["defer"]))
            # the "_z" form with the whole chain deferred: this stream (the text side's dependent chain) only forms dZ; the chain's
            # seeds dWc = dP^T Z -- parameter-gradient work too -- open the other branch's jobs
            seeds_later = proj is not None and k == 0
            data_half(_lib.BWD_PHASE_A if seeds_later else 0)
            if seeds_later:
                tail["jobs"].append(lambda stream, keep=(E, dP): data_half(_lib.BWD_WGRAD_A, stream))
            ws = ctx.ws

            def run_params(stream, first, last, cfg=cfg, ps=ps, gs=gs, dbHO=dbHO, ws=ws, sc=sc, keep=(params, grads)):
                check(lib.immtsf_mmf_xrank_p_backward_params(C.byref(cfg), C.byref(ps), ptr(dbHO), ptr(ws), ws.numel(), ptr(sc), sc.numel(),
                                                             C.byref(gs), first, last, stream), "mmf_xrank_p_backward_params")

            if k > 0:
                pre(stream_ptr())
                run_params(stream_ptr(), 0, k)
            check(lib.immtsf_flag_set(tail["flag"][0], stream_ptr()), "flag_set")
            tail["flag_set"] = True          # (the branch that runs the jobs waits for the flag only when somebody set it)
            if k > 0:
                tail["jobs"].append(lambda stream: run_params(stream, k, 3))
            else:
                tail["jobs"].append(lambda stream: (pre(stream), run_params(stream, 0, 3)))
            # the block's gradients are final behind the LAST of those launches: its hook (data parallel: the bucket's announcement)
            # fires there, on the stream that runs them
            hook, ctx.done_hook = ctx.done_hook, None
            tail["jobs"].append(lambda stream, hook=hook: _fire(hook))
        else:
            data_half()
            pre(stream_ptr())
            check(lib.immtsf_mmf_xrank_p_backward_params(C.byref(cfg), C.byref(ps), ptr(dbHO), ptr(ctx.ws), ctx.ws.numel(), ptr(sc), sc.numel(),
                                                         C.byref(gs), 0, 3, stream_ptr()), "mmf_xrank_p_backward_params")
        if dE_h is not None:
            _shadow_put(dE, dE_h)
        if lowrank:
            basis, rank = C.c_void_p(), C.c_int32()
            check(lib.immtsf_mmf_xrank_lowrank_basis(C.byref(cfg), ptr(ctx.ws), ctx.ws.numel(), C.byref(basis), C.byref(rank)),
                  "mmf_xrank_lowrank_basis")
            _reg_put(_lowrank, dE, (_lib.LowRankGrad(dP.data_ptr(), basis.value, rank.value, dP.shape[-1]), dP, ctx.ws))
        _fire(ctx.done_hook)
        return (dE, None, None, None, None) + tuple(rets[:9]) + (tuple(prets) if proj is not None else ())


class MMFXRankQFn(torch.autograd.Function):
    """Q half of the low-rank form: (Y_ts, P, b_HO, M_txt, ln_w, ln_b) -> Y_out; the serial section between the backbone's forward
    and backward (one kernel per direction)."""

    @staticmethod
    def forward(ctx, Y, P, bHO, M_u8, d, H, kappa, p_drop, training, precision, seed, ln_w, ln_b):
        lib = _lib.load()
        Y, P, bHO, M_u8, ln_w, ln_b = _c(Y), _c(P), _c(bHO), _c(M_u8), _c(ln_w), _c(ln_b)
        _need_gpu(Y, P, bHO, M_u8, ln_w, ln_b)
        B, T, Cc = Y.shape
        cfg = make_cfg(B, 0, T, Cc, 0, d, H, precision, training, p_drop, kappa, seed, Y.device)
        if int(lib.immtsf_mmf_xrank_pw(C.byref(cfg))) != P.shape[2]:
            raise _lib.ImmtsfError("MMF_XAttn_Add low-rank form: P does not have the row pitch of these dimensions")
        ws = _bytes(lib.immtsf_mmf_xrank_q_workspace_bytes(C.byref(cfg)), Y.device)
        out = torch.empty_like(Y)
        check(lib.immtsf_mmf_xrank_q_forward(C.byref(cfg), ptr(ln_w), ptr(ln_b), ptr(Y), ptr(P), ptr(bHO), ptr(M_u8), ptr(out), ptr(ws),
                                             ws.numel(), stream_ptr()), "mmf_xrank_q_forward")
        ctx.cfg, ctx.ws = cfg, ws
        ctx.sinks = _sinks_of((ln_w, ln_b))
        ctx.save_for_backward(Y, P, M_u8, ln_w, ln_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        Y, P, M_u8, ln_w, ln_b = ctx.saved_tensors
        grads, rets = _grad_buffers((ln_w, ln_b), ctx.sinks)
        dY, dP = torch.empty_like(Y), torch.empty_like(P)
        dbHO = torch.empty(Y.shape[2], dtype=torch.float32, device=Y.device)
        cfg, dP_h = ctx.cfg, None
        if _bf16_dataflow(cfg.precision, cfg.d):
            dP_h = torch.empty(dP.shape, dtype=torch.bfloat16, device=dP.device)
            cfg.out_h = dP_h.data_ptr()
        check(lib.immtsf_mmf_xrank_q_backward(C.byref(cfg), ptr(ln_w), ptr(Y), ptr(P), ptr(M_u8), ptr(dout.contiguous()), ptr(dY), ptr(dP),
                                              ptr(dbHO), ptr(grads[0]), ptr(grads[1]), ptr(ctx.ws), ctx.ws.numel(), stream_ptr()),
              "mmf_xrank_q_backward")
        cfg.out_h = None
        if dP_h is not None:
            _shadow_put(dP, dP_h)
        return (dY, dP, dbHO, None, None, None, None, None, None, None, None) + tuple(rets)


_xq_tickets = {}


def _xq_ticket(device) -> torch.Tensor:
    """the zero-initialised ticket word of immtsf_mmf_xrank_q_train: one per device for the life of the process (a captured graph keeps
    pointing at it; the call leaves it zero; one training step per device at a time)"""
    key = (device.type, device.index)
    t = _xq_tickets.get(key)
    if t is None:
        t = _xq_tickets[key] = torch.zeros(64, dtype=torch.int32, device=device)
    return t


class MMFXRankQLossFn(torch.autograd.Function):
    """Q half of the low-rank form + masked MSE (immtsf.ops.masked_mse with global_cnt) + the backward of both, ONE launch: the
    forward computes the loss and every gradient (seeded with 1); backward() hands them on (scaled unless the seed is
    backward_unit's).  (Y_ts, P, b_HO, M_txt, truth, mask, cnt, ln_w, ln_b) -> loss."""

    @staticmethod
    def forward(ctx, Y, P, bHO, M_u8, truth, mask, cnt, d, H, kappa, p_drop, training, precision, seed, ln_w, ln_b):
        lib = _lib.load()
        # the early "dY_ts is ready" flag is only sound when autograd hands THIS call's dY buffer on as Y's gradient: a
        # non-contiguous Y makes AccumulateGrad clone it into Y's layout (a copy kernel behind this one)
        y_dense = Y.is_contiguous()
        Y, P, bHO, M_u8, ln_w, ln_b = _c(Y), _c(P), _c(bHO), _c(M_u8), _c(ln_w), _c(ln_b)
        truth, mask, cnt = _c(truth), _c(mask), _c(cnt.to(torch.float32))
        _need_gpu(Y, P, bHO, M_u8, truth, mask, cnt, ln_w, ln_b)
        B, T, Cc = Y.shape
        cfg = make_cfg(B, 0, T, Cc, 0, d, H, precision, training, p_drop, kappa, seed, Y.device)
        if int(lib.immtsf_mmf_xrank_pw(C.byref(cfg))) != P.shape[2]:
            raise _lib.ImmtsfError("MMF_XAttn_Add low-rank form: P does not have the row pitch of these dimensions")
        sc = _bytes(lib.immtsf_mmf_xrank_q_train_scratch_bytes(C.byref(cfg)), Y.device)
        sinks = _sinks_of((ln_w, ln_b))
        grads, rets = _grad_buffers((ln_w, ln_b), sinks)
        dY, dP = torch.empty_like(Y), torch.empty_like(P)
        dbHO = torch.empty(Cc, dtype=torch.float32, device=Y.device)
        loss = torch.empty((), dtype=torch.float32, device=Y.device)
        dP_h = None
        if _bf16_dataflow(cfg.precision, cfg.d):
            dP_h = torch.empty(dP.shape, dtype=torch.bfloat16, device=dP.device)
            cfg.out_h = dP_h.data_ptr()
        # immtsf.train.FlagStep: the device flag the backbone's stream waits on before its backward is published by this kernel as
        # soon as dY_ts is complete (config.head_done_flag is consumed: the caller sees None and skips its own flag_set)
        flag = None
        if y_dense and config.head_done_flag is not None:
            flag, config.head_done_flag = config.head_done_flag, None
            config.head_dy_ptr = dY.data_ptr()
        check(lib.immtsf_mmf_xrank_q_train(C.byref(cfg), ptr(ln_w), ptr(ln_b), ptr(Y), ptr(P), ptr(bHO), ptr(M_u8), ptr(truth), ptr(mask),
                                           ptr(cnt), 1.0, None, ptr(loss), ptr(dY), ptr(dP), ptr(dbHO), ptr(grads[0]), ptr(grads[1]),
                                           ptr(sc), sc.numel(), ptr(_xq_ticket(Y.device)), flag, stream_ptr()), "mmf_xrank_q_train")
        if dP_h is not None:
            _shadow_put(dP, dP_h)
        ctx.grads = (dY, dP, dbHO) + tuple(rets)
        # LayerNorm's two gradients went straight to FlatTrainer's sinks (rets None): they were written with the unit seed, so a
        # backward with any other seed (loss scaling, loss * k) has to rescale them where they lie
        ctx.sunk = tuple(b for b, r in zip(grads, rets) if r is None)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        g = ctx.grads
        ctx.grads = None
        if not is_unit_grad(dloss):
            g = tuple(None if t is None else t * dloss for t in g)
            for b in ctx.sunk:           # (one writer per sink and step -- the sink contract -- so scaling in place is exact)
                b.mul_(dloss)
        ctx.sunk = ()
        return (g[0], g[1], g[2]) + (None,) * 11 + (g[3], g[4])


# ------------------------------------------------------------------------------------------------ MMF_GR_Add
class MMFGRAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Y, E, M_u8, hidden, p_drop, training, precision, seed, *params):
        lib = _lib.load()
        Y, E, M_u8 = _c(Y), _c(E), _c(M_u8)
        params = tuple(_c(p) for p in params)
        _need_gpu(Y, E, M_u8, *params)
        B, T, Cc = Y.shape
        d = E.shape[2]
        cfg = make_cfg(B, 0, T, Cc, 0, d, 1, precision, training, p_drop, 0.0, seed, Y.device)
        ws = _bytes(lib.immtsf_mmf_gr_add_workspace_bytes(C.byref(cfg), hidden), Y.device)
        out = torch.empty_like(Y)
        ps = _struct(GRParams, params)
        check(lib.immtsf_mmf_gr_add_forward(C.byref(cfg), hidden, C.byref(ps), ptr(Y), ptr(E), ptr(M_u8), ptr(out), ptr(ws),
                                            ws.numel(), stream_ptr()), "mmf_gr_add_forward")
        ctx.cfg, ctx.ws, ctx.hidden = cfg, ws, hidden
        ctx.sinks = _sinks_of(params)
        ctx.cfg.grads_prezeroed = _prezeroed(params, ctx.sinks)
        ctx.done_hook = getattr(params[0], "_immtsf_bwd_hook", None)
        ctx.save_for_backward(Y, E, M_u8, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        Y, E, M_u8, *params = ctx.saved_tensors
        dout = dout.contiguous()
        grads, rets = _grad_buffers(params, ctx.sinks)
        dY, dE = torch.empty_like(Y), torch.empty_like(E)
        sc = _bytes(lib.immtsf_mmf_gr_add_scratch_bytes(C.byref(ctx.cfg), ctx.hidden), Y.device)
        ps, gs = _struct(GRParams, params), _struct(GRParams, grads)
        check(lib.immtsf_mmf_gr_add_backward(C.byref(ctx.cfg), ctx.hidden, C.byref(ps), ptr(Y), ptr(E), ptr(M_u8), ptr(dout),
                                             ptr(dY), ptr(dE), ptr(ctx.ws), ctx.ws.numel(), ptr(sc), sc.numel(), C.byref(gs),
                                             stream_ptr()), "mmf_gr_add_backward")
        _fire(ctx.done_hook)
        return (dY, dE, None, None, None, None, None, None) + tuple(rets)


def gr_split_pw(T, Cc, d, hidden):
    """row pitch of MMF_GR_Add's split-form P for these dimensions, 0 where the split form does not apply (immtsf_mmf_gr_pw)"""
    cfg = make_cfg(1, 0, int(T), int(Cc), 0, int(d), 1, 0, False, 0.0, 0.0, 0, None)
    return int(_lib.load().immtsf_mmf_gr_pw(C.byref(cfg), int(hidden)))


class MMFGRPFn(torch.autograd.Function):
    """P half of MMF_GR_Add's split form (csrc/gr_train.hip): E_txt -> P (B, T, pw), the text columns of the GRU's input map and of the
    gate net (+ both biases).  Depends on the text side only; its backward yields dE_txt and the text columns of d W_ih / d W_g, d b_ih,
    d b_g (the Y columns come out of MMFGRQLossFn).  params: GRParams order."""

    @staticmethod
    def forward(ctx, E, Cc, hidden, precision, *params):
        lib = _lib.load()
        E = _c(E)
        params = tuple(_c(p) for p in params)
        _need_gpu(E, *params)
        B, T, d = E.shape
        cfg = make_cfg(B, 0, T, Cc, 0, d, 1, precision, False, 0.0, 0.0, 0, E.device)
        pw = int(lib.immtsf_mmf_gr_pw(C.byref(cfg), hidden))
        if pw <= 0:
            raise _lib.ImmtsfError("MMF_GR_Add split form: shape outside its limits")
        ws = _bytes(lib.immtsf_mmf_gr_p_workspace_bytes(C.byref(cfg), hidden), E.device)
        P = torch.empty(B, T, pw, dtype=torch.float32, device=E.device)
        E_h = _shadow_get(E) if _bf16_dataflow(precision, d) else None
        cfg.in_h = None if E_h is None else E_h.data_ptr()
        ps = _struct(GRParams, params)
        check(lib.immtsf_mmf_gr_p_forward(C.byref(cfg), hidden, C.byref(ps), ptr(E), ptr(P), ptr(ws), ws.numel(), stream_ptr()),
              "mmf_gr_p_forward")
        cfg.in_h = None
        ctx.cfg, ctx.ws, ctx.hidden, ctx.E_h = cfg, ws, hidden, E_h
        ctx.sinks = _sinks_of(params)
        ctx.save_for_backward(E, *params)
        return P

    @staticmethod
    def backward(ctx, dP):
        lib = _lib.load()
        E, *params = ctx.saved_tensors
        cfg = ctx.cfg
        dP = dP.contiguous()
        # the four gradients this half owns: w_ih (0), b_ih (2), gate_w (6), gate_b (7) -- the text columns of the two matrices only: a
        # sink is written in place (the other half adds its columns to the same buffer), otherwise a zero-filled full-size tensor each
        own = (0, 2, 6, 7)
        bufs, rets = [None] * 10, [None] * 10
        for i in own:
            if ctx.sinks[i] is not None:
                bufs[i] = ctx.sinks[i]
            else:
                bufs[i] = rets[i] = torch.zeros_like(params[i])
        dE = torch.empty_like(E)
        sc = _bytes(lib.immtsf_mmf_gr_p_scratch_bytes(C.byref(cfg), ctx.hidden), E.device)
        ps, gs = _struct(GRParams, params), _struct(GRParams, bufs)
        dE_h = None
        if _bf16_dataflow(cfg.precision, cfg.d):
            cfg.aux_h = None if ctx.E_h is None else ctx.E_h.data_ptr()
            dE_h = torch.empty(dE.shape, dtype=torch.bfloat16, device=dE.device)
            cfg.out_h = dE_h.data_ptr()
        check(lib.immtsf_mmf_gr_p_backward(C.byref(cfg), ctx.hidden, C.byref(ps), ptr(E), ptr(dP), ptr(dE), ptr(ctx.ws), ctx.ws.numel(), ptr(sc),
                                           sc.numel(), C.byref(gs), stream_ptr()), "mmf_gr_p_backward")
        cfg.aux_h = cfg.out_h = None
        if dE_h is not None:
            _shadow_put(dE, dE_h)
        _fire(getattr(params[0], "_immtsf_bwd_hook", None))      # (the block's gradients are final once THIS half's backward has run)
        return (dE, None, None, None) + tuple(rets)


class MMFGRQLossFn(torch.autograd.Function):
    """the Y half of MMF_GR_Add's split form + masked MSE (immtsf.ops.masked_mse with global_cnt) + the backward of both, ONE launch
    (immtsf_mmf_gr_q_train): the forward computes the loss and every gradient (seeded with 1); backward() hands them on.
    (Y_ts, P, M_txt, truth, mask, cnt, params...) -> loss."""

    @staticmethod
    def forward(ctx, Y, P, M_u8, truth, mask, cnt, hidden, p_drop, training, precision, seed, *params):
        lib = _lib.load()
        y_dense = Y.is_contiguous()
        Y, P, M_u8 = _c(Y), _c(P), _c(M_u8)
        truth, mask, cnt = _c(truth), _c(mask), _c(cnt.to(torch.float32))
        params = tuple(_c(p) for p in params)
        _need_gpu(Y, P, M_u8, truth, mask, cnt, *params)
        B, T, Cc = Y.shape
        d = params[0].shape[1] - Cc
        cfg = make_cfg(B, 0, T, Cc, 0, d, 1, precision, training, p_drop, 0.0, seed, Y.device)
        if int(lib.immtsf_mmf_gr_pw(C.byref(cfg), hidden)) != P.shape[2]:
            raise _lib.ImmtsfError("MMF_GR_Add split form: P does not have the row pitch of these dimensions")
        sinks = _sinks_of(params)
        # the eight gradients this half owns (w_ih and gate_w: their Y columns), ADDED into zeroed buffers: pre-zeroed sinks as they are
        own = (0, 1, 3, 4, 5, 6, 8, 9)
        bufs, rets = [None] * 10, [None] * 10
        for i in own:
            if sinks[i] is not None and getattr(params[i], "_immtsf_grad_prezeroed", False):
                bufs[i] = sinks[i]
            else:
                bufs[i] = rets[i] = torch.zeros_like(params[i])
        dY, dP = torch.empty_like(Y), torch.empty_like(P)
        loss = torch.empty((), dtype=torch.float32, device=Y.device)
        scratch = torch.empty(B, dtype=torch.float32, device=Y.device)
        flag = None
        if y_dense and config.head_done_flag is not None:      # immtsf.train.FlagStep: dY_ts is published by the kernel itself
            flag, config.head_done_flag = config.head_done_flag, None
            config.head_dy_ptr = dY.data_ptr()
        ps, gs = _struct(GRParams, params), _struct(GRParams, bufs)
        check(lib.immtsf_mmf_gr_q_train(C.byref(cfg), hidden, C.byref(ps), ptr(Y), ptr(P), ptr(M_u8), ptr(truth), ptr(mask), ptr(cnt), 1.0, None,
                                        ptr(loss), ptr(dY), ptr(dP), C.byref(gs), ptr(scratch), ptr(_xq_ticket(Y.device)[8:]), flag, stream_ptr()),
              "mmf_gr_q_train")
        ctx.grads = (dY, dP) + tuple(rets)
        ctx.sunk = tuple(bufs[i] for i in own if rets[i] is None)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        g = ctx.grads
        ctx.grads = None
        if not is_unit_grad(dloss):
            g = tuple(None if t is None else t * dloss for t in g)
            for b in ctx.sunk:
                b.mul_(dloss)
        ctx.sunk = ()
        return (g[0], g[1]) + (None,) * 9 + tuple(g[2:])


class MMFGRQFn(torch.autograd.Function):
    """the Y half of the split form, output only (inference / no_grad: immtsf_mmf_gr_q_train with truth == NULL)."""

    @staticmethod
    def forward(ctx, Y, P, M_u8, hidden, precision, *params):
        lib = _lib.load()
        Y, P, M_u8 = _c(Y), _c(P), _c(M_u8)
        params = tuple(_c(p) for p in params)
        _need_gpu(Y, P, M_u8, *params)
        B, T, Cc = Y.shape
        cfg = make_cfg(B, 0, T, Cc, 0, params[0].shape[1] - Cc, 1, precision, False, 0.0, 0.0, 0, Y.device)
        out = torch.empty_like(Y)
        ps = _struct(GRParams, params)
        check(lib.immtsf_mmf_gr_q_train(C.byref(cfg), hidden, C.byref(ps), ptr(Y), ptr(P), ptr(M_u8), None, None, None, 0.0, ptr(out), None, None,
                                        None, None, None, None, None, stream_ptr()), "mmf_gr_q_train")
        return out

    @staticmethod
    def backward(ctx, dout):
        raise RuntimeError("MMFGRQFn is the inference form of MMF_GR_Add's split path: training goes through MMFGRQLossFn or MMFGRAddFn")


# ------------------------------------------------------------------------------------------------ tPatchGNN TE + TTCN
class TTCNPatchEncodeFn(torch.autograd.Function):
    """(x, tt, mask: (P,L)) -> (P, ttcn_dim) or, with_flag, (P, ttcn_dim + 1) whose last column is the patch-non-empty
    flag (any mask > 0; no gradient).  params in immtsf_ttcn_params order."""

    @staticmethod
    def forward(ctx, x, tt, mask, precision, with_flag, *params):
        lib = _lib.load()
        x, tt, mask = _c(x), _c(tt), _c(mask)
        params = tuple(_c(p) for p in params)
        _need_gpu(x, tt, mask, *params)
        P, L = x.shape
        te_dim = params[2].numel() + 1
        K = params[10].numel()
        ld = K + 1 if with_flag else K
        out = torch.empty(P, ld, dtype=torch.float32, device=x.device)
        ws = _bytes(lib.immtsf_ttcn_workspace_bytes(P, L, te_dim, K), x.device)
        ps = _struct(TTCNParams, params)
        check(lib.immtsf_ttcn_forward(P, L, te_dim, K, precision, ptr(x), ptr(tt), ptr(mask), C.byref(ps), ptr(out), ld,
                                      K if with_flag else -1, ptr(ws), ws.numel(), stream_ptr()), "ttcn_forward")
        ctx.dims = (P, L, te_dim, K, precision, ld)
        ctx.ws = ws
        ctx.sinks = _sinks_of(params)
        _claim_sinks(params[:4], ctx.sinks[:4], "ttcn_patch_encode")
        ctx.save_for_backward(x, tt, mask, out, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, tt, mask, out, *params = ctx.saved_tensors
        P, L, te_dim, K, precision, ld = ctx.dims
        dout = dout.contiguous()
        te_acc = _shared(params[:4], ctx.sinks[:4])       # time-embedding parameters shared with the decoder's LearnableTE
        grads, rets = _grad_buffers(params, ctx.sinks)
        sc = _bytes(lib.immtsf_ttcn_scratch_bytes(P, L, te_dim, K), x.device)
        ps, gs = _struct(TTCNParams, params), _struct(TTCNParams, grads)
        g = config.sched_gate
        if g is not None and config.sched_armed is not None and config.sched_armed != torch.cuda.current_stream().cuda_stream:
            # parameter gradients only, chip-filling, nothing but the optimizer waits for them: go behind the text side's row-bound
            # backward kernels (a hint: the spin gives up after 50 ms and the call proceeds)
            check(lib.immtsf_flag_wait(g[0], g[1], 50, stream_ptr()), "flag_wait")
        check(lib.immtsf_ttcn_backward(P, L, te_dim, K, precision, ptr(x), ptr(tt), ptr(mask), C.byref(ps), ptr(out), ptr(dout), ld,
                                       ptr(ctx.ws), ctx.ws.numel(), ptr(sc), sc.numel(), C.byref(gs), 1 if te_acc else 0, stream_ptr()),
              "ttcn_backward")
        return (None, None, None, None, None) + tuple(rets)


def ttcn_patch_encode(x, tt, mask, te_scale_w, te_scale_b, te_per_w, te_per_b, W1, b1, W2, b2, W3, b3, T_bias, precision=None,
                      with_flag=False):
    """Fused LearnableTE + TTCN of tPatchGNN (models/tPatchGNN.py:176-195) on (P, L) patch tensors; with_flag appends
    the patch-non-empty column of :268-270."""
    return TTCNPatchEncodeFn.apply(x.float(), tt.float(), mask.float(), config.precision_code(precision), bool(with_flag),
                                   te_scale_w, te_scale_b, te_per_w, te_per_b, W1, b1, W2, b2, W3, b3, T_bias)


def patch_flatten3(X, tt, mask):
    """X, tt, mask (B, M, L, N) data tensors (no gradient) -> three (B*N*M, L) tensors, the patch rows of tPatchGNN's encoder
    (models/tPatchGNN.py:271-275), in one launch instead of three permute copies."""
    lib = _lib.load()
    X, tt, mask = _c(X.float()), _c(tt.float()), _c(mask.float())
    _need_gpu(X, tt, mask)
    B, M, L, N = X.shape
    out = torch.empty(3, B * N * M, L, dtype=torch.float32, device=X.device)
    check(lib.immtsf_patch_flatten3(ptr(X), ptr(tt), ptr(mask), B, M, L, N, ptr(out[0]), ptr(out[1]), ptr(out[2]), stream_ptr()),
          "patch_flatten3")
    return out[0], out[1], out[2]


class GCNAdaptiveFn(torch.autograd.Function):
    """tPatchGNN's adaptive-graph stage on (B,N,M,D): one workgroup per (window, patch) cell, params in
    immtsf_gcn_params order.  Backward recomputes the cell in LDS; parameter gradients accumulate by atomics into one
    zeroed flat buffer whose slices are returned."""

    @staticmethod
    def forward(ctx, x, order, *params):
        lib = _lib.load()
        x = _c(x)
        params = tuple(_c(p) for p in params)
        _need_gpu(x, *params)
        B, N, M, D = x.shape
        nd = params[0].shape[1]
        out = torch.empty_like(x)
        ps = _struct(GCNParams, params)
        ctx.dims = (B, N, M, D, nd, order)
        ctx.sinks = _sinks_of(params)
        if any(ctx.needs_input_grad):
            # a training forward leaves every cell's intermediates for the backward (5 KB per cell) instead of having it recompute them
            saved = torch.empty(lib.immtsf_tpatchgnn_gcn_saved_floats(B, N, M, D, nd, order), dtype=torch.float32, device=x.device)
            check(lib.immtsf_tpatchgnn_gcn_forward_saved(B, N, M, D, nd, order, ptr(x), C.byref(ps), ptr(out), ptr(saved), stream_ptr()),
                  "tpatchgnn_gcn_forward_saved")
            ctx.save_for_backward(saved, *params)
        else:
            check(lib.immtsf_tpatchgnn_gcn_forward(B, N, M, D, nd, order, ptr(x), C.byref(ps), ptr(out), stream_ptr()),
                  "tpatchgnn_gcn_forward")
            ctx.save_for_backward(x, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        saved, *params = ctx.saved_tensors
        B, N, M, D, nd, order = ctx.dims
        dout = dout.contiguous()
        dx = torch.empty_like(dout)
        grads, rets = _zeroed_grad_buffers(params, ctx.sinks)
        ps, gs = _struct(GCNParams, params), _struct(GCNParams, grads)
        check(lib.immtsf_tpatchgnn_gcn_backward_saved(B, N, M, D, nd, order, ptr(saved), C.byref(ps), ptr(dout), ptr(dx), C.byref(gs),
                                                      stream_ptr()), "tpatchgnn_gcn_backward_saved")
        return (dx, None) + tuple(rets)


def gcn_adaptive_supported(N, D, nd, order):
    return _lib.load().immtsf_tpatchgnn_gcn_lds_bytes(N, D, nd, order) > 0


def gcn_adaptive(x, order, nodevec1, nodevec2, gate1, gate2, lin1, lin2, mlp_conv):
    """x (B,N,M,D) -> (B,N,M,D); gate*: nn.Linear(D+nd,1), lin*: nn.Linear(D,nd), mlp_conv: the 1x1 nn.Conv2d."""
    return GCNAdaptiveFn.apply(x.float(), int(order), nodevec1, nodevec2, gate1.weight, gate1.bias, gate2.weight, gate2.bias,
                               lin1.weight, lin1.bias, lin2.weight, lin2.bias, mlp_conv.weight, mlp_conv.bias)


class EncoderLayerFn(torch.autograd.Function):
    """immtsf_encoder_layer_forward/backward: nn.TransformerEncoderLayer (post-norm, relu, batch_first) over short sequences,
    params in immtsf_encoder_layer_params order.  7 launches forward, 15 backward (csrc/encoder_layer.hip)."""

    @staticmethod
    def forward(ctx, x, H, p_attn, p_drop, eps, training, precision, seed, site_base, *params):
        lib = _lib.load()
        x = _c(x)
        params = tuple(_c(p) for p in params)
        _need_gpu(x, *params)
        Bs, S, D = x.shape
        F = params[6].shape[0]
        p_on = training and (p_attn > 0 or p_drop > 0)
        cfg = _lib.EncoderLayerCfg(Bs, S, D, int(H), F, precision, 1 if training else 0, float(p_attn), float(p_drop), float(eps),
                                   int(seed) & 0xFFFFFFFFFFFFFFFF, config.dropout_counter_ptr(x.device) if p_on else None,
                                   int(site_base), 0)
        ws = _bytes(lib.immtsf_encoder_layer_workspace_bytes(C.byref(cfg)), x.device)
        out = torch.empty_like(x)
        ps = _struct(_lib.EncoderLayerParams, params)
        check(lib.immtsf_encoder_layer_forward(C.byref(cfg), C.byref(ps), ptr(x), ptr(out), ptr(ws), ws.numel(), stream_ptr()),
              "encoder_layer_forward")
        ctx.cfg, ctx.ws = cfg, ws
        ctx.sinks = _sinks_of(params)
        ctx.save_for_backward(x, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, *params = ctx.saved_tensors
        cfg = ctx.cfg
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        grads, rets = _grad_buffers(params, ctx.sinks)
        cfg.grads_prezeroed = _prezeroed(params, ctx.sinks)
        sc = _bytes(lib.immtsf_encoder_layer_scratch_bytes(C.byref(cfg)), x.device)
        ps, gs = _struct(_lib.EncoderLayerParams, params), _struct(_lib.EncoderLayerParams, grads)
        check(lib.immtsf_encoder_layer_backward(C.byref(cfg), C.byref(ps), ptr(x), ptr(dout), ptr(dx), ptr(ctx.ws), ctx.ws.numel(),
                                                ptr(sc), sc.numel(), C.byref(gs), stream_ptr()), "encoder_layer_backward")
        ctx.ws = None
        return (dx,) + (None,) * 8 + tuple(rets)


def encoder_layer_supported(S, D, H):
    return S <= 8 and D % 4 == 0 and D <= 1024 and D % H == 0 and (D // H) <= 64 and (D // H) % 4 == 0


def encoder_layer(lyr, x, training, site_base, precision=None):
    """lyr: an nn.TransformerEncoderLayer (post-norm, relu, batch_first); x (Bs, S, D)"""
    at = lyr.self_attn
    p_attn, p_drop = float(at.dropout), float(lyr.dropout.p)
    seed = config.next_seed() if training and (p_attn > 0 or p_drop > 0) else 0
    return EncoderLayerFn.apply(x.float(), at.num_heads, p_attn, p_drop, lyr.norm1.eps, bool(training), config.precision_code(precision),
                                seed, site_base, at.in_proj_weight, at.in_proj_bias, at.out_proj.weight, at.out_proj.bias,
                                lyr.norm1.weight, lyr.norm1.bias, lyr.linear1.weight, lyr.linear1.bias, lyr.linear2.weight,
                                lyr.linear2.bias, lyr.norm2.weight, lyr.norm2.bias)


class ResidualLayerNormFn(torch.autograd.Function):
    """out = LayerNorm(x + Dropout(branch)) in one kernel per direction (+ the two-parameter column sums backward)."""

    @staticmethod
    def forward(ctx, x, branch, gamma, beta, eps, p_drop, training, seed, site):
        lib = _lib.load()
        x2, b2 = _c(x).reshape(-1, x.shape[-1]), _c(branch).reshape(-1, x.shape[-1])
        _need_gpu(x2, b2, gamma, beta)
        rows, d = x2.shape
        p = float(p_drop) if training else 0.0
        cnt = config.dropout_counter_ptr(x.device) if p > 0 else None
        xhat = torch.empty_like(x2)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        check(lib.immtsf_residual_layernorm_forward(ptr(x2), ptr(b2), rows, d, ptr(gamma), ptr(beta), float(eps), 1 if training else 0, p,
                                                    seed, site, cnt, ptr(xhat), ptr(rstd), ptr(out), stream_ptr()), "residual_layernorm_forward")
        ctx.save_for_backward(xhat, rstd, gamma)
        ctx.cfg = (rows, d, 1 if training else 0, p, seed, site, cnt, x.shape)
        ctx.sinks = _sinks_of((gamma, beta))
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        xhat, rstd, gamma = ctx.saved_tensors
        rows, d, tr, p, seed, site, cnt, shape = ctx.cfg
        g = dout.contiguous()
        dx, db = torch.empty_like(xhat), torch.empty_like(xhat)
        (dgamma, dbeta), rets = _grad_buffers((gamma, gamma), ctx.sinks)
        scratch = torch.empty(64 * (d + 8), dtype=torch.float32, device=g.device)
        check(lib.immtsf_residual_layernorm_backward(ptr(g), rows, d, ptr(gamma), ptr(xhat), ptr(rstd), tr, p, seed, site, cnt, ptr(dx),
                                                     ptr(db), ptr(dgamma), ptr(dbeta), ptr(scratch), stream_ptr()), "residual_layernorm_backward")
        return dx.view(shape), db.view(shape), rets[0], rets[1], None, None, None, None, None


def residual_layernorm_supported(d):
    return d % 4 == 0 and d <= 1024


def residual_layer_norm(x, branch, norm, p_drop, training, site):
    """norm: an nn.LayerNorm; x, branch (..., d)"""
    seed = config.next_seed() if training and p_drop > 0 else 0
    return ResidualLayerNormFn.apply(x.float(), branch.float(), norm.weight, norm.bias, norm.eps, float(p_drop), bool(training), seed, site)


class FFNBlockFn(torch.autograd.Function):
    """immtsf_ffn_block_forward/backward: out = LayerNorm(x + Dropout(W2 Dropout(act(W1 x + b1)) + b2)); W1 / W2 may be the
    (out, in, 1) weights of kernel-size-1 nn.Conv1d modules (same memory as (out, in))."""

    @staticmethod
    def forward(ctx, x, act, p_drop, eps, training, precision, seed, site_base, *params):
        lib = _lib.load()
        x2 = _c(x).reshape(-1, x.shape[-1])
        params = tuple(_c(p) for p in params)
        _need_gpu(x2, *params)
        R, D = x2.shape
        F = params[0].shape[0]
        p = float(p_drop) if training else 0.0
        cfg = _lib.FFNBlockCfg(R, D, F, int(act), precision, 1 if training else 0, p, float(eps), int(seed) & 0xFFFFFFFFFFFFFFFF,
                               config.dropout_counter_ptr(x.device) if p > 0 else None, int(site_base), 0)
        ws = _bytes(lib.immtsf_ffn_block_workspace_bytes(C.byref(cfg)), x.device)
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        ps = _struct(_lib.FFNBlockParams, params)
        check(lib.immtsf_ffn_block_forward(C.byref(cfg), C.byref(ps), ptr(x2), ptr(out), ptr(ws), ws.numel(), stream_ptr()), "ffn_block_forward")
        ctx.cfg, ctx.ws, ctx.shape = cfg, ws, x.shape
        ctx.sinks = _sinks_of(params)
        ctx.save_for_backward(x2, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x2, *params = ctx.saved_tensors
        cfg = ctx.cfg
        dout = dout.contiguous()
        dx = torch.empty_like(x2)
        grads, rets = _grad_buffers(params, ctx.sinks)
        pz = _prezeroed(params, ctx.sinks)
        cfg.grads_prezeroed = pz
        sc = _bytes(lib.immtsf_ffn_block_scratch_bytes(C.byref(cfg)), x2.device)
        ps, gs = _struct(_lib.FFNBlockParams, params), _struct(_lib.FFNBlockParams, grads)
        ws = ctx.ws
        ctx.ws = None

        def call(bits, stream):
            cfg.grads_prezeroed = pz | bits
            rc = lib.immtsf_ffn_block_backward(C.byref(cfg), C.byref(ps), ptr(x2), ptr(dout), ptr(dx), ptr(ws), ws.numel(), ptr(sc), sc.numel(),
                                               C.byref(gs), stream)
            cfg.grads_prezeroed = pz
            return rc
        # a step with a parameter-only branch (immtsf.train.FlagStep): the two weight-gradient products leave this stream's dependent chain
        # (LinearBf16Fn.backward has the why); only when every gradient goes straight to a pre-zeroed sink and the library takes the phases
        tail = config.param_tail
        flags = tail.get("wgrad_flags") if tail is not None else None
        if pz and flags and all(r is None for r in rets):
            rc = call(2, stream_ptr())
            if rc == 0:
                flag, err = flags.pop()
                check(lib.immtsf_flag_set(flag, stream_ptr()), "flag_set")

                def job(stream, keep=(x2, dout, dx, ws, sc, params, grads, ps, gs)):
                    check(lib.immtsf_flag_wait(flag, err, 50, stream), "flag_wait")
                    check(call(4, stream), "ffn_block_backward")
                tail["jobs_b"].append(job)
                return (dx.view(ctx.shape),) + (None,) * 7 + tuple(rets)
            if rc != -3:          # (IMMTSF_EUNSUPPORTED: nothing was launched -- the one-call form below)
                check(rc, "ffn_block_backward")
        check(call(0, stream_ptr()), "ffn_block_backward")
        return (dx.view(ctx.shape),) + (None,) * 7 + tuple(rets)


def ffn_block(x, conv1, conv2, norm, act, p_drop, training, site_base, precision=None):
    """conv1 / conv2: nn.Conv1d(kernel_size=1) or nn.Linear; norm: nn.LayerNorm; act: "relu" | "gelu"."""
    seed = config.next_seed() if training and p_drop > 0 else 0
    return FFNBlockFn.apply(x.float(), 1 if act == "relu" else 2, float(p_drop), norm.eps, bool(training), config.precision_code(precision),
                            seed, site_base, conv1.weight, conv1.bias, conv2.weight, conv2.bias, norm.weight, norm.bias)


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class InceptionMergeFn(torch.autograd.Function):
    """(W_0 .. W_{n-1}, b_0 .. b_{n-1}) of an Inception_Block_V1 (kernel sizes 1, 3, .., 2n-1) -> the ONE averaged kernel as a
    GEMM weight W_eff (Cout, KS*KS*Cin) and bias b_eff (immtsf_inception_merge / _unmerge)."""

    @staticmethod
    def forward(ctx, n, *wb):
        lib = _lib.load()
        ws, bs = [_c(t) for t in wb[:n]], [_c(t) for t in wb[n:]]
        _need_gpu(*ws, *bs)
        Cout, Cin = ws[0].shape[0], ws[0].shape[1]
        KS = 2 * n - 1
        Weff = torch.empty(Cout, KS * KS * Cin, dtype=torch.float32, device=ws[0].device)
        beff = torch.empty(Cout, dtype=torch.float32, device=ws[0].device)
        check(lib.immtsf_inception_merge(n, Cin, Cout, _ptr_array(ws), _ptr_array(bs), ptr(Weff), ptr(beff), stream_ptr()), "inception_merge")
        ctx.dims = (n, Cin, Cout)
        ctx.shapes = [w.shape for w in ws] + [b.shape for b in bs]
        ctx.sinks = _sinks_of(wb)
        ctx.params = wb            # parameters (not saved tensors: only their sinks / shapes are needed)
        return Weff, beff

    @staticmethod
    def backward(ctx, dW, db):
        lib = _lib.load()
        n, Cin, Cout = ctx.dims
        grads, rets = _grad_buffers(ctx.params, ctx.sinks)
        ctx.params = None
        check(lib.immtsf_inception_unmerge(n, Cin, Cout, ptr(dW.contiguous()), ptr(db.contiguous()), _ptr_array(grads[:n]), _ptr_array(grads[n:]),
                                           stream_ptr()), "inception_unmerge")
        return (None,) + tuple(rets)


class Conv2dSameCLFn(torch.autograd.Function):
    """y (B,H,W,Cout) = act(conv2d_same(x (B,H,W,Cin), W_eff) + b_eff) on channels-last images: im2col + MFMA GEMM
    (immtsf_conv2d_same_cl_forward/backward); act 0 none, 2 GELU."""

    @staticmethod
    def forward(ctx, x, Weff, beff, KS, act, precision):
        lib = _lib.load()
        x, Weff, beff = _c(x), _c(Weff), _c(beff)
        _need_gpu(x, Weff, beff)
        B, H, W, Cin = x.shape
        Cout = Weff.shape[0]
        rows, K = B * H * W, KS * KS * Cin
        col = torch.empty(rows, K, dtype=torch.float32, device=x.device)
        z = torch.empty(rows, Cout, dtype=torch.float32, device=x.device) if act == 2 else None
        y = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x.device)
        check(lib.immtsf_conv2d_same_cl_forward(precision, ptr(x), B, H, W, Cin, KS, ptr(Weff), ptr(beff), Cout, act, ptr(col), ptr(z), ptr(y),
                                                stream_ptr()), "conv2d_same_cl_forward")
        ctx.save_for_backward(col, z, Weff)
        ctx.dims = (B, H, W, Cin, KS, Cout, act, precision)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        col, z, Weff = ctx.saved_tensors
        B, H, W, Cin, KS, Cout, act, precision = ctx.dims
        rows, K = B * H * W, KS * KS * Cin
        dy = dy.contiguous()
        dx = torch.empty(B, H, W, Cin, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(Weff)
        db = torch.empty(Cout, dtype=torch.float32, device=dy.device)
        scratch = torch.empty(lib.immtsf_conv2d_same_cl_scratch_floats(B, H, W, Cin, KS, Cout), dtype=torch.float32, device=dy.device)
        check(lib.immtsf_conv2d_same_cl_backward(precision, ptr(col), ptr(z), None, ptr(dy), B, H, W, Cin, KS, ptr(Weff), Cout, act, ptr(dx),
                                                 ptr(dW), ptr(db), ptr(scratch), stream_ptr()), "conv2d_same_cl_backward")
        return dx, dW, db, None, None, None


class Conv2dPeriodFn(torch.autograd.Function):
    """Conv2dSameCLFn on a TimesNet period image whose period lives ON THE DEVICE (immtsf_conv2d_period_forward / _backward): x, y are
    position-major (Lmax * B, C) matrices (row l * B + b), `period` / `rows` one device int32 each (ops.period_rows).  Every shape on the
    host is static, so the step can be captured into a hipGraph; rows beyond `rows` are neither read nor written."""

    @staticmethod
    def forward(ctx, x, period, rows, Weff, beff, KS, act, precision, B, Lmax, w16):
        lib = _lib.load()
        x, Weff, beff = _c(x), _c(Weff), _c(beff)
        _need_gpu(x, Weff, beff)
        Cin, Cout = x.shape[1], Weff.shape[0]
        R, K = B * Lmax, KS * KS * Cin
        hf = precision == 1 and Cin % 8 == 0 and Cout % 8 == 0      # bf16 mode: the im2col image as bf16, the products on the bf16-in-HBM kernels
        col = torch.empty(R, K, dtype=torch.bfloat16 if hf else torch.float32, device=x.device)
        ready = 1 if (hf and w16 is not None) else 0                # (the kernel's bf16 image, cast once by the caller for all periods)
        if hf and w16 is None:
            w16 = torch.empty(Cout, K, dtype=torch.bfloat16, device=x.device)
        if not hf:
            w16 = None
        z = torch.empty(R, Cout, dtype=torch.float32, device=x.device) if act == 2 else None
        y = torch.empty(R, Cout, dtype=torch.float32, device=x.device)
        check(lib.immtsf_conv2d_period_forward(precision, ptr(x), B, Lmax, ptr(period), ptr(rows), Cin, KS, ptr(Weff), ptr(beff), Cout, act,
                                               ptr(col), ptr(z), ptr(y), ptr(w16), ready, stream_ptr()), "conv2d_period_forward")
        ctx.hf = hf
        ctx.save_for_backward(col, z, Weff, period, rows)
        ctx.dims = (B, Lmax, Cin, KS, Cout, act, precision)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        col, z, Weff, period, rows = ctx.saved_tensors
        B, Lmax, Cin, KS, Cout, act, precision = ctx.dims
        dy = dy.contiguous()
        # (rows beyond the image are never written: whoever consumes dx -- the previous period convolution, the slice that crops the
        # series -- never reads them either)
        dx = torch.empty(B * Lmax, Cin, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(Weff)
        db = torch.empty(Cout, dtype=torch.float32, device=dy.device)
        scratch = torch.empty(lib.immtsf_conv2d_period_scratch_floats(B, Lmax, Cin, KS, Cout), dtype=torch.float32, device=dy.device)
        w16 = torch.empty(1, dtype=torch.bfloat16, device=dy.device) if ctx.hf else None       # (only says "the bf16 path": images live in scratch)
        check(lib.immtsf_conv2d_period_backward(precision, ptr(col), ptr(z), ptr(dy), B, Lmax, ptr(period), ptr(rows), Cin, KS, ptr(Weff), Cout,
                                                act, ptr(dx), ptr(dW), ptr(db), ptr(scratch), ptr(w16), stream_ptr()), "conv2d_period_backward")
        return dx, None, None, dW, db, None, None, None, None, None, None


class Conv2dPeriodsFn(torch.autograd.Function):
    """Conv2dPeriodFn for the k period images of a TimesBlock in ONE call (immtsf_conv2d_periods_forward / _backward): x is one input shared
    by every image, (R, Cin), or one per image, (k, R, Cin); y (k, R, Cout).  In bf16 mode the images ride along grid.z of one im2col and
    one product launch per direction; the kernel's gradient is the sum over the images, accumulated in place."""

    @staticmethod
    def forward(ctx, x, period, rows, Weff, beff, KS, act, precision, B, Lmax):
        lib = _lib.load()
        x, Weff, beff = _c(x), _c(Weff), _c(beff)
        _need_gpu(x, Weff, beff)
        k = period.numel()
        shared = x.dim() == 2
        Cin, Cout = x.shape[-1], Weff.shape[0]
        R, K = B * Lmax, KS * KS * Cin
        hf = precision == 1 and Cin % 8 == 0 and Cout % 8 == 0
        col = torch.empty(k, R, K, dtype=torch.bfloat16 if hf else torch.float32, device=x.device)
        w16 = torch.empty(Cout, K, dtype=torch.bfloat16, device=x.device) if hf else None
        z = torch.empty(k, R, Cout, dtype=torch.float32, device=x.device) if act == 2 else None
        y = torch.empty(k, R, Cout, dtype=torch.float32, device=x.device)
        check(lib.immtsf_conv2d_periods_forward(precision, ptr(x), 0 if shared else R * Cin, B, Lmax, k, ptr(period), ptr(rows), Cin, KS, ptr(Weff),
                                                ptr(beff), Cout, act, ptr(col), ptr(z), ptr(y), ptr(w16), 0, stream_ptr()), "conv2d_periods_forward")
        ctx.hf, ctx.shared = hf, shared
        ctx.save_for_backward(col, z, Weff, period, rows)
        ctx.dims = (B, Lmax, Cin, KS, Cout, act, precision, k)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        col, z, Weff, period, rows = ctx.saved_tensors
        B, Lmax, Cin, KS, Cout, act, precision, k = ctx.dims
        dy = dy.contiguous()
        R = B * Lmax
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((R, Cin) if ctx.shared else (k, R, Cin), dtype=torch.float32, device=dy.device)
        acc = torch.zeros(Weff.numel() + Cout, dtype=torch.float32, device=dy.device)      # dW | db: the images add into it
        dW, db = acc[:Weff.numel()].view_as(Weff), acc[Weff.numel():]
        scratch = torch.empty(lib.immtsf_conv2d_periods_scratch_floats(B, Lmax, k, Cin, KS, Cout), dtype=torch.float32, device=dy.device)
        w16 = torch.empty(1, dtype=torch.bfloat16, device=dy.device) if ctx.hf else None
        check(lib.immtsf_conv2d_periods_backward(precision, ptr(col), ptr(z), ptr(dy), B, Lmax, k, ptr(period), ptr(rows), Cin, KS, ptr(Weff), Cout,
                                                 act, ptr(dx), 1 if ctx.shared else 0, ptr(dW), ptr(db), ptr(scratch), ptr(w16), stream_ptr()),
              "conv2d_periods_backward")
        return dx, None, None, dW, db, None, None, None, None, None


class InceptionPeriodsFn(torch.autograd.Function):
    """An Inception_Block_V1 (layers/Conv_Blocks.py:8-31: the mean of n same-padded convolutions, kernel sizes 1, 3, .., 2n-1) over the k
    period images of a TimesBlock in the IMPLICIT form (csrc/conv.hip conv_period_mfma_kernel: no im2col image, a workgroup holds a
    (window, period) image in LDS): merge of the n kernels, the convolution (+ GELU), and backward the data gradient through the same
    kernel, the merged kernel's gradient from an im2col image of the INPUT formed only for that product, and the un-merge into the n
    kernels' gradients.  With a parameter branch (immtsf.train.FlagStep) and gradient sinks the kernel-gradient half runs there.
    args: x (R, Cin) shared or (k, R, Cin), period, rows, act, precision, B, Lmax, n, W_0..W_{n-1}, b_0..b_{n-1}."""

    @staticmethod
    def forward(ctx, x, period, rows, act, precision, B, Lmax, n, *wb):
        lib = _lib.load()
        x = _c(x)
        ws, bs = [_c(t) for t in wb[:n]], [_c(t) for t in wb[n:]]
        _need_gpu(x, *ws, *bs)
        k = period.numel()
        shared = x.dim() == 2
        Cout, Cin = ws[0].shape[0], ws[0].shape[1]
        KS = 2 * n - 1
        R, K = B * Lmax, KS * KS * Cin
        dev = x.device
        Weff = torch.empty(Cout, K, dtype=torch.float32, device=dev)
        beff = torch.empty(Cout, dtype=torch.float32, device=dev)
        check(lib.immtsf_inception_merge(n, Cin, Cout, _ptr_array(ws), _ptr_array(bs), ptr(Weff), ptr(beff), stream_ptr()), "inception_merge")
        w16 = torch.empty(Cout, K, dtype=torch.bfloat16, device=dev)
        z = torch.empty(k, R, Cout, dtype=torch.float32, device=dev) if act == 2 else None
        y = torch.empty(k, R, Cout, dtype=torch.float32, device=dev)
        check(lib.immtsf_conv2d_periods_forward(precision, ptr(x), 0 if shared else R * Cin, B, Lmax, k, ptr(period), ptr(rows), Cin, KS, ptr(Weff),
                                                ptr(beff), Cout, act, None, ptr(z), ptr(y), ptr(w16), 0, stream_ptr()), "conv2d_periods_forward")
        ctx.shared = shared
        ctx.save_for_backward(x, z, Weff, period, rows)
        ctx.dims = (B, Lmax, Cin, KS, Cout, act, precision, k, n)
        ctx.sinks = _sinks_of(wb)
        ctx.params = wb
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, z, Weff, period, rows = ctx.saved_tensors
        B, Lmax, Cin, KS, Cout, act, precision, k, n = ctx.dims
        dy = dy.contiguous()
        R = B * Lmax
        dev = dy.device
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((R, Cin) if ctx.shared else (k, R, Cin), dtype=torch.float32, device=dev)
        acc = torch.zeros(Weff.numel() + Cout, dtype=torch.float32, device=dev)      # d W_eff | d b_eff: the images add into it
        dW, db = acc[:Weff.numel()].view_as(Weff), acc[Weff.numel():]
        scratch = torch.empty(lib.immtsf_conv2d_periods_backward_x_scratch_floats(B, Lmax, k, Cin, KS, Cout), dtype=torch.float32, device=dev)
        grads, rets = _grad_buffers(ctx.params, ctx.sinks)
        ctx.params = None
        xs = 0 if ctx.shared else R * Cin

        def call(phase, stream):
            check(lib.immtsf_conv2d_periods_backward_x(precision, ptr(x), xs, ptr(z), ptr(dy), B, Lmax, k, ptr(period), ptr(rows), Cin, KS, ptr(Weff),
                                                       Cout, act, ptr(dx), 1 if ctx.shared else 0, ptr(dW), ptr(db), ptr(scratch), phase, stream),
                  "conv2d_periods_backward_x")

        def wgrad(stream, keep=(x, z, dy, Weff, acc, scratch, grads)):
            call(2, stream)
            check(lib.immtsf_inception_unmerge(n, Cin, Cout, ptr(dW), ptr(db), _ptr_array(grads[:n]), _ptr_array(grads[n:]), stream),
                  "inception_unmerge")
        call(1, stream_ptr())
        # a step with a parameter-only branch: the kernels' gradients -- an im2col image, the summed product, the un-merge; nothing in the
        # backward waits for them -- leave this stream's dependent chain (needs every kernel on a gradient sink: nothing is returned)
        tail = config.param_tail
        flags = tail.get("wgrad_flags") if tail is not None else None
        if flags and all(r is None for r in rets):
            flag, err = flags.pop()
            check(lib.immtsf_flag_set(flag, stream_ptr()), "flag_set")

            def job(stream):
                check(lib.immtsf_flag_wait(flag, err, 50, stream), "flag_wait")
                wgrad(stream)
            tail["jobs_b"].append(job)
        else:
            wgrad(stream_ptr())
        return (dx,) + (None,) * 7 + tuple(rets)


def inception_periods_ok(block, Lmax, precision=None):
    """True where InceptionPeriodsFn applies to this Inception_Block_V1 (immtsf_conv2d_periods_implicit_ok: bf16 mode, channel counts
    multiples of 8 up to 64, Lmax <= 128)"""
    n = len(block.kernels)
    if n > INCEPTION_MAX or not block.kernels[0].weight.is_cuda:
        return False
    Cout, Cin = block.kernels[0].weight.shape[:2]
    return bool(_lib.load().immtsf_conv2d_periods_implicit_ok(config.precision_code(precision), int(Lmax), int(Cin), 2 * n - 1, int(Cout)))


def inception_periods(x, period, rows, block, B, Lmax, act=None, precision=None):
    ks = block.kernels
    return InceptionPeriodsFn.apply(x.float(), period, rows, 2 if act == "gelu" else 0, config.precision_code(precision), int(B), int(Lmax),
                                    len(ks), *[c.weight for c in ks], *[c.bias for c in ks])


class PeriodAggregateFn(torch.autograd.Function):
    """TimesBlock's aggregation of the k period images and its residual as one launch (immtsf_period_aggregate_forward / _backward):
    (Y (k, Lmax*B, N) position-major, w (B, k) softmax weights, x (B, total, N)) -> x + sum_j w[:, j] Y_j cropped to `total` steps."""

    @staticmethod
    def forward(ctx, Y, w, x, B, total, Lmax):
        lib = _lib.load()
        Y, w, x = _c(Y), _c(w), _c(x)
        _need_gpu(Y, w, x)
        k, N = Y.shape[0], Y.shape[2]
        out = torch.empty_like(x)
        check(lib.immtsf_period_aggregate_forward(ptr(Y), ptr(w), ptr(x), B, total, Lmax, N, k, ptr(out), stream_ptr()), "period_aggregate_forward")
        ctx.save_for_backward(Y, w)
        ctx.dims = (B, total, Lmax, N, k)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        Y, w = ctx.saved_tensors
        B, total, Lmax, N, k = ctx.dims
        dout = dout.contiguous()
        dY, dw = torch.empty_like(Y), torch.empty_like(w)
        check(lib.immtsf_period_aggregate_backward(ptr(Y), ptr(w), ptr(dout), B, total, Lmax, N, k, ptr(dY), ptr(dw), stream_ptr()),
              "period_aggregate_backward")
        return dY, dw, dout, None, None, None


def conv2d_periods(x, period, rows, Weff, beff, KS, B, Lmax, act=None, precision=None):
    return Conv2dPeriodsFn.apply(x.float(), period, rows, Weff, beff, int(KS), 2 if act == "gelu" else 0, config.precision_code(precision),
                                 int(B), int(Lmax))


def period_aggregate(Y, w, x, B, total, Lmax):
    return PeriodAggregateFn.apply(Y, w.float(), x.float(), int(B), int(total), int(Lmax))


def period_rows(top, total, B):
    """top (k int64 device values: TimesNet's selected frequency indices) -> (period int32 (k), rows int32 (k)) on the device: period =
    total // top, rows = B * (total rounded up to a multiple of the period) -- models/TimesNet.py:17-18, 50-56 without the host read"""
    k = top.numel()
    period = torch.empty(k, dtype=torch.int32, device=top.device)
    rows = torch.empty(k, dtype=torch.int32, device=top.device)
    check(_lib.load().immtsf_period_rows(ptr(_c(top)), k, int(total), int(B), ptr(period), ptr(rows), stream_ptr()), "period_rows")
    return period, rows


def conv2d_period(x, period, rows, Weff, beff, KS, B, Lmax, act=None, precision=None, w16=None):
    return Conv2dPeriodFn.apply(x.float(), period, rows, Weff, beff, int(KS), 2 if act == "gelu" else 0, config.precision_code(precision),
                                int(B), int(Lmax), w16)


INCEPTION_MAX = 8


def inception_merge(block):
    """block: layers.Conv_Blocks.Inception_Block_V1 -> (W_eff, b_eff, KS)"""
    ks = block.kernels
    n = len(ks)
    Weff, beff = InceptionMergeFn.apply(n, *[k.weight for k in ks], *[k.bias for k in ks])
    return Weff, beff, 2 * n - 1


def conv2d_same_cl(x, Weff, beff, KS, act=None, precision=None):
    return Conv2dSameCLFn.apply(x.float(), Weff, beff, int(KS), 2 if act == "gelu" else 0, config.precision_code(precision))


class TPatchDecoderFn(torch.autograd.Function):
    """tPatchGNN's forecast decoder on (h (B,N,D), te (B,Lp,E)) -> (B,Lp,N): one kernel per direction -- exact fp32, or in bf16
    mode the second layer and its gradients on MFMA tiles (csrc/decoder.hip).  Backward recomputes the forward; parameter
    gradients accumulate by atomics into one zeroed flat buffer."""

    @staticmethod
    def forward(ctx, h, te, precision, *params):
        lib = _lib.load()
        h, te = _c(h), _c(te)
        params = tuple(_c(p) for p in params)
        _need_gpu(h, te, *params)
        B, N, D = h.shape
        Lp, E = te.shape[1], te.shape[2]
        H = params[2].shape[0]
        out = torch.empty(B, Lp, N, dtype=torch.float32, device=h.device)
        ps = _struct(DecoderParams, params)
        check(lib.immtsf_tpatchgnn_decoder_forward_p(B, N, Lp, D, E, H, precision, ptr(h), ptr(te), C.byref(ps), ptr(out), stream_ptr()),
              "tpatchgnn_decoder_forward")
        ctx.dims = (B, N, Lp, D, E, H, precision)
        ctx.sinks = _sinks_of(params)
        ctx.save_for_backward(h, te, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        h, te, *params = ctx.saved_tensors
        B, N, Lp, D, E, H, precision = ctx.dims
        dout = dout.contiguous()
        dh, dte = torch.empty_like(h), torch.empty_like(te)
        grads, rets = _zeroed_grad_buffers(params, ctx.sinks)
        ps, gs = _struct(DecoderParams, params), _struct(DecoderParams, grads)
        check(lib.immtsf_tpatchgnn_decoder_backward_p(B, N, Lp, D, E, H, precision, ptr(h), ptr(te), C.byref(ps), ptr(dout), ptr(dh),
                                                      ptr(dte), C.byref(gs), stream_ptr()), "tpatchgnn_decoder_backward")
        return (dh, dte, None) + tuple(rets)


class TPatchDecoderTEFn(torch.autograd.Function):
    """The decoder with LearnableTE of the prediction times inside: (h (B,N,D), t (B,Lp)) -> (B,Lp,N).  te = [w0 t + b0 ; sin(w t + b)]
    is built while a window is staged and its gradient is reduced to the four parameter gradients in the backward kernel: no
    (B,Lp,E) tensor, no separate Time2Vec launches (models/tPatchGNN.py:176-180, 283-291).  t is data (no gradient)."""

    @staticmethod
    def forward(ctx, h, t, precision, w0, b0, w, b, *params):
        lib = _lib.load()
        h, t = _c(h), _c(t)
        tparams = tuple(_c(q) for q in (w0, b0, w, b))
        params = tuple(_c(q) for q in params)
        _need_gpu(h, t, *tparams, *params)
        B, N, D = h.shape
        Lp, E = t.shape[1], 1 + tparams[2].numel()
        H = params[2].shape[0]
        out = torch.empty(B, Lp, N, dtype=torch.float32, device=h.device)
        ps, ts = _struct(DecoderParams, params), _struct(_lib.Time2VecParams, tparams)
        check(lib.immtsf_tpatchgnn_decoder_forward_te(B, N, Lp, D, E, H, precision, ptr(h), ptr(t), C.byref(ts), C.byref(ps), ptr(out),
                                                      stream_ptr()), "tpatchgnn_decoder_forward_te")
        ctx.dims = (B, N, Lp, D, E, H, precision)
        ctx.sinks, ctx.tsinks = _sinks_of(params), _sinks_of(tparams)
        _claim_sinks(tparams, ctx.tsinks, "tpatch_decoder_te")
        ctx.save_for_backward(h, t, *tparams, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        h, t, w0, b0, w, b, *params = ctx.saved_tensors
        tparams = (w0, b0, w, b)
        B, N, Lp, D, E, H, precision = ctx.dims
        dout = dout.contiguous()
        dh = torch.empty_like(h)
        grads, rets = _zeroed_grad_buffers(params, ctx.sinks)
        tgrads, trets = _zeroed_grad_buffers(tparams, ctx.tsinks)       # (shared sinks -- the patch encoder adds into them too -- are zero-filled by their owner)
        ps, gs = _struct(DecoderParams, params), _struct(DecoderParams, grads)
        ts, tg = _struct(_lib.Time2VecParams, tparams), _struct(_lib.Time2VecParams, tgrads)
        check(lib.immtsf_tpatchgnn_decoder_backward_te(B, N, Lp, D, E, H, precision, ptr(h), ptr(t), C.byref(ts), C.byref(ps), ptr(dout),
                                                       ptr(dh), C.byref(gs), C.byref(tg), stream_ptr()), "tpatchgnn_decoder_backward_te")
        return (dh, None, None) + tuple(trets) + tuple(rets)


def tpatch_decoder_supported(seq, N, Lp, D, E):
    """seq: the reference's decoder nn.Sequential (Linear, ReLU, Linear, ReLU, Linear(H, 1))"""
    mods = list(seq)
    if len(mods) != 5 or not all(isinstance(m, torch.nn.Linear) for m in mods[0::2]) \
            or not all(isinstance(m, torch.nn.ReLU) for m in mods[1::2]):
        return False
    l1, l2, l3 = mods[0], mods[2], mods[4]
    H = l1.out_features
    if l1.in_features != D + E or l2.in_features != H or l2.out_features != H or l3.in_features != H or l3.out_features != 1:
        return False
    if any(m.bias is None for m in (l1, l2, l3)):
        return False
    return _lib.load().immtsf_tpatchgnn_decoder_lds_bytes(N, Lp, D, E, H) > 0


def tpatch_decoder(seq, h, te, precision=None):
    """decoder(cat[h broadcast over Lp ; te broadcast over N]).squeeze(-1).permute(0, 2, 1) of models/tPatchGNN.py:283-291
    without materialising the (B, N, Lp, D+E) tensor: h (B,N,D), te (B,Lp,E) -> (B,Lp,N)."""
    l1, l2, l3 = seq[0], seq[2], seq[4]
    return TPatchDecoderFn.apply(h.float(), te.float(), config.precision_code(precision), l1.weight, l1.bias, l2.weight, l2.bias,
                                 l3.weight, l3.bias)


def tpatch_decoder_te(seq, h, t, w0, b0, w, b, precision=None):
    """tpatch_decoder with the time embedding of the prediction times t (B, Lp) computed inside (w0 / b0: te_scale, w / b: te_periodic)"""
    if t.requires_grad:
        raise RuntimeError("immtsf.tpatch_decoder_te: timestamps are data; no gradient is produced for them")
    l1, l2, l3 = seq[0], seq[2], seq[4]
    return TPatchDecoderTEFn.apply(h.float(), t.float(), config.precision_code(precision), w0, b0, w, b, l1.weight, l1.bias, l2.weight,
                                   l2.bias, l3.weight, l3.bias)


# ------------------------------------------------------------------------------------------------ layer primitives
_ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2}


def _gemm(layout, prec, A, lda, B, ldb, Cm, ldc, bias, M, N, K, alpha=1.0, accumulate=0, act=0):
    lib = _lib.load()
    check(lib.immtsf_gemm(layout, prec, ptr(A), lda, ptr(B), ldb, ptr(Cm), ldc, ptr(bias), M, N, K, alpha, accumulate, act,
                          stream_ptr()), "gemm")


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) on the MFMA GEMM (NT forward with the bias / ReLU epilogue; backward = one
    immtsf_linear_backward call: NN data gradient, TN weight gradient with the bias gradient folded in)."""

    @staticmethod
    def forward(ctx, x, W, b, precision, relu):
        x2 = _c(x).reshape(-1, x.shape[-1])
        W = _c(W)
        _need_gpu(x2, W, b)
        M, K = x2.shape
        N = W.shape[0]
        y = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)   # not a view: callers may relu_ it
        _gemm(0, precision, x2, K, W, K, y, N, b, M, N, K, act=1 if relu else 0)
        ctx.save_for_backward(x2, W, y if relu else None)
        ctx.has_bias, ctx.precision, ctx.shape = b is not None, precision, x.shape
        ctx.sinks = _sinks_of((W, b))
        ctx.pre = all(s is not None and getattr(p, "_immtsf_grad_prezeroed", False) for p, s in zip((W, b), ctx.sinks) if p is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, W, y = ctx.saved_tensors
        M, K = x2.shape
        N = W.shape[0]
        dy2 = dy.contiguous().reshape(M, N)
        if y is not None:
            dy2 = torch.ops.aten.threshold_backward(dy2, y.reshape(M, N), 0.0)
        need_w = ctx.needs_input_grad[1]
        dx = torch.empty(M, K, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[0] else None
        dW = db = rW = rb = None
        if need_w and ctx.pre:  # gradient sinks that their owner zero-fills every step: written in place, nothing returned
            dW, db = ctx.sinks
        elif need_w:            # ONE zero fill for dW|db: the split-K weight gradient then needs no memsets of its own
            flat = torch.zeros(N * K + (N if ctx.has_bias else 0), dtype=torch.float32, device=dy.device)
            dW = rW = flat[:N * K].view(N, K)
            db = rb = flat[N * K:] if ctx.has_bias else None
        check(lib.immtsf_linear_backward(ctx.precision, ptr(x2), ptr(W), ptr(dy2), M, N, K, ptr(dx), None, ptr(dW), ptr(db),
                                         1, stream_ptr()), "linear_backward")
        return (dx.view(ctx.shape) if dx is not None else None), rW, rb, None, None


class LinearBf16Fn(torch.autograd.Function):
    """nl (1..3) linear layers y_i = act(x W_i^T + b_i) that share ONE input, in the bf16 dataflow (immtsf_linear_bf16_forward /
    _backward): the input is cast once and kept as bf16 for the weight gradients, the weights are read from FlatTrainer's bf16 twins
    (else cast into a scratch image), the forward is ONE launch, the backward one cast per upstream gradient, nl accumulating
    data-gradient launches and ONE grouped weight-gradient launch.  args: x, relu, nl, W_1..W_nl, b_1..b_nl (None allowed)."""

    @staticmethod
    def forward(ctx, x, relu, nl, *wb):
        lib = _lib.load()
        Ws, bs = [_c(w) for w in wb[:nl]], list(wb[nl:])
        x2 = _c(x).reshape(-1, x.shape[-1])
        _need_gpu(x2, *Ws)
        M, K = x2.shape
        N = Ws[0].shape[0]
        dev = x.device
        x16 = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
        w16 = [torch.empty(N, K, dtype=torch.bfloat16, device=dev) for _ in range(nl)]       # (unused when a twin is registered)
        ys = [torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=dev) for _ in range(nl)]
        arr = lambda ts: (C.c_void_p * nl)(*[None if t is None else t.data_ptr() for t in ts])      # noqa: E731
        check(lib.immtsf_linear_bf16_forward(nl, ptr(x2), ptr(x16), arr(Ws), arr(w16), arr(bs), arr(ys), M, N, K, 1 if relu else 0,
                                             stream_ptr()), "linear_bf16_forward")
        ctx.save_for_backward(x16, *Ws, *([ys[0]] if relu else []))
        ctx.nl, ctx.relu, ctx.shape, ctx.w16 = nl, relu, x.shape, w16
        ctx.has_bias = [b is not None for b in bs]
        ctx.wsinks, ctx.bsinks = _sinks_of(Ws), _sinks_of(bs)
        ctx.pre = all(s is not None and getattr(p, "_immtsf_grad_prezeroed", False)
                      for p, s in zip(list(Ws) + bs, ctx.wsinks + ctx.bsinks) if p is not None)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        lib = _lib.load()
        nl = ctx.nl
        saved = ctx.saved_tensors
        x16, Ws = saved[0], saved[1:1 + nl]
        M, K = x16.shape
        N = Ws[0].shape[0]
        dev = x16.device
        dys = [torch.zeros(M, N, dtype=torch.float32, device=dev) if d is None else d.contiguous().reshape(M, N) for d in dys]
        if ctx.relu:
            dys[0] = torch.ops.aten.threshold_backward(dys[0], saved[1 + nl].reshape(M, N), 0.0)
        need_w = any(ctx.needs_input_grad[3 + i] for i in range(nl))
        dx = torch.empty(M, K, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        rW, rb = [None] * nl, [None] * nl
        dW, db = [None] * nl, [None] * nl
        if need_w and ctx.pre:          # gradient sinks that their owner zero-fills every step: written in place, nothing returned
            dW, db = list(ctx.wsinks), list(ctx.bsinks)
        elif need_w:                    # ONE zero fill for every dW | db
            per = N * K + N
            flat = torch.zeros(nl * per, dtype=torch.float32, device=dev)
            for i in range(nl):
                dW[i] = rW[i] = flat[i * per:i * per + N * K].view(N, K)
                if ctx.has_bias[i]:
                    db[i] = rb[i] = flat[i * per + N * K:(i + 1) * per]
        dy16 = torch.empty(nl * M * N, dtype=torch.bfloat16, device=dev)
        arr = lambda ts: (C.c_void_p * nl)(*[None if t is None else t.data_ptr() for t in ts])      # noqa: E731
        # a step with a parameter-only branch (immtsf.train.FlagStep): the weight gradients -- nothing in the backward waits for them --
        # leave this stream's dependent chain: the data gradient (and the casts) here, then a flag; the grouped weight-gradient launch
        # runs on that branch behind the flag.  Only when every gradient goes straight to a pre-zeroed sink (nothing is returned).
        tail = config.param_tail
        flags = tail.get("wgrad_flags") if tail is not None else None
        if need_w and ctx.pre and flags and dx is not None:
            check(lib.immtsf_linear_bf16_backward(nl, ptr(x16), arr(Ws), arr(ctx.w16), arr(dys), ptr(dy16), ptr(dx), None, None, M, N, K, 1,
                                                  stream_ptr()), "linear_bf16_backward")
            flag, err = flags.pop()
            check(lib.immtsf_flag_set(flag, stream_ptr()), "flag_set")
            keep = (x16, Ws, ctx.w16, dys, dy16, dW, db)          # the job runs later, on another stream: everything it touches stays alive

            def job(stream, keep=keep, flag=flag, err=err, nl=nl, M=M, N=N, K=K):
                x16_, Ws_, w16_, dys_, dy16_, dW_, db_ = keep
                arr_ = lambda ts: (C.c_void_p * nl)(*[None if t is None else t.data_ptr() for t in ts])      # noqa: E731
                check(lib.immtsf_flag_wait(flag, err, 50, stream), "flag_wait")
                check(lib.immtsf_linear_bf16_backward(nl, ptr(x16_), arr_(Ws_), arr_(w16_), arr_(dys_), ptr(dy16_), None, arr_(dW_), arr_(db_), M, N, K,
                                                      3, stream), "linear_bf16_backward")
            tail["jobs_b"].append(job)
            return (dx.view(ctx.shape), None, None) + tuple(rW) + tuple(rb)
        check(lib.immtsf_linear_bf16_backward(nl, ptr(x16), arr(Ws), arr(ctx.w16), arr(dys), ptr(dy16), ptr(dx), arr(dW) if need_w else None,
                                              arr(db) if need_w else None, M, N, K, 1, stream_ptr()), "linear_bf16_backward")
        return ((dx.view(ctx.shape) if dx is not None else None), None, None) + tuple(rW) + tuple(rb)


def _bf16_linear_ok(x, W, precision):
    # (small layers -- tPatchGNN's 64 -> 32 temporal aggregation -- keep their exact-fp32 one-launch kernels)
    return (config.precision_code(precision) == 1 and x.is_cuda and W.shape[1] % 8 == 0 and W.shape[0] % 8 == 0 and W.shape[1] >= 128 and
            W.shape[0] >= 64 and x.numel() // x.shape[-1] >= 256)


def linear(x, W, b=None, precision=None, relu=False):
    if _bf16_linear_ok(x, W, precision):      # bf16 mode, enough rows: the bf16-in-HBM kernels (operands cast once, weights from their twins)
        return LinearBf16Fn.apply(x.float(), relu, 1, W, b)[0]
    return LinearFn.apply(x.float(), W, b, config.precision_code(precision), relu)


def linear_multi(x, weights, biases, precision=None):
    """[x W_i^T + b_i] for layers sharing one input (a self-attention's q | k | v projections) -- one launch forward, one grouped
    weight-gradient launch backward in bf16 mode; one linear() each otherwise"""
    nl = len(weights)
    same = all(w.shape == weights[0].shape for w in weights)
    if 1 <= nl <= 3 and same and _bf16_linear_ok(x, weights[0], precision):
        return list(LinearBf16Fn.apply(x.float(), False, nl, *weights, *biases))
    return [linear(x, w, b, precision) for w, b in zip(weights, biases)]


class MLPFn(torch.autograd.Function):
    """Linear (ReLU Linear)* chain, e.g. tPatchGNN's decoder (models/tPatchGNN.py:158-163): one GEMM per layer forward
    (bias + ReLU in the epilogue), two per layer backward -- the ReLU mask is applied in the epilogue of the NEXT
    layer's data-gradient GEMM and the bias gradient rides the weight-gradient GEMM as a ones column."""

    @staticmethod
    def forward(ctx, x, precision, has_bias, *params):
        nl = len(has_bias)
        Ws = [_c(w) for w in params[:nl]]
        bs = list(params[nl:])
        acts = [_c(x).reshape(-1, x.shape[-1])]
        _need_gpu(acts[0], *Ws)
        for i, W in enumerate(Ws):
            M, K = acts[-1].shape
            N = W.shape[0]
            shape = (M, N) if i + 1 < nl else (*x.shape[:-1], N)
            y = torch.empty(shape, dtype=torch.float32, device=x.device)
            _gemm(0, precision, acts[-1], K, W, K, y, N, bs[i], M, N, K, act=1 if i + 1 < nl else 0)
            acts.append(y)
        ctx.save_for_backward(*acts[:-1], *Ws)
        ctx.nl, ctx.has_bias, ctx.precision, ctx.shape = nl, has_bias, precision, x.shape
        ctx.wsinks, ctx.bsinks = _sinks_of(params[:nl]), _sinks_of(bs)
        ctx.pre = all(s is not None and getattr(p, "_immtsf_grad_prezeroed", False)
                      for p, s in zip(list(params[:nl]) + bs, ctx.wsinks + ctx.bsinks) if p is not None)
        return acts[-1]

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        nl = ctx.nl
        acts, Ws = ctx.saved_tensors[:nl], ctx.saved_tensors[nl:]
        dWs, dbs = [None] * nl, [None] * nl
        g = dy.contiguous().reshape(-1, Ws[-1].shape[0])
        dev = dy.device
        if ctx.pre:             # FlatTrainer sinks (zero-filled by their owner every step): written in place
            dWs, dbs = list(ctx.wsinks), list(ctx.bsinks)
            rWs, rbs = [None] * nl, [None] * nl
        else:
            # ONE zero-filled buffer for every layer's dW|db (32-byte aligned slices): no per-GEMM memsets
            sizes = [((W.numel() + 7) // 8 * 8, (W.shape[0] + 7) // 8 * 8) if ctx.needs_input_grad[3 + i] else (0, 0)
                     for i, W in enumerate(Ws)]
            flat = torch.zeros(max(1, sum(a + b for a, b in sizes)), dtype=torch.float32, device=dev)
            off = 0
            for i, W in enumerate(Ws):
                if ctx.needs_input_grad[3 + i]:
                    dWs[i] = flat[off:off + W.numel()].view(W.shape)
                    off += sizes[i][0]
                    if ctx.has_bias[i]:
                        dbs[i] = flat[off:off + W.shape[0]]
                    off += sizes[i][1]
            rWs, rbs = dWs, dbs
        for i in reversed(range(nl)):
            x, W = acts[i], Ws[i]
            M, K = x.shape
            N = W.shape[0]
            need_x = i > 0 or ctx.needs_input_grad[0]
            dx = torch.empty(M, K, dtype=torch.float32, device=dev) if need_x else None
            check(lib.immtsf_linear_backward(ctx.precision, ptr(x), ptr(W), ptr(g), M, N, K, ptr(dx),
                                             ptr(x) if i > 0 else None, ptr(dWs[i]), ptr(dbs[i]), 1, stream_ptr()),
                  "linear_backward")
            g = dx
        return (g.view(ctx.shape) if g is not None else None), None, None, *rWs, *rbs


def mlp(x, weights, biases, precision=None):
    """weights[i] (N_i, K_i), biases[i] (N_i) or None; ReLU between layers, none after the last."""
    return MLPFn.apply(x.float(), config.precision_code(precision), tuple(b is not None for b in biases), *weights, *biases)


class Time2VecFn(torch.autograd.Function):
    """[w0 t + b0, sin(w t + b)] rows (Time2Vec / tPatchGNN.LearnableTE); t is data (no gradient)."""

    @staticmethod
    def forward(ctx, t, w0, b0, w, b):
        lib = _lib.load()
        t1 = _c(t).reshape(-1)
        _need_gpu(t1, w0, b0)
        d = 1 + (w.numel() if w is not None else 0)
        out = torch.empty(*t.shape, d, dtype=torch.float32, device=t.device)
        check(lib.immtsf_time2vec_forward(ptr(t1), t1.numel(), d, ptr(w0), ptr(b0), ptr(w), ptr(b), ptr(out), stream_ptr()),
              "time2vec_forward")
        _claim_sinks((w0, b0, w, b), _sinks_of((w0, b0, w, b)), "time2vec")
        ctx.save_for_backward(t1, w0, b0, w, b)
        ctx.d = d
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        t1, w0, b0, w, b = ctx.saved_tensors
        d = ctx.d
        g = dout.contiguous().reshape(-1, d)
        params = (w0, b0, w, b)
        sinks = _sinks_of(params)
        acc = w is not None and _shared(params, sinks)      # parameters shared with another HIP op (tPatchGNN: the TTCN encoder)
        if acc:
            (dw0, db0, dw, db), rets = sinks, (None,) * 4
        else:
            dw0, db0 = torch.empty_like(w0), torch.empty_like(b0)
            dw = torch.empty_like(w) if w is not None else None
            db = torch.empty_like(b) if b is not None else None
            rets = (dw0, db0, dw, db)
        scratch = torch.empty(2 * 64 * d, dtype=torch.float32, device=g.device)
        check(lib.immtsf_time2vec_backward(ptr(t1), t1.numel(), d, ptr(w), ptr(b), ptr(g), ptr(dw0), ptr(db0), ptr(dw), ptr(db),
                                           ptr(scratch), 1 if acc else 0, stream_ptr()), "time2vec_backward")
        return (None,) + tuple(rets)


def time2vec(t, w0, b0, w, b):
    """t (...,) -> (..., 1 + len(w)); w0/b0 = Linear(1,1) weight/bias, w/b = Linear(1,d-1) weight (d-1,1) / bias."""
    if t.requires_grad:
        raise RuntimeError("immtsf.time2vec: timestamps are data; no gradient is produced for them")
    return Time2VecFn.apply(t.float(), _c(w0), _c(b0), None if w is None else _c(w), None if b is None else _c(b))


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        lib = _lib.load()
        x2 = _c(x).reshape(-1, x.shape[-1])
        _need_gpu(x2, gamma, beta)
        rows, d = x2.shape
        xhat = torch.empty_like(x2)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        z = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        check(lib.immtsf_layernorm_forward(ptr(x2), rows, d, ptr(gamma), ptr(beta), float(eps), ptr(xhat), ptr(rstd), ptr(z),
                                           0.0, 0, 0, stream_ptr()), "layernorm_forward")
        ctx.save_for_backward(xhat, rstd, gamma)
        ctx.shape = x.shape
        ctx.sinks = _sinks_of((gamma, beta))
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        xhat, rstd, gamma = ctx.saved_tensors
        rows, d = xhat.shape
        g = dz.contiguous().reshape(rows, d).clone()
        dx = torch.empty_like(xhat)
        (dgamma, dbeta), rets = _grad_buffers((gamma, gamma), ctx.sinks)        # overwritten by the kernel
        scratch = torch.empty(64 * d, dtype=torch.float32, device=dz.device)
        check(lib.immtsf_layernorm_backward(ptr(g), rows, d, ptr(gamma), ptr(xhat), ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta),
                                            ptr(scratch), 0.0, 0, 0, stream_ptr()), "layernorm_backward")
        return dx.view(ctx.shape), rets[0], rets[1], None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x.float(), gamma, beta, eps)


class FullAttentionFn(torch.autograd.Function):
    """softmax(scale * Q K^T) V per (batch, head) for (B,L,H,E)/(B,S,H,E)/(B,S,H,D) tensors
    (layers/SelfAttention_Family.py:50-77): batched MFMA GEMMs + fused softmax/dropout rows."""

    @staticmethod
    def forward(ctx, q, k, v, scale, p_drop, training, seed, site, causal, precision):
        lib = _lib.load()
        q, k, v = _c(q), _c(k), _c(v)
        _need_gpu(q, k, v)
        B, L, H, E = q.shape
        S, D = k.shape[1], v.shape[3]
        dev = q.device
        P = torch.empty(B, H, L, S, dtype=torch.float32, device=dev)
        p = float(p_drop) if training else 0.0
        if config.attn_mid and lib.immtsf_attn_mid_supported(L, S, E, D):
            # few positions, wide heads (PatchTST: 10 x 10 scores with E = 256): one kernel per direction (csrc/attn_mid.hip)
            cnt = config.dropout_counter_ptr(dev) if p > 0 else None
            out = torch.empty(B, L, H, D, dtype=torch.float32, device=dev)
            check(lib.immtsf_attn_mid_forward(ptr(q), ptr(k), ptr(v), B, L, S, H, E, D, float(scale), 1 if causal else 0, p, seed, site, cnt,
                                              ptr(P), ptr(out), stream_ptr()), "attn_mid_forward")
            ctx.save_for_backward(q, k, v, P)
            ctx.cfg = (scale, p, seed, site, precision, cnt)
            ctx.mid = True
            return out
        ctx.mid = False
        check(lib.immtsf_gemm_batched(0, precision, ptr(q), H * E, L * H * E, E, ptr(k), H * E, S * H * E, E, ptr(P), S,
                                      H * L * S, L * S, B, H, L, S, E, float(scale), stream_ptr()), "qk^T")
        A = torch.empty_like(P) if p > 0 else P
        cnt = config.dropout_counter_ptr(dev) if p > 0 else None
        check(lib.immtsf_softmax_rows_forward(ptr(P), ptr(A), B, H, L, S, None, p, seed, site, 1 if causal else 0, cnt,
                                              stream_ptr()), "softmax")
        out = torch.empty(B, L, H, D, dtype=torch.float32, device=dev)
        check(lib.immtsf_gemm_batched(1, precision, ptr(A), S, H * L * S, L * S, ptr(v), H * D, S * H * D, D, ptr(out), H * D,
                                      L * H * D, D, B, H, L, D, S, 1.0, stream_ptr()), "a.v")
        ctx.save_for_backward(q, k, v, P, A)
        ctx.cfg = (scale, p, seed, site, precision, cnt)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        if ctx.mid:
            q, k, v, P = ctx.saved_tensors
            A = None
        else:
            q, k, v, P, A = ctx.saved_tensors
        scale, p, seed, site, precision, cnt = ctx.cfg
        B, L, H, E = q.shape
        S, D = k.shape[1], v.shape[3]
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        st = stream_ptr()
        if ctx.mid:
            check(lib.immtsf_attn_mid_backward(ptr(q), ptr(k), ptr(v), ptr(P), ptr(dout), B, L, S, H, E, D, float(scale), p, seed, site, cnt,
                                               ptr(dq), ptr(dk), ptr(dv), st), "attn_mid_backward")
            return dq, dk, dv, None, None, None, None, None, None, None
        dA = torch.empty_like(P)
        # dA = dO V^T ; dV = A^T dO
        check(lib.immtsf_gemm_batched(0, precision, ptr(dout), H * D, L * H * D, D, ptr(v), H * D, S * H * D, D, ptr(dA), S,
                                      H * L * S, L * S, B, H, L, S, D, 1.0, st), "dA")
        check(lib.immtsf_gemm_batched(2, precision, ptr(A), S, H * L * S, L * S, ptr(dout), H * D, L * H * D, D, ptr(dv), H * D,
                                      S * H * D, D, B, H, S, D, L, 1.0, st), "dV")
        check(lib.immtsf_softmax_rows_backward(ptr(dA), ptr(P), B, H, L, S, p, seed, site, cnt, st), "softmax_bwd")
        # dQ = scale dS K ; dK = scale dS^T Q
        check(lib.immtsf_gemm_batched(1, precision, ptr(dA), S, H * L * S, L * S, ptr(k), H * E, S * H * E, E, ptr(dq), H * E,
                                      L * H * E, E, B, H, L, E, S, float(scale), st), "dQ")
        check(lib.immtsf_gemm_batched(2, precision, ptr(dA), S, H * L * S, L * S, ptr(q), H * E, L * H * E, E, ptr(dk), H * E,
                                      S * H * E, E, B, H, S, E, L, float(scale), st), "dK")
        return dq, dk, dv, None, None, None, None, None, None, None


class FullAttentionQKVFn(torch.autograd.Function):
    """self-attention on a PACKED in-projection output qkv (B, L, 3, H, E) (what `x @ in_proj_weight^T` yields): the
    batched GEMMs read q / k / v through strides and the backward writes dq / dk / dv straight into one (B,L,3,H,E)
    tensor -- no slice copies forward, no zero-fill + copy + add per slice backward."""

    @staticmethod
    def forward(ctx, qkv, scale, p_drop, training, seed, site, causal, precision):
        lib = _lib.load()
        qkv = _c(qkv)
        _need_gpu(qkv)
        B, L, three, H, E = qkv.shape
        assert three == 3
        dev = qkv.device
        ld, so = 3 * H * E, L * 3 * H * E
        q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * H * E, qkv.data_ptr() + 8 * H * E
        P = torch.empty(B, H, L, L, dtype=torch.float32, device=dev)
        check(lib.immtsf_gemm_batched(0, precision, C.c_void_p(q), ld, so, E, C.c_void_p(k), ld, so, E, ptr(P), L, H * L * L, L * L,
                                      B, H, L, L, E, float(scale), stream_ptr()), "qk^T")
        p = float(p_drop) if training else 0.0
        A = torch.empty_like(P) if p > 0 else P
        cnt = config.dropout_counter_ptr(dev) if p > 0 else None
        check(lib.immtsf_softmax_rows_forward(ptr(P), ptr(A), B, H, L, L, None, p, seed, site, 1 if causal else 0, cnt,
                                              stream_ptr()), "softmax")
        out = torch.empty(B, L, H, E, dtype=torch.float32, device=dev)
        check(lib.immtsf_gemm_batched(1, precision, ptr(A), L, H * L * L, L * L, C.c_void_p(v), ld, so, E, ptr(out), H * E,
                                      L * H * E, E, B, H, L, E, L, 1.0, stream_ptr()), "a.v")
        ctx.save_for_backward(qkv, P, A)
        ctx.cfg = (scale, p, seed, site, precision, cnt)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        qkv, P, A = ctx.saved_tensors
        scale, p, seed, site, precision, cnt = ctx.cfg
        B, L, _, H, E = qkv.shape
        ld, so = 3 * H * E, L * 3 * H * E
        q, k, v = qkv.data_ptr(), qkv.data_ptr() + 4 * H * E, qkv.data_ptr() + 8 * H * E
        dout = dout.contiguous()
        dA = torch.empty_like(P)
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + 4 * H * E, dqkv.data_ptr() + 8 * H * E
        st = stream_ptr()
        check(lib.immtsf_gemm_batched(0, precision, ptr(dout), H * E, L * H * E, E, C.c_void_p(v), ld, so, E, ptr(dA), L,
                                      H * L * L, L * L, B, H, L, L, E, 1.0, st), "dA")
        check(lib.immtsf_gemm_batched(2, precision, ptr(A), L, H * L * L, L * L, ptr(dout), H * E, L * H * E, E, C.c_void_p(dv), ld,
                                      so, E, B, H, L, E, L, 1.0, st), "dV")
        check(lib.immtsf_softmax_rows_backward(ptr(dA), ptr(P), B, H, L, L, p, seed, site, cnt, st), "softmax_bwd")
        check(lib.immtsf_gemm_batched(1, precision, ptr(dA), L, H * L * L, L * L, C.c_void_p(k), ld, so, E, C.c_void_p(dq), ld, so, E,
                                      B, H, L, E, L, float(scale), st), "dQ")
        check(lib.immtsf_gemm_batched(2, precision, ptr(dA), L, H * L * L, L * L, C.c_void_p(q), ld, so, E, C.c_void_p(dk), ld, so, E,
                                      B, H, L, E, L, float(scale), st), "dK")
        return dqkv, None, None, None, None, None, None, None


class SharedKVAttentionFn(torch.autograd.Function):
    """softmax(scale * q k^T) v where the keys / values are ONE set shared by the whole batch (TimeLLM's ReprogrammingLayer,
    models/TimeLLM.py:43-61: the mapped word prototypes): q (B,L,H,E), k (S,H,E), v (S,H,E) -> (B,L,H,E).  Per head one MFMA
    GEMM over all B*L query rows (batched over the heads through strides), the fused softmax + dropout row kernel, and one
    GEMM back; the scores live as (H, B*L, S)."""

    @staticmethod
    def forward(ctx, q, k, v, scale, p_drop, training, seed, site, precision):
        lib = _lib.load()
        q, k, v = _c(q), _c(k), _c(v)
        _need_gpu(q, k, v)
        B, L, H, E = q.shape
        S = k.shape[0]
        R = B * L
        dev = q.device
        P = torch.empty(H, R, S, dtype=torch.float32, device=dev)
        st = stream_ptr()
        check(lib.immtsf_gemm_batched(0, precision, ptr(q), H * E, 0, E, ptr(k), H * E, 0, E, ptr(P), S, 0, R * S, 1, H, R, S, E,
                                      float(scale), st), "q k^T")
        p = float(p_drop) if training else 0.0
        A = torch.empty_like(P) if p > 0 else P
        cnt = config.dropout_counter_ptr(dev) if p > 0 else None
        check(lib.immtsf_softmax_rows_forward(ptr(P), ptr(A), 1, H, R, S, None, p, seed, site, 0, cnt, st), "softmax")
        out = torch.empty(B, L, H, E, dtype=torch.float32, device=dev)
        check(lib.immtsf_gemm_batched(1, precision, ptr(A), S, 0, R * S, ptr(v), H * E, 0, E, ptr(out), H * E, 0, E, 1, H, R, E, S,
                                      1.0, st), "a v")
        ctx.save_for_backward(q, k, v, P, A)
        ctx.cfg = (scale, p, seed, site, precision, cnt)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        q, k, v, P, A = ctx.saved_tensors
        scale, p, seed, site, precision, cnt = ctx.cfg
        B, L, H, E = q.shape
        S, R = k.shape[0], B * L
        dout = dout.contiguous()
        dA = torch.empty_like(P)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        st = stream_ptr()
        check(lib.immtsf_gemm_batched(0, precision, ptr(dout), H * E, 0, E, ptr(v), H * E, 0, E, ptr(dA), S, 0, R * S, 1, H, R, S, E,
                                      1.0, st), "dA")
        check(lib.immtsf_gemm_batched(2, precision, ptr(A), S, 0, R * S, ptr(dout), H * E, 0, E, ptr(dv), H * E, 0, E, 1, H, S, E, R,
                                      1.0, st), "dV")
        check(lib.immtsf_softmax_rows_backward(ptr(dA), ptr(P), 1, H, R, S, p, seed, site, cnt, st), "softmax_bwd")
        check(lib.immtsf_gemm_batched(1, precision, ptr(dA), S, 0, R * S, ptr(k), H * E, 0, E, ptr(dq), H * E, 0, E, 1, H, R, E, S,
                                      float(scale), st), "dQ")
        check(lib.immtsf_gemm_batched(2, precision, ptr(dA), S, 0, R * S, ptr(q), H * E, 0, E, ptr(dk), H * E, 0, E, 1, H, S, E, R,
                                      float(scale), st), "dK")
        return dq, dk, dv, None, None, None, None, None, None


def shared_kv_attention(q, k, v, scale, p_drop=0.0, training=False, seed=0, site=16, precision=None):
    return SharedKVAttentionFn.apply(q.float(), k.float(), v.float(), scale, p_drop, training, seed, site,
                                     config.precision_code(precision))


ATTN_SHORT_MAX = 8      # IMMTSF_ATTN_SHORT_MAX


class ShortAttentionQKVFn(torch.autograd.Function):
    """FullAttentionQKVFn for sequences of at most ATTN_SHORT_MAX positions: one thread per (sequence, head, position), one
    launch per direction, exact fp32, same dropout stream; the backward recomputes the softmax (nothing but qkv is saved)."""

    @staticmethod
    def forward(ctx, qkv, scale, p_drop, training, seed, site, causal):
        lib = _lib.load()
        qkv = _c(qkv)
        _need_gpu(qkv)
        B, L, three, H, E = qkv.shape
        assert three == 3 and L <= ATTN_SHORT_MAX
        p = float(p_drop) if training else 0.0
        cnt = config.dropout_counter_ptr(qkv.device) if p > 0 else None
        out = torch.empty(B, L, H, E, dtype=torch.float32, device=qkv.device)
        check(lib.immtsf_attention_short_forward(ptr(qkv), B, L, H, E, float(scale), 1 if causal else 0, p, seed, site, cnt,
                                                 ptr(out), stream_ptr()), "attention_short_forward")
        ctx.save_for_backward(qkv)
        ctx.cfg = (float(scale), p, seed, site, 1 if causal else 0, cnt)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (qkv,) = ctx.saved_tensors
        scale, p, seed, site, causal, cnt = ctx.cfg
        B, L, _, H, E = qkv.shape
        dqkv = torch.empty_like(qkv)
        check(lib.immtsf_attention_short_backward(ptr(qkv), ptr(dout.contiguous()), B, L, H, E, scale, causal, p, seed, site, cnt,
                                                  ptr(dqkv), stream_ptr()), "attention_short_backward")
        return dqkv, None, None, None, None, None, None


def full_attention_qkv(qkv, scale, p_drop=0.0, training=False, seed=0, site=16, causal=False, precision=None):
    """qkv (B, L, 3, H, E) -> (B, L, H, E)"""
    if qkv.shape[1] <= ATTN_SHORT_MAX and qkv.shape[4] <= 64 and qkv.shape[4] % 4 == 0:
        return ShortAttentionQKVFn.apply(qkv.float(), scale, p_drop, training, seed, site, causal)
    return FullAttentionQKVFn.apply(qkv.float(), scale, p_drop, training, seed, site, causal, config.precision_code(precision))


def full_attention(q, k, v, scale, p_drop=0.0, training=False, seed=0, site=16, causal=False, precision=None):
    return FullAttentionFn.apply(q.float(), k.float(), v.float(), scale, p_drop, training, seed, site, causal,
                                 config.precision_code(precision))


# ------------------------------------------------------------------------------------------------ embeddings (a12 / a13)
EMBED_KMAX, EMBED_DMAX = 64, 512      # csrc/embed.hip limits
SITE_LAYER_BASE = 16                  # csrc/common.hpp: dropout sites of the layers/ modules


class EmbedFn(torch.autograd.Function):
    """immtsf_embed_forward/backward: mode 0 = replication pad + unfold + Linear(patch_len -> D) + pe + dropout on (R, L)
    rows, mode 1 = circular 3-tap token convolution + pe + dropout on (B, L, c_in); W (D, K) resp. (D, c_in, 3)."""

    @staticmethod
    def forward(ctx, x, W, pe, mode, P, K, stride, p_drop, training, seed, site):
        lib = _lib.load()
        x, W, pe = _c(x), _c(W), _c(pe)
        _need_gpu(x, W, pe)
        if mode == 0:
            R, L, c_in = x.shape[0], x.shape[1], 1
        else:
            R, L, c_in = x.shape
        D = W.shape[0]
        p = float(p_drop) if training else 0.0
        cnt = config.dropout_counter_ptr(x.device) if p > 0 else None
        out = torch.empty(R, P, D, dtype=torch.float32, device=x.device)
        check(lib.immtsf_embed_forward(mode, ptr(x), R, L, c_in, P, K, stride, D, ptr(W), ptr(pe), ptr(out), p, seed, site, cnt,
                                       stream_ptr()), "embed_forward")
        ctx.save_for_backward(x, W)
        ctx.cfg = (mode, R, L, c_in, P, K, stride, D, p, seed, site, cnt)
        ctx.sink = getattr(W, "_immtsf_grad_sink", None) if getattr(W, "_immtsf_grad_prezeroed", False) else None
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, W = ctx.saved_tensors
        mode, R, L, c_in, P, K, stride, D, p, seed, site, cnt = ctx.cfg
        dW = ctx.sink if ctx.sink is not None else torch.empty_like(W)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        check(lib.immtsf_embed_backward(mode, ptr(x), R, L, c_in, P, K, stride, D, ptr(W), ptr(dout.contiguous()), ptr(dW),
                                        1 if ctx.sink is not None else 0, ptr(dx), p, seed, site, cnt, stream_ptr()), "embed_backward")
        return (dx, None if ctx.sink is not None else dW) + (None,) * 9


def embed_supported(K, D):
    return K <= EMBED_KMAX and D <= EMBED_DMAX


def patch_embed(x, W, pe, patch_len, stride, pad, p_drop=0.0, training=False, site=None):
    """x (B, n_vars, L) -> (B * n_vars, P, D): PatchEmbedding (layers/Embed.py:165-190) in one kernel"""
    B, n_vars, L = x.shape
    P = (L + pad - patch_len) // stride + 1
    p = float(p_drop) if training else 0.0
    return EmbedFn.apply(x.float().reshape(B * n_vars, L), W, pe.reshape(-1, pe.shape[-1]), 0, P, patch_len, stride, p, training,
                         config.next_seed() if p > 0 else 0, SITE_LAYER_BASE + 40 if site is None else site)


def token_embed(x, W, pe, p_drop=0.0, training=False, site=None):
    """x (B, L, c_in) -> (B, L, D): TokenEmbedding + PositionalEmbedding + dropout of DataEmbedding in one kernel"""
    B, L, c_in = x.shape
    p = float(p_drop) if training else 0.0
    return EmbedFn.apply(x.float(), W, pe.reshape(-1, pe.shape[-1]), 1, L, 3 * c_in, 1, p, training,
                         config.next_seed() if p > 0 else 0, SITE_LAYER_BASE + 41 if site is None else site)


# ------------------------------------------------------------------------------------------------ loss
class MaskedMSEFn(torch.autograd.Function):
    """compute_error(truth, pred, mask, "MSE", "mean") (lib/evaluation.py:17-62) with a fused backward.

    With `group` (a torch.distributed process group) the per-variable (sum, count) pair is all-reduced before the
    divide, so every rank sees the loss of the global batch and sum-reduced gradients equal the single-process ones."""

    @staticmethod
    def forward(ctx, pred, truth, mask, group, global_cnt):
        lib = _lib.load()
        pred, truth, mask = _c(pred), _c(truth), _c(mask)
        _need_gpu(pred, truth, mask)
        Cc = pred.shape[-1]
        rows = pred.numel() // Cc
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        if group is None and global_cnt is not None and Cc <= 64:
            # counts known beforehand: nothing waits for a reduction, several workgroups, one launch
            cnt = _c(global_cnt.to(torch.float32))
            check(lib.immtsf_masked_mse_counted(ptr(truth), ptr(pred), ptr(mask), rows, Cc, ptr(cnt), ptr(_mse_scratch(pred.device)),
                                                ptr(loss), ptr(dpred), 1.0, stream_ptr()), "masked_mse_counted")
            ctx.save_for_backward(dpred)
            return loss
        if group is None and rows * Cc <= _MSE_SMALL_MAX and Cc <= 4096:
            # one single-workgroup kernel instead of three launches (they sit between the forward and the backward)
            cnt = _c(global_cnt.to(torch.float32)) if global_cnt is not None else None
            check(lib.immtsf_masked_mse(ptr(truth), ptr(pred), ptr(mask), rows, Cc, ptr(cnt) if cnt is not None else None, None,
                                        None, ptr(loss), ptr(dpred), 1.0, stream_ptr()), "masked_mse")
            ctx.save_for_backward(dpred)
            return loss
        buf = torch.empty(2 + 128, Cc, dtype=torch.float32, device=pred.device)
        sums = buf[:2]
        check(lib.immtsf_masked_mse_sums(ptr(truth), ptr(pred), ptr(mask), rows, Cc, ptr(sums[0]), ptr(sums[1]),
                                         ptr(buf[2]), stream_ptr()), "masked_mse_sums")
        cnt = sums[1]
        if global_cnt is not None:
            cnt = _c(global_cnt.to(torch.float32))
        elif group is not None:
            import torch.distributed as dist
            dist.all_reduce(sums, group=group)
        check(lib.immtsf_masked_mse_finish(ptr(truth), ptr(pred), ptr(mask), rows, Cc, ptr(sums[0]), ptr(cnt), ptr(loss),
                                           ptr(dpred), 1.0, stream_ptr()), "masked_mse_finish")
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dpred,) = ctx.saved_tensors
        if is_unit_grad(dloss):      # d(loss)/d(loss) = 1 from backward_unit(): the saved gradient is the answer
            return dpred, None, None, None, None
        return dpred * dloss, None, None, None, None


_MSE_SMALL_MAX = 1 << 17        # IMMTSF_MSE_SMALL_MAX
_mse_scratch_cache = {}


def _mse_scratch(device) -> torch.Tensor:
    """Ticket word + partial losses of immtsf_masked_mse_counted: one zero-initialised buffer per device, kept for the life of
    the process (a captured graph keeps pointing at it).  One loss per step and process: calls are ordered on their stream; two
    concurrent calls on different streams of one device would need their own buffers (the C entry point takes any)."""
    key = (device.type, device.index)
    t = _mse_scratch_cache.get(key)
    if t is None:
        t = _mse_scratch_cache[key] = torch.zeros(65, dtype=torch.float32, device=device)
    return t
_unit_grads = {}


def unit_grad(device) -> torch.Tensor:
    """The constant scalar 1.0 that `backward_unit` seeds the backward pass with (one per device, never written)."""
    key = (device.type, device.index)
    t = _unit_grads.get(key)
    if t is None:
        t = _unit_grads[key] = torch.ones((), dtype=torch.float32, device=device)
    return t


def is_unit_grad(t: torch.Tensor) -> bool:
    u = _unit_grads.get((t.device.type, t.device.index))
    return u is not None and t.data_ptr() == u.data_ptr() and t.dim() == 0


def backward_unit(loss: torch.Tensor):
    """loss.backward() seeded with the cached constant 1.0: no fill kernel for the seed, and loss functions that saved
    their input gradient (masked_mse) hand it on without a multiply -- two launches less between forward and backward."""
    loss.backward(gradient=unit_grad(loss.device))


def masked_mse(pred, truth, mask, group=None, global_cnt=None):
    """Data parallel, two ways to get single-process-equivalent gradients:
      group=...       all-reduce the per-variable (sum, count) inside the call (loss = global loss on every rank);
      global_cnt=...  per-variable observation counts of the GLOBAL batch, reduced once when the batch is built
                      (they depend on the mask only): no collective inside the step, the returned value is this
                      rank's share of the global loss (the shares add up to it)."""
    return MaskedMSEFn.apply(pred, truth, mask, group, global_cnt)


def dropout_keep_mask(seed: int, site: int, n: int, p: float, device) -> torch.Tensor:
    """uint8 keep-mask of `n` consecutive elements of a dropout site (for tests / mask export)."""
    lib = _lib.load()
    out = torch.empty(n, dtype=torch.uint8, device=device)
    check(lib.immtsf_dropout_mask(int(seed) & 0xFFFFFFFFFFFFFFFF, site, n, float(p), ptr(out), stream_ptr()), "dropout_mask")
    return out
