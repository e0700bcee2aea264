"""Loss / train-step functions with the reference's names and semantics (lib/evaluation.py:17-164).

`compute_error(..., "MSE", "mean")` on GPU tensors that require grad goes through the fused HIP masked-MSE
(immtsf.ops.masked_mse: per-variable sums -> optional all-reduce -> loss and d(pred) in one more kernel); every
other combination is evaluated with plain torch ops (metrics only, off the hot path).
"""
import os

import torch

from immtsf import config
from immtsf.ops import masked_mse


def compute_error(truth, pred_y, mask, func, reduce, norm_dict=None, group=None):
    if func == "MSE" and reduce == "mean" and pred_y.is_cuda and norm_dict is None and pred_y.dim() == 3:
        return masked_mse(pred_y, truth, mask.to(pred_y.dtype), group)
    if pred_y.dim() == 3:
        pred_y = pred_y.unsqueeze(0)
    n_dim = pred_y.shape[-1]
    truth_r = truth.unsqueeze(0).expand_as(pred_y)
    mask = mask.unsqueeze(0).expand_as(pred_y)
    if func == "MSE":
        error = (truth_r - pred_y) ** 2 * mask
    elif func == "MAE":
        error = (truth_r - pred_y).abs() * mask
    elif func == "MAPE":
        if norm_dict is None:
            mask = (truth_r != 0) * mask
            error = (truth_r - pred_y).abs() / (truth_r + (truth_r == 0) * 1e-8) * mask
        else:
            lo, hi = norm_dict["data_min"], norm_dict["data_max"]
            t, p = truth_r * (hi - lo) + lo, pred_y * (hi - lo) + lo
            mask = (t != 0) * mask
            error = (t - p).abs() / (t + (t == 0) * 1e-8) * mask
    else:
        raise Exception("Error function not specified")
    err_sum = error.reshape(-1, n_dim).sum(dim=0)
    cnt = mask.reshape(-1, n_dim).sum(dim=0)
    if reduce == "mean":
        return (err_sum / (cnt + 1e-8)).sum() / torch.count_nonzero(cnt)
    if reduce == "sum":
        return err_sum, cnt
    raise Exception("Reduce argument not specified!")


def forecast_and_fuse(model, fusion, batch_dict, side_stream=None, loss=None):
    """backbone forecast -> fusion.  The backbone and the text-timestamp fusion (TTF) do not depend on each other
    -- only the modality fusion (MMF) needs both -- so with `side_stream` the backbone is enqueued on that HIP stream
    while TTF runs on the current one, joined before MMF.  autograd replays each backward on its forward's stream, so
    the two backward halves overlap the same way.  Both halves are latency-bound at 64 windows; overlapping them is
    worth more than any single kernel.

    loss = (truth, mask, global_cnt): return masked_mse(fused forecast, truth, mask, global_cnt=global_cnt) instead of the forecast
    -- a fusion whose last block can run its head, that loss and the backward of both as one launch (MMF_XAttn_Add.forward_loss)
    does so."""
    from immtsf.ops import masked_mse
    notes, tau, tp = batch_dict["notes_embeddings"], batch_dict["tau"], batch_dict["tp_to_predict"]
    fc_args = (tp, batch_dict["observed_data"], batch_dict["observed_tp"], batch_dict["observed_mask"])
    if side_stream is None or not hasattr(fusion, "ttf"):
        out = fusion(notes, tau, tp, model.forecasting(*fc_args))
        return out if loss is None else masked_mse(out, loss[0], loss[1], None, loss[2])
    main = torch.cuda.current_stream()
    side_stream.wait_stream(main)
    # Host order: text side first, backbone second.  autograd runs ready backward nodes in reverse creation order, so the
    # backbone's backward -- the longer, latency-bound chain, which only needs dY_ts from the MMF query half -- is then
    # enqueued (and, under hipGraph capture, placed in the graph's submission order) BEFORE the key/value-half and TTF
    # backward instead of behind them (r02 trace: placed last it started 200 us after its input was ready).  Re-measured after
    # the launch cuts of r02 (backbone first / text first): 0.908 / 0.906 ms per step -- no difference any more.
    E_txt, M_txt, kv = fusion.text_side(notes, tau, tp)      # TTF + the text-only half of the MMF block
    with torch.cuda.stream(side_stream):
        pred_y = model.forecasting(*fc_args)
    main.wait_stream(side_stream)
    pred_y.record_stream(main)
    if loss is not None and hasattr(fusion.mmf, "forward_loss"):
        return fusion.mmf.forward_loss(pred_y, E_txt, M_txt, loss[0], loss[1], loss[2], kv=kv)
    out = fusion.mmf(pred_y, E_txt, M_txt) if kv is None else fusion.mmf(pred_y, E_txt, M_txt, kv=kv)
    return out if loss is None else masked_mse(out, loss[0], loss[1], None, loss[2])


# ---- the zero-edit seam as a replayed hipGraph --------------------------------------------------------------------------------
# An unmodified main.py calls compute_all_losses(...), loss.backward(), clip_grad_norm_, optimizer.step() (main.py:1093-1101): eager,
# that is ~100 launches enqueued from Python -- host-bound at 3 ms per step.  With IMMTSF_NAN_CHECK=deferred (no host syncs in the step)
# the forward, the loss and the WHOLE backward of a (model, fusion, batch shape) that has been seen before are captured once into a
# hipGraph over static copies of the batch; a call then copies the batch in, replays the graph, and returns a loss whose backward()
# only hands the gradients the graph computed to the parameters' .grad (added to what is there, like autograd does).  Same kernels,
# same results as the eager call (tests/test_gpu_train.py::test_dropin_seam_graph_replay_equals_eager); clip and optimizer stay the
# caller's (immtsf.optim routes them to the fused kernels).  A new batch shape is run eagerly the first time and captured the second.
#
# The cache lives with the MODEL (a WeakKeyDictionary: it dies with the module -- an id()-keyed cache would serve a later model that
# happens to get the same address the old model's graph, computing on the old parameters); an entry is valid only for the very fusion
# module, parameter storages, requires_grad pattern and configuration knobs it was captured with; entries are evicted least recently
# used first under a byte budget (static batch copies: the note embeddings can be hundreds of MB per shape) and a count cap; all seam
# graphs share one memory pool (they never replay concurrently).
import weakref

_GRAPH_CAP = 8
_GRAPH_BYTES = int(os.environ.get("IMMTSF_SEAM_GRAPH_BYTES", str(4 << 30)))      # budget for the static batch copies of all cached graphs
_SEEN_CAP = 64
_graphs = weakref.WeakKeyDictionary()      # model -> {key: _SeamGraph}, most recently used last
_seen = weakref.WeakKeyDictionary()        # model -> {key: eager calls so far} (bounded)
_pool = {}                                 # device index -> shared graph memory pool handle
_pool_users = {}                           # device index -> live seam graphs in that pool (the handle dies with the last of them)


def _pool_release(index):
    _pool_users[index] = _pool_users.get(index, 1) - 1
    if _pool_users[index] <= 0:            # the allocator retires a pool nobody captures into: a later graph must not be given its handle
        _pool.pop(index, None)
        _pool_users.pop(index, None)


class _SeamLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, g):
        ctx.g, ctx.replay = g, g.replays
        return g.loss.detach().clone()

    @staticmethod
    def backward(ctx, dloss):
        g = ctx.g
        if g.replays != ctx.replay:
            # the static gradient buffers hold the LAST replay's gradients: a loss from an earlier call of the same shape (gradient
            # accumulation: two compute_all_losses calls, then backward) would hand on the other batch's gradients without a word
            raise RuntimeError("immtsf seam graph: compute_all_losses() ran again for this batch shape before this loss's backward(); "
                               "call backward() after every compute_all_losses(), or set immtsf.config.seam_graph = False")
        ps = [p for p, gr in zip(g.params, g.grads) if gr is not None]
        gs = [gr for gr in g.grads if gr is not None]
        # the static gradient buffers are rewritten by the next replay: the parameters get their own copies, scaled by the seed --
        # multi-tensor launches, not one kernel per parameter (a unit seed -- loss.backward() -- needs no scaling pass)
        from immtsf.ops import is_unit_grad
        unit = is_unit_grad(dloss)
        fresh = [(p, v) for p, v in zip(ps, gs) if p.grad is None]
        old = [(p, v) for p, v in zip(ps, gs) if p.grad is not None]
        if fresh:
            vals = [v.clone() for _, v in fresh] if unit else torch._foreach_mul([v for _, v in fresh], dloss)
            for (p, _), v in zip(fresh, vals):
                p.grad = v
        if old:
            if unit:
                torch._foreach_add_([p.grad for p, _ in old], [v for _, v in old])
            else:
                torch._foreach_add_([p.grad for p, _ in old], torch._foreach_mul([v for _, v in old], dloss))
        return None, None


class _SeamGraph:
    def __init__(self, model, fusion, batch_dict, names):
        from immtsf import ops
        dev = batch_dict["tp_to_predict"].device
        self.names = names
        self.static = {k: batch_dict[k].detach().clone() for k in names}
        self.bytes = sum(v.numel() * v.element_size() for v in self.static.values())
        self.fusion_ref = weakref.ref(fusion)
        self.replays = 0
        from torch.nn.utils import stateless
        named_m = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
        named_f = [(k, p) for k, p in fusion.named_parameters() if p.requires_grad]
        self.params = [p for _, p in named_m + named_f]
        self.unit = ops.unit_grad(dev)
        self.anchor = torch.zeros((), device=dev, requires_grad=True)
        _, self.drop_dev = config.enable_device_counters(dev)       # the dropout key advances on the device, once per replay
        C = self.static["mask_predicted_data"].shape[-1]

        def run():
            b = self.static
            cnt = b["mask_predicted_data"].reshape(-1, C).sum(0)
            # The captured forward sees ALIASES of the parameters (detached views of the same storage, fresh autograd leaves), and the
            # gradients come back through torch.autograd.grad.  Why: a parameter's gradient-accumulator node remembers the stream it
            # was created on and is shared by every live autograd graph that uses the parameter -- e.g. the caller's previous eager
            # `loss`, still referenced while this call runs (main.py:1093 reassigns it afterwards).  The engine would synchronise the
            # capturing stream with THAT stream: an event recorded outside the capture, and hipStreamEndCapture crashes (ROCm 7.2).
            al_m = {k: p.detach().requires_grad_(True) for k, p in named_m}
            al_f = {k: p.detach().requires_grad_(True) for k, p in named_f}
            with stateless._reparametrize_module(model, al_m), stateless._reparametrize_module(fusion, al_f):
                # (one stream: what the seam buys is the host side -- one graph launch instead of ~100 enqueued from Python)
                loss = forecast_and_fuse(model, fusion, b, None, loss=(b["data_to_predict"], b["mask_predicted_data"], cnt))
            grads = torch.autograd.grad(loss, list(al_m.values()) + list(al_f.values()), grad_outputs=self.unit, allow_unused=True)
            self.drop_dev.add_(1)
            return loss, grads

        # the warm-up runs are REAL forward / backward passes: what they change besides the (discarded) gradients -- module buffers such
        # as BatchNorm's running statistics, the dropout key counter -- is put back, so that building the graph is invisible to training
        bufs = [b_ for b_ in list(model.buffers()) + list(fusion.buffers())]
        saved = [b_.detach().clone() for b_ in bufs]
        drop0 = self.drop_dev.clone()
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for _ in range(2):
                run()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        with torch.no_grad():
            for b_, s_ in zip(bufs, saved):
                b_.copy_(s_)
            self.drop_dev.copy_(drop0)
        torch.cuda.synchronize()
        pool = _pool.get(dev.index)
        if pool is None:
            pool = _pool[dev.index] = torch.cuda.graph_pool_handle()
        _pool_users[dev.index] = _pool_users.get(dev.index, 0) + 1
        weakref.finalize(self, _pool_release, dev.index)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=pool):
            self.loss, self.grads = run()

    def __call__(self, batch_dict):
        torch._foreach_copy_([self.static[k] for k in self.names], [batch_dict[k] for k in self.names])
        self.graph.replay()
        self.replays += 1
        return _SeamLoss.apply(self.anchor, self)


def _seam_key(model, fusion, batch_dict, names):
    """everything a captured graph bakes in besides the batch's VALUES: the fusion module (checked by weak reference on lookup), the
    train flags, the configuration knobs that pick kernels, the batch's shapes / dtypes, and -- per parameter -- where its storage lives
    and whether it takes a gradient (a re-bound or frozen parameter must not meet a graph captured before)"""
    sig = tuple((p.data_ptr(), p.requires_grad) for m in (model, fusion) for p in m.parameters())
    return (id(fusion), model.training, fusion.training, config.precision, config.t2v_form, config.fuse_tail, config.xattn_rank,
            config.xattn_fused_loss, config.attn_mid, config.note_index, sig,
            tuple((k, tuple(batch_dict[k].shape), batch_dict[k].dtype) for k in names))


def _seam_lookup(model, fusion, batch_dict, names):
    """the cached graph for this call, building it on the second sighting of a key; None: run eagerly"""
    key = _seam_key(model, fusion, batch_dict, names)
    cache = _graphs.setdefault(model, {})
    g = cache.get(key)
    if g is not None and g.fusion_ref() is not fusion:          # (the id was recycled by another fusion module)
        del cache[key]
        g = None
    if g is not None:
        cache[key] = cache.pop(key)                              # most recently used last
        return g
    seen = _seen.setdefault(model, {})
    n = seen.get(key, 0)
    if n < 1:
        if len(seen) >= _SEEN_CAP:
            seen.pop(next(iter(seen)))
        seen[key] = n + 1
        return None
    need = sum(batch_dict[k].numel() * batch_dict[k].element_size() for k in names)
    if need > _GRAPH_BYTES:
        return None
    while cache and (len(cache) >= _GRAPH_CAP or sum(x.bytes for x in cache.values()) + need > _GRAPH_BYTES):
        cache.pop(next(iter(cache)))                             # evict the least recently used shape
    g = cache[key] = _SeamGraph(model, fusion, batch_dict, names)
    return g


# ---- deferred NaN guards ----------------------------------------------------------------------------------------------------------
# nan_check "deferred" (the default): the reference's guards (fusions/FusionModel.py:103-112, lib/evaluation.py:107-160) raise the same
# ValueError, but at the NEXT host-visible point instead of inside the step: every call leaves (is the loss NaN, did the kernels see a
# NaN note embedding) in pinned host memory by an asynchronous copy; the next compute_all_losses() / evaluation() call looks at what
# the PREVIOUS step left -- main.py's own `loss.item()` (main.py:1104) has synchronised by then -- and raises.  No host sync of its own.
_probe = {}       # device index -> (device int32[2], pinned int32[2], event or None)


def _probe_note(loss, fusion):
    dev = loss.device
    slot = _probe.get(dev.index)
    if slot is None:
        slot = _probe[dev.index] = [torch.zeros(2, dtype=torch.int32, device=dev), torch.zeros(2, dtype=torch.int32).pin_memory(), None]
    d, h, _ = slot
    d[0:1].copy_(torch.isnan(loss.detach()).reshape(1))
    flag = getattr(getattr(getattr(fusion, "ttf", None), "_nan", None), "flag", None) if fusion is not None else None
    if flag is not None:
        d[1:2].copy_(flag.reshape(1))
    h.copy_(d, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    slot[2] = ev


def check_deferred_nan(fusion=None):
    """raise the reference's ValueError for what an EARLIER step left behind (deferred NaN guards); never blocks: a probe whose copy has
    not landed yet is looked at by the next call"""
    for slot in _probe.values():
        d, h, ev = slot
        if ev is None or not ev.query():
            continue
        slot[2] = None
        bad_loss, bad_notes = int(h[0]), int(h[1])
        if bad_notes:
            f = getattr(getattr(fusion, "ttf", None), "_nan", None) if fusion is not None else None
            if f is not None and f.flag is not None:
                f.flag.zero_()
            h.zero_()
            raise ValueError("Input embeddings V contain NaN values.")
        if bad_loss:
            h.zero_()
            raise ValueError("MSE is NaN")


def compute_all_losses(model, fusion, batch_dict, enable_text=True, use_text_embeddings=True, group=None):
    """One training-step forward: backbone forecast -> fusion -> masked MSE (lib/evaluation.py:72-164).
    The reference's per-step host syncs (NaN checks, per-row mask loop, .item()) follow immtsf.config.nan_check:
    in "sync" mode they are all performed; otherwise the loss stays on the device (results["mse"] is a tensor) and -- config.seam_graph
    -- a repeated (model, fusion, batch shape) is served by a replayed hipGraph (see above)."""
    sync = config.nan_check == "sync"
    if config.nan_check == "deferred":
        check_deferred_nan(fusion)
    names = ("tp_to_predict", "observed_data", "observed_tp", "observed_mask", "notes_embeddings", "tau", "data_to_predict",
             "mask_predicted_data")
    if (not sync and config.seam_graph and enable_text and fusion is not None and use_text_embeddings and group is None and
            torch.is_grad_enabled() and model.training and hasattr(fusion, "ttf") and getattr(model, "immtsf_graphable", False) and
            all(torch.is_tensor(batch_dict.get(k)) and batch_dict[k].is_cuda for k in names)):
        g = _seam_lookup(model, fusion, batch_dict, names)
        if g is not None:
            loss = g(batch_dict)
            if config.nan_check == "deferred":
                _probe_note(loss, fusion)
            return {"loss": loss, "mse": loss.detach()}
    pred_y = model.forecasting(batch_dict["tp_to_predict"], batch_dict["observed_data"], batch_dict["observed_tp"],
                               batch_dict["observed_mask"])
    if sync and torch.isnan(pred_y).any():
        raise ValueError("pred_y contains NaN values.")
    if enable_text and fusion is not None:
        notes = batch_dict["notes_embeddings"] if use_text_embeddings else batch_dict["notes_text"]
        pred_y = fusion(notes, batch_dict["tau"], batch_dict["tp_to_predict"], pred_y)
    if sync:
        if torch.isnan(pred_y).any():
            raise ValueError("pred_y contains NaN values.")
        if torch.isnan(batch_dict["data_to_predict"]).any():
            raise ValueError("data_to_predict contains NaN values.")
        empty = batch_dict["mask_predicted_data"].flatten(1).sum(1) == 0      # one sync instead of B
        if bool(empty.any()):
            i = int(torch.nonzero(empty)[0])
            raise ValueError(f"mask_predicted_data for sample {i} is all zeros: {batch_dict['mask_predicted_data'][i]}")
    mse = compute_error(batch_dict["data_to_predict"], pred_y, mask=batch_dict["mask_predicted_data"], func="MSE",
                        reduce="mean", group=group)
    if sync and torch.isnan(mse).any():
        raise ValueError("MSE is NaN")
    if config.nan_check == "deferred" and mse.is_cuda:
        _probe_note(mse, fusion if enable_text else None)
    return {"loss": mse, "mse": mse.item() if sync else mse.detach()}


def evaluation(model, fusion, dataloader, enable_text=True, use_text_embeddings=True):
    """Test/validation metrics with the reference's definitions (lib/evaluation.py:192-283): per-variable sums of the
    squared / absolute / relative errors and of the observation counts over the whole loader, then mean over the
    variables that were observed.  Everything accumulates on the device; the only host syncs are the final `.item()`s
    (the reference syncs several times per batch).  Returns the same dict of python floats."""
    acc = None
    if config.nan_check == "deferred":
        check_deferred_nan(fusion)
    with torch.no_grad():
        for batch_dict in dataloader:
            pred_y = model.forecasting(batch_dict["tp_to_predict"], batch_dict["observed_data"], batch_dict["observed_tp"],
                                       batch_dict["observed_mask"])
            if enable_text and fusion is not None:
                notes = batch_dict["notes_embeddings"] if use_text_embeddings else batch_dict["notes_text"]
                pred_y = fusion(notes, batch_dict["tau"], batch_dict["tp_to_predict"], pred_y)
            truth, mask = batch_dict["data_to_predict"], batch_dict["mask_predicted_data"]
            se, cnt = compute_error(truth, pred_y, mask, "MSE", "sum")
            ae, _ = compute_error(truth, pred_y, mask, "MAE", "sum")
            ape, cnt_ape = compute_error(truth, pred_y, mask, "MAPE", "sum")
            part = torch.stack([se, ae, ape, cnt, cnt_ape.to(se.dtype)])
            acc = part if acc is None else acc + part
    if acc is None:
        raise ValueError("evaluation(): empty dataloader")
    se, ae, ape, cnt, cnt_ape = acc
    n_var, n_var_ape = torch.count_nonzero(cnt), torch.count_nonzero(cnt_ape)
    mse = (se / (cnt + 1e-8)).sum() / n_var
    mae = (ae / (cnt + 1e-8)).sum() / n_var
    mape = (ape / (cnt_ape + 1e-8)).sum() / n_var_ape
    vals = torch.stack([mse, mse, mae, torch.sqrt(mse), mape]).tolist()        # one device -> host transfer
    return dict(zip(("loss", "mse", "mae", "rmse", "mape"), vals))


from immtsf.dropin import reexport_missing as _reexport_missing  # noqa: E402

_reexport_missing(globals())     # names of the reference module this build does not mirror

# the optimizer side of the zero-edit seam: torch.optim.Adam / clip_grad_norm_ of an unmodified main.py (main.py:1024, 1098-1101) on the
# fused kernels (immtsf/optim.py; IMMTSF_OPTIM_SHIM=0 leaves torch's in place)
from immtsf import optim as _optim  # noqa: E402

_optim.install_from_env()
