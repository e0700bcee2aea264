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
# caller's.  A new batch shape is run eagerly the first time and captured the second; at most _GRAPH_CAP graphs are kept.
_GRAPH_CAP = 8
_graphs = {}      # key -> _SeamGraph
_seen = {}        # key -> number of eager calls so far


class _SeamLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, g):
        ctx.g = g
        return g.loss.detach().clone()

    @staticmethod
    def backward(ctx, dloss):
        g = ctx.g
        ps = [p for p, gr in zip(g.params, g.grads) if gr is not None]
        gs = [gr for gr in g.grads if gr is not None]
        # the static gradient buffers are rewritten by the next replay: the parameters get their own copies, scaled by the seed --
        # multi-tensor launches, not one kernel per parameter
        vals = torch._foreach_mul(gs, dloss)
        fresh = [(p, v) for p, v in zip(ps, vals) if p.grad is None]
        old = [(p, v) for p, v in zip(ps, vals) if p.grad is not None]
        for p, v in fresh:
            p.grad = v
        if old:
            torch._foreach_add_([p.grad for p, _ in old], [v for _, v in old])
        return None, None


class _SeamGraph:
    def __init__(self, model, fusion, batch_dict, names):
        from immtsf import ops
        dev = batch_dict["tp_to_predict"].device
        self.names = names
        self.static = {k: batch_dict[k].detach().clone() for k in names}
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

        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for _ in range(2):
                run()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.grads = run()

    def __call__(self, batch_dict):
        torch._foreach_copy_([self.static[k] for k in self.names], [batch_dict[k] for k in self.names])
        self.graph.replay()
        return _SeamLoss.apply(self.anchor, self)


def _seam_key(model, fusion, batch_dict, names):
    return (id(model), id(fusion), model.training, fusion.training, config.precision, config.t2v_form,
            tuple((k, tuple(batch_dict[k].shape), batch_dict[k].dtype) for k in names))


def compute_all_losses(model, fusion, batch_dict, enable_text=True, use_text_embeddings=True, group=None):
    """One training-step forward: backbone forecast -> fusion -> masked MSE (lib/evaluation.py:72-164).
    The reference's per-step host syncs (NaN checks, per-row mask loop, .item()) follow immtsf.config.nan_check:
    in "sync" mode they are all performed; otherwise the loss stays on the device (results["mse"] is a tensor) and -- config.seam_graph
    -- a repeated (model, fusion, batch shape) is served by a replayed hipGraph (see above)."""
    sync = config.nan_check == "sync"
    names = ("tp_to_predict", "observed_data", "observed_tp", "observed_mask", "notes_embeddings", "tau", "data_to_predict",
             "mask_predicted_data")
    if (not sync and config.seam_graph and enable_text and fusion is not None and use_text_embeddings and group is None and
            torch.is_grad_enabled() and model.training and hasattr(fusion, "ttf") and getattr(model, "immtsf_graphable", False) and
            all(torch.is_tensor(batch_dict.get(k)) and batch_dict[k].is_cuda for k in names)):
        key = _seam_key(model, fusion, batch_dict, names)
        g = _graphs.get(key)
        if g is None and _seen.get(key, 0) >= 1 and len(_graphs) < _GRAPH_CAP:
            g = _graphs[key] = _SeamGraph(model, fusion, batch_dict, names)
        if g is not None:
            loss = g(batch_dict)
            return {"loss": loss, "mse": loss.detach()}
        _seen[key] = _seen.get(key, 0) + 1
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
    return {"loss": mse, "mse": mse.item() if sync else mse.detach()}


def evaluation(model, fusion, dataloader, enable_text=True, use_text_embeddings=True):
    """Test/validation metrics with the reference's definitions (lib/evaluation.py:192-283): per-variable sums of the
    squared / absolute / relative errors and of the observation counts over the whole loader, then mean over the
    variables that were observed.  Everything accumulates on the device; the only host syncs are the final `.item()`s
    (the reference syncs several times per batch).  Returns the same dict of python floats."""
    acc = None
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
