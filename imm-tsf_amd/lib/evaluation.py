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
    E_txt, M_txt = fusion.ttf(notes, tau, tp)
    kv = fusion.mmf.project_kv(E_txt) if hasattr(fusion.mmf, "project_kv") else None    # text-only half of the MMF block
    with torch.cuda.stream(side_stream):
        pred_y = model.forecasting(*fc_args)
    main.wait_stream(side_stream)
    pred_y.record_stream(main)
    if loss is not None and hasattr(fusion.mmf, "forward_loss"):
        return fusion.mmf.forward_loss(pred_y, E_txt, M_txt, loss[0], loss[1], loss[2], kv=kv)
    out = fusion.mmf(pred_y, E_txt, M_txt) if kv is None else fusion.mmf(pred_y, E_txt, M_txt, kv=kv)
    return out if loss is None else masked_mse(out, loss[0], loss[1], None, loss[2])


def compute_all_losses(model, fusion, batch_dict, enable_text=True, use_text_embeddings=True, group=None):
    """One training-step forward: backbone forecast -> fusion -> masked MSE (lib/evaluation.py:72-164).
    The reference's per-step host syncs (NaN checks, per-row mask loop, .item()) follow immtsf.config.nan_check:
    in "sync" mode they are all performed; otherwise the loss stays on the device (results["mse"] is a tensor)."""
    sync = config.nan_check == "sync"
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
