"""Resident dataset store + device-side batch builder.

The reference keeps the dataset as a Python list of per-window tensors and builds every batch on the host with
pad_sequence / Python loops (lib/parse_datasets.py:252-366, 764-824).  Here the whole dataset is laid out ONCE as
window-major CSR arrays in HBM (a 288 GB device holds every dataset the reference ships many times over), the
text-embedding matrices stay resident, and a batch is built by three gather kernels from a list of window ids.
`collate()` returns the same dict the reference's collate returns (bit-exact), plus the packed ragged note index
(`note_lengths`, `note_offsets`, `note_rowmap`) that makes the zero-padded embeddings optional.

Host side (one-time, numpy): per-window metadata -- history/prediction lengths, note counts and the per-window
maximum patch population -- so output shapes are known without a device sync.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Store, check, ptr, stream_ptr


class ResidentStore:
    def __init__(self, tt, vals, mask, row_off, note_tau, note_src, note_off, emb, history, pred_window, device):
        """numpy inputs: tt [R] f32, vals/mask [R,C] f32, row_off [W+1] i64, note_tau [S] f32, note_src [S] i64,
        note_off [W+1] i64, emb [E,d_m] f32 (the concatenated per-entity embedding matrices)."""
        self.history, self.pred_window = float(history), float(pred_window)
        self.time_max = np.float32(history + pred_window)
        self.W = len(row_off) - 1
        self.C = int(vals.shape[1])
        self.d_m = int(emb.shape[1]) if emb is not None and emb.ndim == 2 else 0
        self.device = torch.device(device)
        # ---- host metadata
        self._tt, self._mask, self._row_off = tt, mask, row_off.astype(np.int64)
        seg = np.repeat(np.arange(self.W), np.diff(self._row_off))
        is_hist = tt < np.float32(history)
        self.hist_len = np.bincount(seg[is_hist], minlength=self.W).astype(np.int32)
        self.pred_len = (np.diff(self._row_off) - self.hist_len).astype(np.int32)
        # history rows must be a prefix of every window: times non-decreasing inside a window (the dataset sorts by time;
        # equal timestamps -- duplicate date_time rows -- are fine, the reference only splits on tt < history,
        # lib/parse_datasets.py:252-295).  One vectorised check over all rows.
        if len(tt) > 1:
            drop = np.diff(tt) < 0
            b = self._row_off[1:-1]                        # the step from one window's last row to the next window's first
            b = b[(b > 0) & (b < len(tt))]                 # (empty windows at either end have no such step)
            drop[b - 1] = False
            if np.any(drop):
                w = int(np.searchsorted(self._row_off, int(np.argmax(drop)) + 1, side="right") - 1)
                raise ValueError(f"window {w}: timestamps must be non-decreasing")
        self.n_notes = np.diff(note_off).astype(np.int32)
        self._patch_cache = {}
        # ---- device arrays
        dev = self.device
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(dev)    # noqa: E731
        self.d = dict(tt=t(tt, np.float32), vals=t(vals, np.float32), mask=t(mask, np.float32), row_off=t(row_off, np.int64),
                      hist_len=t(self.hist_len, np.int32), note_tau=t(note_tau, np.float32), note_src=t(note_src, np.int64),
                      note_off=t(note_off, np.int64), emb=t(emb, np.float32) if self.d_m else None)
        self._struct = Store(*[None if self.d[k] is None else self.d[k].data_ptr() for k in
                               ("tt", "vals", "mask", "row_off", "hist_len", "note_tau", "note_src", "note_off", "emb")],
                             self.C, self.d_m)

    # ------------------------------------------------------------------------------------------ construction
    @classmethod
    def from_chunks(cls, chunks, history, pred_window, device):
        """chunks: the reference dataset's list of (chunk_id, tt, vals, mask, texts) with texts = [(t, embedding)],
        exactly `ChunkedTimeSeriesDataset.chunks` (lib/parse_datasets.py:150-236).  Embedding tensors that are rows of
        one matrix (they are: `emb[i]`, :129-131) are stored once."""
        tt, vals, mask, row_off = [], [], [], [0]
        note_tau, note_src, note_off = [], [], [0]
        rows, emb_rows = {}, []
        for _, t, v, m, texts in chunks:
            tt.append(np.asarray(t, dtype=np.float32))
            vals.append(np.asarray(v, dtype=np.float32))
            mask.append(np.asarray(m, dtype=np.float32))
            row_off.append(row_off[-1] + len(tt[-1]))
            for (tn, payload) in texts:
                key = (payload.untyped_storage().data_ptr(), payload.storage_offset()) if torch.is_tensor(payload) \
                    else id(payload)
                if key not in rows:
                    rows[key] = len(emb_rows)
                    emb_rows.append(np.asarray(payload, dtype=np.float32))
                note_tau.append(np.float32(tn))
                note_src.append(rows[key])
            note_off.append(note_off[-1] + len(texts))
        emb = np.stack(emb_rows) if emb_rows else np.zeros((0, 0), np.float32)
        return cls(np.concatenate(tt), np.concatenate(vals), np.concatenate(mask), np.array(row_off, np.int64),
                   np.array(note_tau, np.float32), np.array(note_src, np.int64), np.array(note_off, np.int64), emb,
                   history, pred_window, device)

    @classmethod
    def from_dataset_dir(cls, root, history, pred_window, stride, device, time_unit="days", llm_model_fusion="GPT2",
                         llm_layers_fusion=None, max_length=1024, rec_ids=None):
        """Build the store from the reference's on-disk layout:
            <root>/processed/<entity>/time_series.csv                       date_time, record_id, <features...>
            <root>/processed/<entity>/text_embeddings_model={llm}_layers={n|full}_maxlen={L}.pt
                                                                            {embeddings [n,d_m] f32, rel_times [n] f32}
        and window it exactly like ChunkedTimeSeriesDataset (lib/parse_datasets.py:17-245): per-entity z-scored
        features (pandas mean / std, as there), times in `time_unit` from the entity's first observation, windows
        [st, st + history + pred_window) stepped by `stride` that hold >= 2 observations, an observed value on both
        sides of `history` and at least one note in [st, st + history).  The windows are found with searchsorted over
        the sorted times instead of a scan per window; each entity's embedding matrix goes to the device once.
        Returns (store, window_ids) with the reference's chunk ids ("<entity>_chunk<k>")."""
        import os
        import pandas as pd
        unit = {"seconds": 1.0, "minutes": 60.0, "hours": 3600.0, "days": 86400.0, "weeks": 604800.0}[time_unit]
        proc = os.path.join(root, "processed")
        ents = sorted(d for d in os.listdir(proc) if os.path.isdir(os.path.join(proc, d))) if rec_ids is None else list(rec_ids)
        total = history + pred_window
        tt_all, vals_all, mask_all, row_off, ids = [], [], [], [0], []
        tau_all, src_all, note_off, embs, emb_base = [], [], [0], [], 0
        for ent in ents:
            ts_path = os.path.join(proc, ent, "time_series.csv")
            if not os.path.isfile(ts_path):
                continue
            df = pd.read_csv(ts_path)
            when = pd.to_datetime(df["date_time"])
            df = df.assign(_when=when).sort_values("_when")
            feats = [c for c in df.columns if c not in ("date_time", "record_id", "_when")]
            x = df[feats]
            mu, sd = x.mean(), x.std()
            x = (x - mu) / sd.where(sd != 0, 1.0)              # z-score; constant columns are only centred
            secs = (df["_when"] - df["_when"].min()).dt.total_seconds()
            tt = (secs / unit).values.astype(np.float32)
            v32 = x.values.astype(np.float32)
            mask = (~np.isnan(v32)).astype(np.float32)
            vals = np.nan_to_num(v32, nan=0.0, posinf=np.finfo(np.float32).max, neginf=np.finfo(np.float32).min)
            if mask.sum() == 0:
                raise ValueError(f"Mask for {ent} is all zeros")
            fname = f"text_embeddings_model={llm_model_fusion}_layers={llm_layers_fusion or 'full'}_maxlen={max_length}.pt"
            path = os.path.join(proc, ent, fname)
            if not os.path.isfile(path):
                raise FileNotFoundError(f"Missing text embeddings file: {path}")
            blob = torch.load(path, map_location="cpu")
            emb = blob["embeddings"].float().numpy()
            if np.isnan(emb).any():
                raise ValueError("text embeddings contains NaN values.")
            rel = blob["rel_times"].float().numpy().astype(np.float64)      # what `.item()` of the fp32 times yields
            embs.append(emb)
            # ---- windows: st_k = t_min + k * stride while st_k + total <= t_max
            t_min, t_max = float(tt.min()), float(tt.max())
            k, cnt = 0, 0
            order_ok = np.all(np.diff(tt) >= 0)
            while t_min + k * stride + total <= t_max:
                st = t_min + k * stride
                k += 1
                st32 = np.float32(st)
                if order_ok:
                    a, b = np.searchsorted(tt, st32, "left"), np.searchsorted(tt, np.float32(st + total), "left")
                    sel = np.arange(a, b)
                else:
                    sel = np.flatnonzero((tt >= st32) & (tt < np.float32(st + total)))
                if len(sel) < 2:
                    continue
                sub_tt = tt[sel] - st32
                hist = sub_tt < np.float32(history)
                if mask[sel][hist].sum() == 0 or mask[sel][~hist].sum() == 0:
                    continue
                notes = np.flatnonzero((rel >= st) & (rel < st + history))
                this = cnt
                cnt += 1
                if len(notes) == 0:          # windows without text are dropped, but they still consume a chunk number
                    continue
                ids.append(f"{ent}_chunk{this}")
                tt_all.append(sub_tt)
                vals_all.append(vals[sel])
                mask_all.append(mask[sel])
                row_off.append(row_off[-1] + len(sel))
                tau_all.append((rel[notes] - st).astype(np.float32))
                src_all.append(notes.astype(np.int64) + emb_base)
                note_off.append(note_off[-1] + len(notes))
            emb_base += len(emb)
        if not ids:
            raise RuntimeError("No chunks created; check history/pred_window/stride")
        store = cls(np.concatenate(tt_all), np.concatenate(vals_all), np.concatenate(mask_all), np.array(row_off, np.int64),
                    np.concatenate(tau_all), np.concatenate(src_all), np.array(note_off, np.int64), np.concatenate(embs),
                    history, pred_window, device)
        return store, ids

    # ------------------------------------------------------------------------------------------ batches
    def _ids(self, window_ids):
        ids = np.asarray(window_ids, dtype=np.int32)
        if ids.ndim != 1 or (len(ids) and (ids.min() < 0 or ids.max() >= self.W)):
            raise IndexError("window ids out of range")
        return ids, torch.from_numpy(ids).to(self.device, non_blocking=True)

    def _patch_max(self, patch_size, npatch, patch_stride):
        """per window: max over (patch, variable) of the number of observed history points (host, cached)"""
        key = (patch_size, npatch, patch_stride)
        if key not in self._patch_cache:
            out = np.zeros(self.W, dtype=np.int32)
            for w in range(self.W):
                a = self._row_off[w]
                t, m = self._tt[a:a + self.hist_len[w]], self._mask[a:a + self.hist_len[w]]
                for i in range(npatch):
                    st = np.float32(i * patch_stride)
                    ed = np.float32(self.history if i == npatch - 1 else i * patch_stride + patch_size)
                    sel = (t >= st) & (t < ed)
                    if sel.any():
                        out[w] = max(out[w], int((m[sel] != 0).sum(0).max()))
            self._patch_cache[key] = out
        return self._patch_cache[key]

    def collate(self, window_ids, patch=None, padded_notes=True):
        """-> dict of device tensors.  patch=None: the standard collate's keys; patch=(patch_size, npatch, patch_stride):
        tPatchGNN's.  Always: tau, note_lengths, note_offsets, note_rowmap (+ notes_embeddings unless padded_notes=False)."""
        lib = _lib.load()
        ids, ids_dev = self._ids(window_ids)
        B, dev, f32 = len(ids), self.device, torch.float32
        Lmax = int(self.hist_len[ids].max()) if B else 0
        Lpmax = int(self.pred_len[ids].max()) if B else 0
        Nmax = int(self.n_notes[ids].max()) if B else 0
        out = {"tp_to_predict": torch.empty(B, Lpmax, dtype=f32, device=dev),
               "data_to_predict": torch.empty(B, Lpmax, self.C, dtype=f32, device=dev),
               "mask_predicted_data": torch.empty(B, Lpmax, self.C, dtype=f32, device=dev)}
        st = C.byref(self._struct)
        if B == 0:          # nothing to launch (empty tensors have no device pointer)
            shp = (0, 0, self.C) if patch is None else (0, patch[1], 0, self.C)
            out.update(observed_tp=torch.empty(shp[:-1], dtype=f32, device=dev), observed_data=torch.empty(shp, dtype=f32, device=dev),
                       observed_mask=torch.empty(shp, dtype=f32, device=dev), tau=torch.empty(0, 0, dtype=f32, device=dev),
                       note_lengths=torch.empty(0, dtype=torch.int32, device=dev),
                       note_offsets=torch.zeros(1, dtype=torch.int32, device=dev),
                       note_rowmap=torch.empty(0, dtype=torch.int32, device=dev))
            if padded_notes and self.d_m:
                out["notes_embeddings"] = torch.empty(0, 0, self.d_m, dtype=f32, device=dev)
            return out
        if patch is None:
            out.update(observed_tp=torch.empty(B, Lmax, dtype=f32, device=dev),
                       observed_data=torch.empty(B, Lmax, self.C, dtype=f32, device=dev),
                       observed_mask=torch.empty(B, Lmax, self.C, dtype=f32, device=dev))
            check(lib.immtsf_collate_series(st, ptr(ids_dev), B, Lmax, Lpmax, float(self.time_max), ptr(out["observed_tp"]),
                                            ptr(out["observed_data"]), ptr(out["observed_mask"]), ptr(out["tp_to_predict"]),
                                            ptr(out["data_to_predict"]), ptr(out["mask_predicted_data"]), stream_ptr()),
                  "collate_series")
        else:
            ps, npatch, pstride = patch
            Lp = int(self._patch_max(ps, npatch, pstride)[ids].max()) if B else 0
            for k in ("observed_tp", "observed_data", "observed_mask"):
                out[k] = torch.empty(B, npatch, Lp, self.C, dtype=f32, device=dev)
            check(lib.immtsf_collate_series(st, ptr(ids_dev), B, 0, Lpmax, float(self.time_max), None, None, None,
                                            ptr(out["tp_to_predict"]), ptr(out["data_to_predict"]),
                                            ptr(out["mask_predicted_data"]), stream_ptr()), "collate_series")
            check(lib.immtsf_collate_patches(st, ptr(ids_dev), B, npatch, float(ps), float(pstride), self.history, Lp,
                                             float(self.time_max), ptr(out["observed_tp"]), ptr(out["observed_data"]),
                                             ptr(out["observed_mask"]), stream_ptr()), "collate_patches")
        total = int(self.n_notes[ids].sum()) if B else 0
        out["tau"] = torch.empty(B, Nmax, dtype=f32, device=dev)
        out["note_lengths"] = torch.empty(B, dtype=torch.int32, device=dev)
        out["note_offsets"] = torch.empty(B + 1, dtype=torch.int32, device=dev)
        out["note_rowmap"] = torch.empty(total, dtype=torch.int32, device=dev)
        notes = None
        if padded_notes and self.d_m:
            notes = out["notes_embeddings"] = torch.empty(B, Nmax, self.d_m, dtype=f32, device=dev)
        check(lib.immtsf_collate_notes(st, ptr(ids_dev), B, Nmax, ptr(out["tau"]), ptr(notes), ptr(out["note_lengths"]),
                                       ptr(out["note_offsets"]), ptr(out["note_rowmap"]), stream_ptr()), "collate_notes")
        if self.d_m:
            from .ops import PackedNotes
            out["notes_packed"] = PackedNotes(self.d["emb"], out["note_rowmap"], out["note_lengths"], Nmax)
        return out
