"""CPU restatement (numpy) of the reference's batch collate -- TEST INFRASTRUCTURE ONLY (the oracle for the device-side
batch builder).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

Pinned against tests/golden/collate_{standard,patch}.npz, which hold the batches the real reference's loaders
produced from a synthetic on-disk dataset (tests/golden/make_golden_collate.py).

Follows, by behaviour:
  * lib/parse_datasets.py:252-295  variable_time_collate_fn (history / prediction split, zero padding, time normalisation)
  * lib/parse_datasets.py:298-366  patch_variable_time_collate_fn + lib/utils.py:359-413 split_and_patch_batch
  * lib/parse_datasets.py:764-824  multimodal wrapper (tau, zero-padded note embeddings)
  * lib/utils.py:335-347           normalize_masked_tp

A `chunk` here is (tt [L] f32 ascending chunk-relative times, vals [L,C] f32, mask [L,C] f32,
note_t [n] f32 chunk-relative note times, note_emb [n,d_m] f32).
"""
import numpy as np


def normalize_tp(tp, time_max):
    """(tp - 0) / scale in fp32, scale = time_max + (time_max == 0) * 1e-8   (lib/utils.py:335-347)"""
    scale = np.float32(time_max) - np.float32(0.0)
    scale = np.float32(scale + np.float32(scale == 0) * np.float32(1e-8))
    return ((tp.astype(np.float32) - np.float32(0.0)) / scale).astype(np.float32)


def _pad(seqs, width, tail):
    out = np.zeros((len(seqs), width) + tail, dtype=np.float32)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = s
    return out


def series_collate(chunks, history, time_max):
    """standard collate: rows with tt < history are the observed part, the rest the prediction part"""
    obs, pred = [], []
    for tt, vals, mask, _, _ in chunks:
        h = tt < np.float32(history)
        obs.append((tt[h], vals[h], mask[h]))
        pred.append((tt[~h], vals[~h], mask[~h]))
    C = chunks[0][1].shape[1]
    L = max(len(o[0]) for o in obs)
    Lp = max(len(p[0]) for p in pred)
    return {
        "observed_tp": normalize_tp(_pad([o[0] for o in obs], L, ()), time_max),
        "observed_data": _pad([o[1] for o in obs], L, (C,)),
        "observed_mask": _pad([o[2] for o in obs], L, (C,)),
        "tp_to_predict": normalize_tp(_pad([p[0] for p in pred], Lp, ()), time_max),
        "data_to_predict": _pad([p[1] for p in pred], Lp, (C,)),
        "mask_predicted_data": _pad([p[2] for p in pred], Lp, (C,)),
    }


def patch_collate(chunks, history, time_max, patch_size, npatch, patch_stride):
    """tPatchGNN collate: per (window, patch, variable) the observed points whose time falls in the patch's range,
    compacted to the front of a (B, npatch, Lmax, C) tensor (times normalised, pads 0).  Stated per window -- the
    reference goes through the batch-wide union of timestamps, which only serves to order points by time."""
    B, C = len(chunks), chunks[0][1].shape[1]
    per = []                      # per (b, i, d): indices of the observed history rows inside patch i
    Lmax = 0
    for tt, vals, mask, _, _ in chunks:
        h = tt < np.float32(history)
        rows = []
        for i in range(npatch):
            st = i * patch_stride
            ed = history if i == npatch - 1 else st + patch_size
            inp = h & (tt >= np.float32(st)) & (tt < np.float32(ed))
            rows.append([np.flatnonzero(inp & (mask[:, d] != 0)) for d in range(C)])
            Lmax = max([Lmax] + [len(r) for r in rows[-1]])
        per.append(rows)
    out = {k: np.zeros((B, npatch, Lmax, C), dtype=np.float32) for k in ("observed_tp", "observed_data", "observed_mask")}
    for b, (tt, vals, mask, _, _) in enumerate(chunks):
        ntt = normalize_tp(tt, time_max)
        for i in range(npatch):
            for d in range(C):
                r = per[b][i][d]
                out["observed_tp"][b, i, :len(r), d] = ntt[r]
                out["observed_data"][b, i, :len(r), d] = vals[r, d]
                out["observed_mask"][b, i, :len(r), d] = mask[r, d]
    s = series_collate(chunks, history, time_max)
    for k in ("tp_to_predict", "data_to_predict", "mask_predicted_data"):
        out[k] = s[k]
    return out


def notes_collate(chunks):
    """tau (B, Nmax) and zero-padded embeddings (B, Nmax, d_m), plus the packed-ragged index the fusion kernels use:
    lengths[B], offsets[B+1] (int32)"""
    n = [len(c[3]) for c in chunks]
    N = max(n)
    d_m = chunks[0][4].shape[1]
    tau = _pad([c[3] for c in chunks], N, ())
    emb = _pad([c[4] for c in chunks], N, (d_m,))
    lengths = np.array(n, dtype=np.int32)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
    return {"tau": tau, "notes_embeddings": emb, "lengths": lengths, "offsets": offsets}


def chunks_from_golden(z, prefix="chunks."):
    """rebuild the chunk list from a collate_*.npz fixture (embedding rows come from the fixture's per-entity
    matrices `file.entXX/emb` in collate_standard.npz, passed as `z_emb` by the caller when needed)"""
    tt_off, n_off = z[prefix + "tt_off"], z[prefix + "note_off"]
    out = []
    for w in range(int(z[prefix + "n"])):
        a, b = tt_off[w], tt_off[w + 1]
        na, nb = n_off[w], n_off[w + 1]
        out.append(dict(tt=z[prefix + "tt"][a:b], vals=z[prefix + "vals"][a:b], mask=z[prefix + "mask"][a:b],
                        note_t=z[prefix + "note_t"][na:nb], note_row=z[prefix + "note_row"][na:nb],
                        note_ent=z[prefix + "note_ent"][na:nb], id=str(z[prefix + "ids"][w])))
    return out
