"""The numpy restatement of the reference collate (oracle/collate_ref.py) against batches produced by the REAL
reference's loaders (tests/golden/collate_*.npz): every tensor bit-exact."""
import os

import numpy as np
import pytest

from oracle import collate_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    zs = np.load(os.path.join(GOLDEN, "collate_standard.npz"))
    emb = {int(k[8:10]): zs[k] for k in zs.files if k.startswith("file.ent") and k.endswith("/emb")}
    chunks = []
    for c in R.chunks_from_golden(z):
        ne = np.stack([emb[int(e)][int(r)] for e, r in zip(c["note_ent"], c["note_row"])]) if len(c["note_row"]) \
            else np.zeros((0, 16), np.float32)
        chunks.append((c["tt"], c["vals"], c["mask"], c["note_t"], ne))
    return z, chunks


def same(a, b):
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)


@pytest.mark.parametrize("name", ["collate_standard", "collate_patch"])
def test_oracle_matches_reference_batches(name):
    z, chunks = load(name)
    hist, tmax = float(z["history"]), float(z["history"] + z["pred_window"])
    ps, npatch, pstride = [int(v) for v in z["patch"]]
    assert int(z["n_batches"]) >= 4
    for b in range(int(z["n_batches"])):
        sel = [chunks[i] for i in z[f"b{b}.window_ids"]]
        got = R.series_collate(sel, hist, tmax) if name == "collate_standard" else \
            R.patch_collate(sel, hist, tmax, ps, npatch, pstride)
        got.update({k: v for k, v in R.notes_collate(sel).items() if k in ("tau", "notes_embeddings")})
        keys = [k[len(f"b{b}."):] for k in z.files if k.startswith(f"b{b}.") and not k.endswith("window_ids")]
        assert sorted(keys) == sorted(got.keys())
        for k in keys:
            assert same(got[k], z[f"b{b}.{k}"]), (name, b, k)


def test_ragged_index_of_notes():
    _, chunks = load("collate_standard")
    out = R.notes_collate(chunks[:5])
    assert out["lengths"].dtype == np.int32 and out["offsets"][-1] == out["lengths"].sum()
    # the padded tensor's non-zero rows are exactly the first lengths[b] rows (what the fusion's note_mask re-derives)
    nz = (np.abs(out["notes_embeddings"]).sum(-1) > 0).sum(1)
    assert np.array_equal(nz, out["lengths"])
