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


def test_on_disk_loader_matches_reference_chunks(tmp_path):
    """ResidentStore.from_dataset_dir on the synthetic dataset (its bytes are in the fixture) must reproduce the chunk
    list the reference's ChunkedTimeSeriesDataset built from the same files: ids, times, values, masks, note times and
    note rows -- bit-exact."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "imm-tsf_amd"))
    from immtsf.data import ResidentStore
    z = np.load(os.path.join(GOLDEN, "collate_standard.npz"))
    ents = sorted({k.split("/")[0][5:] for k in z.files if k.startswith("file.")})
    base = {}
    for i, e in enumerate(ents):
        d = tmp_path / "SYN" / "processed" / e
        d.mkdir(parents=True)
        (d / "time_series.csv").write_bytes(z[f"file.{e}/time_series.csv"].tobytes())
        torch.save({"embeddings": torch.from_numpy(z[f"file.{e}/emb"]), "rel_times": torch.from_numpy(z[f"file.{e}/rel"])},
                   str(d / "text_embeddings_model=TOY16_layers=full_maxlen=1024.pt"))
        base[int(e[3:])] = sum(len(z[f"file.{x}/emb"]) for x in ents[:i])
    store, ids = ResidentStore.from_dataset_dir(str(tmp_path / "SYN"), 24, 24, 24, "cpu", time_unit="hours",
                                                llm_model_fusion="TOY16")
    assert ids == [str(s) for s in z["chunks.ids"]]
    assert np.array_equal(store.d["row_off"].numpy(), z["chunks.tt_off"])
    for k, g in (("tt", "tt"), ("vals", "vals"), ("mask", "mask")):
        assert np.array_equal(store.d[k].numpy(), z["chunks." + g]), k
    assert np.array_equal(store.d["note_off"].numpy(), z["chunks.note_off"])
    assert np.array_equal(store.d["note_tau"].numpy(), z["chunks.note_t"].astype(np.float32))
    exp_src = np.array([base[int(e)] + int(r) for e, r in zip(z["chunks.note_ent"], z["chunks.note_row"])], dtype=np.int64)
    assert np.array_equal(store.d["note_src"].numpy(), exp_src)
    assert np.array_equal(store.d["emb"].numpy(), np.concatenate([z[f"file.{e}/emb"] for e in ents]))
