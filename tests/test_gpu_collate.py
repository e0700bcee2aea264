"""GPU parity of the device-side batch builder: ResidentStore.collate against the batches the
REAL reference's loaders produced (tests/golden/collate_*.npz) -- every tensor bit-exact -- and the packed ragged
note index against immtsf_ragged_index run on the padded embeddings it replaces."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _chunks(name):
    """the reference dataset's chunk list rebuilt from the fixture: (id, tt, vals, mask, [(t, emb_row_view)])"""
    from oracle import collate_ref as R
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    zs = np.load(os.path.join(GOLDEN, "collate_standard.npz"))
    emb = {int(k[8:10]): torch.from_numpy(zs[k]) for k in zs.files if k.startswith("file.ent") and k.endswith("/emb")}
    chunks = []
    for c in R.chunks_from_golden(z):
        texts = [(float(t), emb[int(e)][int(r)]) for t, e, r in zip(c["note_t"], c["note_ent"], c["note_row"])]
        chunks.append((c["id"], torch.from_numpy(c["tt"]), torch.from_numpy(c["vals"]), torch.from_numpy(c["mask"]), texts))
    return z, chunks


@pytest.mark.parametrize("name", ["collate_standard", "collate_patch"])
def test_collate_bit_exact_vs_reference_batches(name):
    dev = _dev()
    from immtsf.data import ResidentStore
    z, chunks = _chunks(name)
    store = ResidentStore.from_chunks(chunks, float(z["history"]), float(z["pred_window"]), dev)
    twice = ResidentStore.from_chunks(chunks + chunks[:9], float(z["history"]), float(z["pred_window"]), dev)
    assert twice.d["emb"].shape[0] == store.d["emb"].shape[0]            # rows of one embedding matrix are stored once
    patch = None if name == "collate_standard" else tuple(int(v) for v in z["patch"])
    for b in range(int(z["n_batches"])):
        got = store.collate(z[f"b{b}.window_ids"], patch=patch)
        keys = [k[len(f"b{b}."):] for k in z.files if k.startswith(f"b{b}.") and not k.endswith("window_ids")]
        for k in keys:
            exp = z[f"b{b}.{k}"]
            g = got[k].cpu().numpy()
            assert g.shape == exp.shape and g.dtype == exp.dtype and np.array_equal(g, exp), (name, b, k)


def test_packed_note_index_matches_ragged_index_of_padded_tensor():
    dev = _dev()
    from immtsf import _lib
    from immtsf.data import ResidentStore
    lib = _lib.load()
    z, chunks = _chunks("collate_standard")
    store = ResidentStore.from_chunks(chunks, float(z["history"]), float(z["pred_window"]), dev)
    ids = np.array([0, 7, 13, 21, 34, 58, 2, 2], dtype=np.int64)
    got = store.collate(ids)
    notes = got["notes_embeddings"]
    B, N, d_m = notes.shape
    mask = torch.zeros(B * N, dtype=torch.uint8, device=dev)
    lengths = torch.zeros(B, dtype=torch.int32, device=dev)
    offsets = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    rowmap = torch.full((B * N,), -1, dtype=torch.int32, device=dev)
    seg = torch.full((B * N,), -1, dtype=torch.int32, device=dev)
    mtxt = torch.zeros(B, dtype=torch.uint8, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.immtsf_ragged_index(_lib.ptr(notes), B, N, d_m, _lib.ptr(mask), _lib.ptr(lengths), _lib.ptr(offsets),
                                       _lib.ptr(rowmap), _lib.ptr(seg), _lib.ptr(mtxt), _lib.ptr(flag), _lib.stream_ptr()),
               "ragged_index")
    assert torch.equal(lengths, got["note_lengths"]) and torch.equal(offsets, got["note_offsets"])
    tot = int(offsets[-1])
    packed_from_padded = notes.reshape(B * N, d_m)[rowmap[:tot].long()]
    packed_from_store = store.d["emb"][got["note_rowmap"]]
    assert torch.equal(packed_from_padded, packed_from_store)
    # without the padded tensor
    lean = store.collate(ids, padded_notes=False)
    assert "notes_embeddings" not in lean and torch.equal(lean["note_rowmap"], got["note_rowmap"])
    assert torch.equal(lean["tau"], got["tau"])


def test_collate_edge_cases():
    dev = _dev()
    from immtsf.data import ResidentStore
    z, chunks = _chunks("collate_standard")
    store = ResidentStore.from_chunks(chunks, float(z["history"]), float(z["pred_window"]), dev)
    one = store.collate([5])
    assert one["observed_tp"].shape[0] == 1 and one["tau"].shape[1] == int(store.n_notes[5])
    with pytest.raises(IndexError):
        store.collate([len(chunks)])
    empty = store.collate([])
    assert empty["observed_data"].shape == (0, 0, store.C) and empty["note_offsets"].cpu().tolist() == [0]


@pytest.mark.parametrize("d_txt", [16, 8])       # d_txt == d_m: no input projection (gather only); 8: input_proj GEMM
def test_ttf_on_packed_notes_equals_padded(d_txt):
    """TTF_T2V_XAttn fed the packed ragged form (resident matrix + row map from the batch builder) must give the
    same E_txt / M_txt / parameter gradients as on the zero-padded tensor it replaces (same kernels, different
    gather source): tolerance 1e-6 (atomics order in split-K only)."""
    dev = _dev()
    import types
    from fusions.load_llm import register_d_model
    from fusions.TTF_T2V_XAttn import TTF_T2V_XAttn
    from immtsf import config
    from immtsf.data import ResidentStore
    config.precision = "fp32"
    z, chunks = _chunks("collate_standard")
    store = ResidentStore.from_chunks(chunks, float(z["history"]), float(z["pred_window"]), dev)
    batch = store.collate(np.array([3, 11, 12, 40, 41, 57], dtype=np.int64))
    register_d_model("TOY16", 16)
    torch.manual_seed(0)
    ttf = TTF_T2V_XAttn("TOY16", None, 1024, str(dev), True, n_heads_fusion=2, dropout=0.0,
                        d_txt=None if d_txt == 16 else d_txt).to(dev).train()
    t_hat = batch["tp_to_predict"]
    up = torch.randn(t_hat.shape[0], t_hat.shape[1], ttf.d_txt, device=dev)
    res = []
    for notes in (batch["notes_embeddings"], batch["notes_packed"]):
        ttf.zero_grad()
        E, M = ttf(notes, batch["tau"], t_hat)
        (E * up).sum().backward()
        res.append((E.detach().clone(), M.clone(), {n: p.grad.clone() for n, p in ttf.named_parameters() if p.grad is not None}))
    assert torch.equal(res[0][1], res[1][1])
    assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-6 * float(res[0][0].abs().max())
    assert res[0][2].keys() == res[1][2].keys() and len(res[0][2]) >= 12
    for n in res[0][2]:
        a, b = res[0][2][n], res[1][2][n]
        assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-3), n


def test_disk_to_batch_end_to_end(tmp_path):
    """files on disk -> ResidentStore.from_dataset_dir -> device collate == the reference loaders' batches (bit-exact)"""
    dev = _dev()
    from immtsf.data import ResidentStore
    z = np.load(os.path.join(GOLDEN, "collate_patch.npz"))
    zs = np.load(os.path.join(GOLDEN, "collate_standard.npz"))
    for e in sorted({k.split("/")[0][5:] for k in zs.files if k.startswith("file.")}):
        d = tmp_path / "SYN" / "processed" / e
        d.mkdir(parents=True)
        (d / "time_series.csv").write_bytes(zs[f"file.{e}/time_series.csv"].tobytes())
        torch.save({"embeddings": torch.from_numpy(zs[f"file.{e}/emb"]), "rel_times": torch.from_numpy(zs[f"file.{e}/rel"])},
                   str(d / "text_embeddings_model=TOY16_layers=full_maxlen=1024.pt"))
    store, ids = ResidentStore.from_dataset_dir(str(tmp_path / "SYN"), 24, 24, 24, dev, time_unit="hours", llm_model_fusion="TOY16")
    assert len(ids) == int(z["chunks.n"])
    patch = tuple(int(v) for v in z["patch"])
    for b in range(int(z["n_batches"])):
        got = store.collate(z[f"b{b}.window_ids"], patch=patch)
        for k in [k[len(f"b{b}."):] for k in z.files if k.startswith(f"b{b}.") and not k.endswith("window_ids")]:
            assert np.array_equal(got[k].cpu().numpy(), z[f"b{b}.{k}"]), (b, k)
