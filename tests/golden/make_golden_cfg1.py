#!/usr/bin/env python3
"""cfg1 end to end from the REAL reference (BASELINE.json configs[0]): DLinear + TTF_RecAvg + MMF_GR_Add
on a synthetic 8-entity dataset in the reference's on-disk layout, batch_size 4.

Run in the build container only (needs /root/reference):

    cd /tmp && python /root/repo/tests/golden/make_golden_cfg1.py

Writes the dataset (8 entities, C = 5, 300-500 irregular observations over 20 days with 30 % NaN, 60-120 notes with
16-wide embeddings; history = pred_window = stride = 24 h), runs the unmodified `lib.parse_datasets.parse_datasets`, builds
the reference's DLinear and FusionModel (dropout 0, d_txt 64), and executes the first five iterations of
`main.trainable`'s training loop (main.py:1057-1104: zero_grad -> compute_all_losses -> backward ->
clip_grad_norm_(1.0) -> Adam) on the shuffled train loader.  Recorded: the dataset bytes, the initial state_dicts, the
window (chunk) ids of each of the five batches, the five losses and a fingerprint of the final parameters.  Tensors and
data only -- no reference source.
"""
import argparse
import os
import sys
import tempfile

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import D_MODEL_TABLE, _install_shims  # noqa: E402


def write_dataset(root, seed=11, entities=8, C=5, d_m=16):
    rng = np.random.default_rng(seed)
    files = {}
    for e in range(entities):
        d = os.path.join(root, "SYN", "processed", f"ent{e:02d}")
        os.makedirs(d)
        n_obs = int(rng.integers(300, 501))
        secs = np.sort(rng.choice(np.arange(0, 20 * 86400, 300), size=n_obs, replace=False))
        t0 = pd.Timestamp("2021-03-01 00:00:00")
        vals = rng.normal(size=(n_obs, C)).astype(np.float64) * (1 + 0.3 * e) + 0.5 * e
        vals[rng.random((n_obs, C)) < 0.3] = np.nan
        vals[0, :] = rng.normal(size=C)
        df = pd.DataFrame({"date_time": [str(t0 + pd.Timedelta(seconds=int(s))) for s in secs], "record_id": f"ent{e:02d}"})
        for c in range(C):
            df[f"f{c}"] = vals[:, c]
        ts_path = os.path.join(d, "time_series.csv")
        df.to_csv(ts_path, index=False)
        n_notes = int(rng.integers(60, 121))
        rel = np.sort(rng.random(n_notes) * 20 * 24).astype(np.float32)
        emb = rng.normal(size=(n_notes, d_m)).astype(np.float32)
        torch.save({"embeddings": torch.from_numpy(emb), "rel_times": torch.from_numpy(rel)},
                   os.path.join(d, "text_embeddings_model=TOY16_layers=full_maxlen=1024.pt"))
        files[f"ent{e:02d}/time_series.csv"] = open(ts_path, "rb").read()
        files[f"ent{e:02d}/emb"] = emb
        files[f"ent{e:02d}/rel"] = rel
    return files


def main():
    _install_shims()
    import prettytable
    prettytable.PrettyTable = type("PrettyTable", (), {"__init__": lambda s, *a, **k: None, "add_row": lambda s, *a: None,
                                                       "__str__": lambda s: ""})
    import importlib
    import fusions.load_llm as ll
    ll.get_d_model = lambda name: D_MODEL_TABLE[name]
    for n in ("fusions.TTF_RecAvg", "fusions.TTF_T2V_XAttn"):
        importlib.import_module(n).get_d_model = ll.get_d_model
    import lib.parse_datasets as pdmod
    from fusions.FusionModel import FusionModel
    from lib.evaluation import compute_all_losses
    from models.DLinear import DLinear
    root = tempfile.mkdtemp(prefix="immtsf_cfg1_")
    files = write_dataset(root)
    args = argparse.Namespace(
        data_root=root, dataset="SYN", device=torch.device("cpu"), history=24, pred_window=24, stride=24, time_unit="hours",
        enable_text=True, use_text_embeddings=True, llm_model_fusion="TOY16", llm_layers_fusion=None, max_length=1024,
        split_method="sample", model="DLinear", batch_size=4, patch_size=8, npatch=3, patch_stride=8, rec_ids=None,
        TTF_module="TTF_RecAvg", MMF_module="MMF_GR_Add", recency_sigma=1.0, n_heads_fusion=1, dropout=0.0, d_txt=64, kappa=0.5,
        moving_avg=25, individual=False, lr=1e-3, w_decay=0.0)
    torch.manual_seed(1234)
    data_obj = pdmod.parse_datasets(args, show_summary=False)
    args.C = data_obj["input_dim"]
    args.enc_in = args.c_out = args.C
    mx = lambda key: max(int(b[key].shape[1]) for b in data_obj["train_dataloader"])    # noqa: E731
    try:
        import main as ref_main
        args.input_len, args.pred_len = ref_main.get_input_and_pred_len(data_obj)
    except Exception:
        args.input_len, args.pred_len = mx("observed_tp"), mx("tp_to_predict")
    torch.manual_seed(99)
    model = DLinear(args)
    fusion = FusionModel(args)
    out = {"input_len": np.int64(args.input_len), "pred_len": np.int64(args.pred_len), "C": np.int64(args.C)}
    for k, v in model.state_dict().items():
        out["model." + k] = v.detach().numpy().copy()
    for k, v in fusion.state_dict().items():
        out["fusion." + k] = v.detach().numpy().copy()
    # window ids of the train batches: record the chunk ids the collate function is handed
    ds = data_obj["ds"]
    index_of = {c[0]: i for i, c in enumerate(ds.chunks)}
    loader = data_obj["train_dataloader"]
    seen = []
    orig = loader.collate_fn

    def spy(batch):
        seen.append([index_of[b[0]] for b in batch])
        return orig(batch)
    loader.collate_fn = spy
    params = list(model.parameters()) + list(fusion.parameters())
    opt = torch.optim.Adam(params, lr=args.lr, weight_decay=args.w_decay)
    model.train()
    fusion.train()
    losses = []
    torch.manual_seed(7)            # the loader's shuffle
    for step, batch in enumerate(loader):
        if step == 5:
            break
        opt.zero_grad()
        res = compute_all_losses(model, fusion, batch, True)
        res["loss"].backward()
        torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
        opt.step()
        losses.append(float(res["loss"].item()))
    for i in range(5):
        out[f"step{i}.window_ids"] = np.array(seen[i], dtype=np.int64)
    out["losses"] = np.array(losses, dtype=np.float64)
    out["final_norm"] = np.float64(np.sqrt(sum(float((p.detach().double() ** 2).sum()) for p in params)))
    out["n_windows"] = np.int64(len(ds.chunks))
    for k, v in files.items():
        out["file." + k] = np.frombuffer(v, dtype=np.uint8) if isinstance(v, bytes) else v
    np.savez_compressed(os.path.join(HERE, "cfg1_e2e.npz"), **out)
    print("cfg1: windows", len(ds.chunks), "input_len", args.input_len, "pred_len", args.pred_len, "losses", losses)


if __name__ == "__main__":
    main()
