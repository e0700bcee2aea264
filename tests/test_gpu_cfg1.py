"""cfg1 (BASELINE.json configs[0]) end to end on the GPU: dataset files on disk -> ResidentStore -> device collate ->
DLinear + TTF_RecAvg + MMF_GR_Add -> lib.evaluation.compute_all_losses -> backward -> clip_grad_norm_(1.0) -> Adam,
against the first five training-step losses of the REAL reference's main.trainable loop on the same files, the same
initial weights and the same batches (tests/golden/make_golden_cfg1.py; dropout 0, fp32): 1e-4."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def test_cfg1_disk_to_five_training_steps(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    from fusions.FusionModel import FusionModel
    from fusions.load_llm import register_d_model
    from immtsf import config
    from immtsf.data import ResidentStore
    from lib.evaluation import compute_all_losses
    from models.DLinear import DLinear
    z = np.load(os.path.join(GOLDEN, "cfg1_e2e.npz"))
    for e in sorted({k.split("/")[0][5:] for k in z.files if k.startswith("file.")}):
        d = tmp_path / "SYN" / "processed" / e
        d.mkdir(parents=True)
        (d / "time_series.csv").write_bytes(z[f"file.{e}/time_series.csv"].tobytes())
        torch.save({"embeddings": torch.from_numpy(z[f"file.{e}/emb"]), "rel_times": torch.from_numpy(z[f"file.{e}/rel"])},
                   str(d / "text_embeddings_model=TOY16_layers=full_maxlen=1024.pt"))
    store, ids = ResidentStore.from_dataset_dir(str(tmp_path / "SYN"), 24, 24, 24, dev, time_unit="hours", llm_model_fusion="TOY16")
    assert len(ids) == int(z["n_windows"])
    register_d_model("TOY16", 16)
    config.precision = "fp32"
    old_nan = config.nan_check
    config.nan_check = "sync"            # the reference's guards, as main.py runs them
    C = int(z["C"])
    a = types.SimpleNamespace(
        device=str(dev), C=C, enc_in=C, c_out=C, input_len=int(z["input_len"]), pred_len=int(z["pred_len"]), moving_avg=25,
        individual=False, batch_size=4, TTF_module="TTF_RecAvg", MMF_module="MMF_GR_Add", llm_model_fusion="TOY16",
        llm_layers_fusion=None, max_length=1024, use_text_embeddings=True, recency_sigma=1.0, n_heads_fusion=1, dropout=0.0,
        d_txt=64, kappa=0.5)
    try:
        model, fusion = DLinear(a).to(dev).train(), FusionModel(a).to(dev).train()
        model.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("model.")}, strict=True)
        fusion.load_state_dict({k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("fusion.")}, strict=True)
        params = list(model.parameters()) + list(fusion.parameters())
        opt = torch.optim.Adam(params, lr=1e-3, weight_decay=0.0)
        losses = []
        for i in range(5):
            batch = store.collate(z[f"step{i}.window_ids"])
            opt.zero_grad()
            res = compute_all_losses(model, fusion, batch, True)
            res["loss"].backward()
            torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
            opt.step()
            losses.append(float(res["loss"].item()))
        torch.cuda.synchronize()
    finally:
        config.nan_check = old_nan
    want = z["losses"]
    err = np.abs(np.array(losses) - want) / np.abs(want)
    assert err.max() < 1e-4, (losses, want.tolist())
    norm = float(np.sqrt(sum(float((p.detach().double() ** 2).sum()) for p in params)))
    assert abs(norm / float(z["final_norm"]) - 1.0) < 1e-5
