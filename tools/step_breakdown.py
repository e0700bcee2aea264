#!/usr/bin/env python3
"""Where does the cfg2 training step spend its time?  Captures each segment of the step (backbone, TTF, MMF, loss,
optimizer; forward and forward+backward) in its own hipGraph and times replays.  Diagnostic only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "imm-tsf_amd")]
import torch  # noqa: E402
import bench  # noqa: E402


def timed_graph(fn, reps=30):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    from fusions.FusionModel import FusionModel
    from immtsf import _lib, config
    from immtsf.ops import masked_mse
    from immtsf.train import FlatTrainer
    from models.tPatchGNN import tPatchGNN
    _lib.load()
    dev = torch.device("cuda", 0)
    config.precision = os.environ.get("IMMTSF_PRECISION", "bf16")
    config.nan_check = "deferred"
    config.manual_seed(1)
    torch.manual_seed(0)
    a = bench.model_args(str(dev))
    model = tPatchGNN(a).to(dev).train()
    fusion = FusionModel(a).to(dev).train()
    trainer = FlatTrainer([list(fusion.mmf.parameters()), list(fusion.ttf.parameters()), list(model.parameters())],
                          lr=1e-3, weight_decay=0.0, max_norm=1.0, group=None, sink_buckets=(0, 1, 2), sink_shared=[model.te_scale.weight, model.te_scale.bias, model.te_periodic.weight, model.te_periodic.bias], overlap=False,
                          device_step=True)
    cpu_batch, _ = bench.synth_batch(100, bench.B_PER_GPU)
    b = {k: v.to(dev) for k, v in cpu_batch.items()}
    cnt = b["mask_predicted_data"].reshape(-1, bench.C).sum(0)
    fc = (b["tp_to_predict"], b["observed_data"], b["observed_tp"], b["observed_mask"])
    pred0 = model.forecasting(*fc).detach()
    E0, M0 = fusion.ttf(b["notes_embeddings"], b["tau"], b["tp_to_predict"])
    E0 = E0.detach()

    def bb_f():
        return model.forecasting(*fc)

    def bb_fb():
        trainer.zero_grad()
        bb_f().square().mean().backward()

    def ttf_f():
        return fusion.ttf(b["notes_embeddings"], b["tau"], b["tp_to_predict"])[0]

    def ttf_fb():
        trainer.zero_grad()
        ttf_f().backward(E0)

    def mmf_f():
        return fusion.mmf(pred0, E0, M0)

    pr, Er = pred0.clone().requires_grad_(True), E0.clone().requires_grad_(True)

    def mmf_fb():
        trainer.zero_grad()
        masked_mse(fusion.mmf(pr, Er, M0), b["data_to_predict"], b["mask_predicted_data"], None, cnt).backward()

    def whole():
        trainer.zero_grad()
        out = fusion(b["notes_embeddings"], b["tau"], b["tp_to_predict"], model.forecasting(*fc))
        masked_mse(out, b["data_to_predict"], b["mask_predicted_data"], None, cnt).backward()

    rows = [("backbone fwd", bb_f), ("backbone fwd+bwd", bb_fb), ("TTF fwd", ttf_f), ("TTF fwd+bwd", ttf_fb),
            ("MMF fwd", mmf_f), ("MMF+loss fwd+bwd", mmf_fb), ("zero_grad", trainer.zero_grad),
            ("clip+Adam", trainer.step), ("whole fwd+bwd (1 stream)", whole)]
    if len(sys.argv) > 1:       # eager run of ONE segment, for a rocprofv3 --kernel-trace of just that segment
        fn = dict(rows)[sys.argv[1]]
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        return
    for name, fn in rows:
        print(f"{name:28s} {timed_graph(fn):9.1f} us")


if __name__ == "__main__":
    main()
