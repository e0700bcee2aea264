#!/usr/bin/env python3
"""Host cost of one hipGraph launch of the whole-step graph (GraphedStep, two-stream capture) with the GPU idle:
time.perf_counter around graph_a.replay() after a device sync."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "imm-tsf_amd")]
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    from fusions.FusionModel import FusionModel
    from immtsf import _lib, config
    from immtsf.ops import masked_mse
    from immtsf.train import FlatTrainer, GraphedStep
    from lib.evaluation import forecast_and_fuse
    from models.tPatchGNN import tPatchGNN
    _lib.load()
    dev = torch.device("cuda", 0)
    config.precision = "bf16"
    config.nan_check = "deferred"
    config.manual_seed(1)
    torch.manual_seed(0)
    a = bench.model_args(str(dev))
    model = tPatchGNN(a).to(dev).train()
    fusion = FusionModel(a).to(dev).train()
    te = [model.te_scale.weight, model.te_scale.bias, model.te_periodic.weight, model.te_periodic.bias]
    trainer = FlatTrainer([list(fusion.mmf.parameters()), list(fusion.ttf.parameters()), list(model.parameters())],
                          lr=1e-3, weight_decay=0.0, max_norm=1.0, group=None, sink_buckets=(0, 1, 2), sink_shared=te, overlap=False,
                          device_step=True)
    cpu_batch, _ = bench.synth_batch(100, bench.B_PER_GPU)
    b = {k: v.to(dev) for k, v in cpu_batch.items()}
    cnt = b["mask_predicted_data"].reshape(-1, bench.C).sum(0)
    for label, stream in (("two streams", torch.cuda.Stream(device=dev)), ("one stream", None)):
        def loss_fn():
            return masked_mse(forecast_and_fuse(model, fusion, b, stream), b["data_to_predict"], b["mask_predicted_data"], None, cnt)
        st = GraphedStep(trainer, loss_fn)
        for _ in range(10):
            st()
        host, dev_t = [], []
        for _ in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st.graph_a.replay()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append((t1 - t0) * 1e6)
            dev_t.append((t2 - t0) * 1e6)
        host.sort(); dev_t.sort()
        print(f"{label}: graph A launch, host {host[len(host)//2]:.0f} us (min {host[0]:.0f}); launch + completion {dev_t[len(dev_t)//2]:.0f} us")


main()
