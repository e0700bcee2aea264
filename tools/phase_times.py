#!/usr/bin/env python3
"""Where the phases of immtsf.train.PhasedStep start and end on the device (HIP event timestamps, no profiler):
replays the six graphs exactly as PhasedStep.__call__ does, with timing events recorded around every graph launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "imm-tsf_amd")]
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    from fusions.FusionModel import FusionModel
    from immtsf import _lib, config
    from immtsf.ops import masked_mse
    from immtsf.train import FlatTrainer, PhasedStep
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
    trainer = FlatTrainer([list(fusion.mmf.parameters()), list(fusion.ttf.parameters()), list(model.parameters())],
                          lr=1e-3, weight_decay=0.0, max_norm=1.0, group=None, sink_buckets=(0, 1, 2), sink_shared=[model.te_scale.weight, model.te_scale.bias, model.te_periodic.weight, model.te_periodic.bias], overlap=False,
                          device_step=True)
    cpu_batch, _ = bench.synth_batch(100, bench.B_PER_GPU)
    b = {k: v.to(dev) for k, v in cpu_batch.items()}
    cnt = b["mask_predicted_data"].reshape(-1, bench.C).sum(0)
    fc = (b["tp_to_predict"], b["observed_data"], b["observed_tp"], b["observed_mask"])

    def text_fn():
        E, M = fusion.ttf(b["notes_embeddings"], b["tau"], b["tp_to_predict"])
        return (E, M) + tuple(fusion.mmf.project_kv(E))

    def head_fn(pred, E, M, kv, fold):
        return masked_mse(fusion.mmf(pred, E, M, kv=(kv, fold)), b["data_to_predict"], b["mask_predicted_data"], None, cnt)

    st = PhasedStep(trainer, text_fn, lambda: model.forecasting(*fc), head_fn)
    for _ in range(20):
        st()
    torch.cuda.synchronize()
    T, B = st.T, st.B
    ev = lambda: torch.cuda.Event(enable_timing=True)      # noqa: E731
    names = ["start", "T1", "B1s", "B1", "T2s", "T2", "B2s", "B2", "T3s", "T3", "Os", "O"]
    acc = {n: 0.0 for n in names}
    reps = 30
    for _ in range(reps):
        e = {n: ev() for n in names}
        torch.cuda.synchronize()
        with torch.cuda.stream(T):
            e["start"].record(T)
            st.gT1.replay()
            e["T1"].record(T)
        with torch.cuda.stream(B):
            B.wait_event(e["start"])
            e["B1s"].record(B)
            st.gB1.replay()
            e["B1"].record(B)
        with torch.cuda.stream(T):
            T.wait_event(e["B1"])
            e["T2s"].record(T)
            st.gT2.replay()
            e["T2"].record(T)
        with torch.cuda.stream(B):
            B.wait_event(e["T2"])
            e["B2s"].record(B)
            st.gB2.replay()
            e["B2"].record(B)
        with torch.cuda.stream(T):
            if os.environ.get("SERIAL"):          # solo timings: the text-side backward only starts once the backbone's is done
                T.wait_event(e["B2"])
                e["T3s"].record(T)
            st.gT3.replay()
            e["T3"].record(T)
            T.wait_event(e["B2"])
            e["Os"].record(T)
            st.gO.replay()
            e["O"].record(T)
        torch.cuda.synchronize()
        if not os.environ.get("SERIAL"):
            e["T3s"] = e["T2"]
        for n in names:
            acc[n] += e["start"].elapsed_time(e[n]) * 1e3
    print("phase boundaries, us after the step's start (mean of %d steps):" % reps)
    for n in names:
        print(f"  {n:6s} {acc[n] / reps:8.1f}")


main()
