#!/usr/bin/env python3
"""bench.py -- forecast windows/sec of the multimodal-fusion training step on MI355X.

Workload (BASELINE.json configs[1]): tPatchGNN + TTF_T2V_XAttn + MMF_XAttn_Add, GPT-2 sized note embeddings
(d_m = d_txt = 768, H = 1), a ragged batch of 64 windows (one per entity) PER GPU, bf16 MFMA operands with fp32
accumulation, train mode with the reference's default dropout 0.1.  One step = backbone forecast -> fusion ->
masked-MSE loss -> backward -> (N>1: RCCL gradient all-reduce) -> clip_grad_norm(1.0) + Adam.  Inputs are
synthetic, generated once and resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]

For N>1 the driver launches one rank per GPU with torch.distributed.run; entities shard across ranks (weak
scaling: 64 windows per GPU).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "imm-tsf_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# ---- workload constants ---------------------------------------------------------------
B_PER_GPU, C, M_PATCH, L_PATCH, N_MAX, T_MAX, D_M, D_TXT, H = 64, 8, 2, 32, 32, 32, 768, 768, 1
P_DROP, KAPPA = 0.1, 0.5
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def model_args(device):
    return types.SimpleNamespace(
        device=device, hid_dim=32, C=C, npatch=M_PATCH, nlayer=1, te_dim=10, n_heads=1, tf_layer=1, node_dim=10, hop=1,
        outlayer="Linear", TTF_module="TTF_T2V_XAttn", MMF_module="MMF_XAttn_Add", llm_model_fusion="GPT2",
        llm_layers_fusion=6, max_length=1024, use_text_embeddings=True, recency_sigma=1.0, n_heads_fusion=H,
        dropout=P_DROP, d_txt=D_TXT, kappa=KAPPA, batch_size=B_PER_GPU)


def synth_batch(seed, B):
    """cfg2-shaped ragged batch: all fp32, CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    rng = np.random.default_rng(seed)
    # history, patched (B, M, L, C): ragged observation counts, 30 % empty patches
    cnt = rng.integers(1, L_PATCH + 1, size=(B, M_PATCH, C))
    cnt[rng.random((B, M_PATCH, C)) < 0.3] = 0
    l_idx = np.arange(L_PATCH).reshape(1, 1, L_PATCH, 1)
    obs_mask = torch.from_numpy((l_idx < cnt[:, :, None, :]).astype(np.float32))
    X = torch.randn(B, M_PATCH, L_PATCH, C, generator=g) * obs_mask
    tt = torch.sort(torch.rand(B, M_PATCH, L_PATCH, C, generator=g), dim=2).values * obs_mask
    # notes
    n_notes = torch.from_numpy(rng.integers(1, N_MAX + 1, size=B))
    notes = torch.randn(B, N_MAX, D_M, generator=g)
    tau = torch.sort(torch.rand(B, N_MAX, generator=g) * 24.0, dim=1).values
    keep = (torch.arange(N_MAX).view(1, -1) < n_notes.view(-1, 1))
    notes = notes * keep.unsqueeze(-1)
    tau = tau * keep
    # horizon
    t_len = torch.from_numpy(rng.integers(8, T_MAX + 1, size=B))
    tvalid = (torch.arange(T_MAX).view(1, -1) < t_len.view(-1, 1))
    t_hat = torch.sort(24.0 + 24.0 * torch.rand(B, T_MAX, generator=g), dim=1).values / 48.0 * tvalid
    truth = torch.randn(B, T_MAX, C, generator=g)
    tmask = (torch.rand(B, T_MAX, C, generator=g) < 0.7).float()
    tmask[:, 0, :] = torch.maximum(tmask[:, 0, :], (tmask.sum(1) == 0).float())     # >= 1 observation per row
    tmask = tmask * tvalid.unsqueeze(-1)
    return dict(observed_data=X, observed_tp=tt, observed_mask=obs_mask, tp_to_predict=t_hat, notes_embeddings=notes,
                tau=tau, data_to_predict=truth * tmask, mask_predicted_data=tmask), int(n_notes.sum())


def fusion_flops_per_window(sum_n, B):
    """SURVEY 8d algorithmic FLOPs (fwd+bwd = 3x fwd for the GEMM terms), per window, cfg2."""
    d, dt, T, nbar = D_TXT, D_TXT // 2, T_MAX, sum_n / B
    f_t2v = sum_n * (2 * D_M * d + 2 * (d + dt) * d + 4 * d * d) + B * T * (4 * nbar * d + 2 * d * d) + B * T * 2 * d * d
    f_xadd = B * T * (12 * d * d + 4 * T * d + 4 * C * d)
    return 3.0 * (f_t2v + f_xadd) / B


def cpu_baseline(batch, steps=4, warmup=1):
    """The oracle (CPU restatement, op-for-op incl. the T-fold K/V expansion) timed on this box's host cores: same
    batch, same region (backbone fwd -> fusion fwd -> masked MSE -> backward -> clip -> Adam), dropout masks drawn on
    the CPU each step like torch's dropout does."""
    from models.tPatchGNN import tPatchGNN
    from oracle import fusion_ref as R
    from fusions.FusionModel import FusionModel
    cores = os.cpu_count() or 1
    torch.manual_seed(0)
    a = model_args("cpu")
    a.immtsf_patch_encoder = "torch"
    model = tPatchGNN(a).train()
    fus = FusionModel(a)          # parameter container only; the arithmetic below is the oracle's
    params = {k: v.detach().clone().requires_grad_(True) for k, v in fus.state_dict().items()}
    opt = torch.optim.Adam(list(model.parameters()) + list(params.values()), lr=1e-3)
    B, T = batch["tp_to_predict"].shape
    keep = 1.0 - P_DROP

    def step():
        opt.zero_grad(set_to_none=True)
        pred = model.forecasting(batch["tp_to_predict"], batch["observed_data"], batch["observed_tp"], batch["observed_mask"])
        drop = {"ttf": {"attn": torch.bernoulli(torch.full((B, T, H, N_MAX), keep)),
                        "out": torch.bernoulli(torch.full((B, T, D_TXT), keep))},
                "mmf": {"attn": torch.bernoulli(torch.full((B, H, T, T), keep)),
                        "out": torch.bernoulli(torch.full((B, T, C), keep))}}
        out = R.fusion_forward("TTF_T2V_XAttn", "MMF_XAttn_Add", params, batch["notes_embeddings"], batch["tau"],
                               batch["tp_to_predict"], pred, H=H, kappa=KAPPA, drop=drop, p_drop=P_DROP, expand_T=True)
        loss = R.masked_mse(batch["data_to_predict"], out, batch["mask_predicted_data"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(model.parameters()) + list(params.values()), 1.0)
        opt.step()

    # pick the torch thread count that serves this op mix best on this host (all cores is far from optimal for
    # the many small ops; the chosen count is what `cores` reports)
    best = None
    for nt in sorted({min(cores, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (nt, dt)
        if dt > 8.0:
            break
    cores = best[0]
    torch.set_num_threads(cores)
    for _ in range(warmup):
        step()
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": round(B / med, 2), "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps of the same {B}-window cfg2 batch after {warmup} warm-up (median {med*1e3:.0f} ms/step), "
                      f"oracle/fusion_ref.py with the T-expanded K/V + torch-CPU tPatchGNN, fp32, dropout {P_DROP}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying hipGraphs")
    ap.add_argument("--gemm-config", type=lambda x: int(x, 0), default=0,
                    help="A/B measurements only: immtsf_debug_gemm_config bits (0x100 no XCD order, 0x2000 no specialised wgrad kernel)")
    ap.add_argument("--gemm2-variant", type=int, default=0, help="A/B measurements only: force one tile variant of the bf16-in-memory GEMM")
    ap.add_argument("--phased", action="store_true",
                    help="immtsf.train.PhasedStep: six single-stream hipGraphs on two HIP streams with events between them, instead "
                         "of the whole step as parallel branches of one hipGraph (DESIGN.md section 6 has both measured)")
    ap.add_argument("--captured-comm", action="store_true",
                    help="N>1 graph mode: capture the bucketed RCCL all-reduces inside graph A (overlapped with the backward). "
                         "Verified here only on a 1-rank group, so the default is one eager all-reduce between the two graphs")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even for one rank: exercises the N>1 code path on a 1-GPU box")
    ap.add_argument("--no-overlap", action="store_true", help="do not run the backbone on a second HIP stream beside TTF")
    ap.add_argument("--grad-wire", default="auto", choices=["auto", "fp32", "bf16"],
                    help="N>1: element type of the gradient all-reduce; auto = bf16 in bf16 mode (half the xGMI bytes), fp32 otherwise")
    ap.add_argument("--no-wgrad-fork", action="store_true",
                    help="A/B measurements only: weight-gradient GEMMs on the caller's stream instead of the library's side stream")
    ap.add_argument("--windows-per-gpu", type=int, default=B_PER_GPU,
                    help="exploration only (DESIGN.md section 8, batch-size table): the metric is quoted on 64 windows per GPU")
    args = ap.parse_args()
    globals()["B_PER_GPU"] = args.windows_per_gpu
    # stdout carries exactly one line, the JSON result: everything else a module prints (the fusion registry announces
    # its choices like the reference does) goes to stderr
    json_out, sys.stdout = sys.stdout, sys.stderr

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run); got {world}")
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    # control-flow test of the N>1 path on a 1-GPU box: IMMTSF_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo
    # (RCCL refuses two ranks on one device); numbers from such a run mean nothing
    share_gpu = bool(os.environ.get("IMMTSF_BENCH_SHARE_GPU"))
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        group = dist.group.WORLD

    from fusions.FusionModel import FusionModel
    from immtsf import _lib, config
    from immtsf.ops import backward_unit, masked_mse
    from immtsf.train import FlatTrainer, GraphedStep, PhasedStep
    from models.tPatchGNN import tPatchGNN
    lib = _lib.load()
    if args.gemm_config:
        lib.immtsf_debug_gemm_config(args.gemm_config, 0)
    if args.gemm2_variant:
        lib.immtsf_debug_gemm2_config(args.gemm2_variant, 0, -1)
    if args.no_wgrad_fork:
        lib.immtsf_set_side_stream(0)
    config.precision = args.precision
    config.nan_check = "deferred"       # no host syncs inside the step; the flag is checked after the run
    config.manual_seed(1234 + rank)

    torch.manual_seed(0)                # identical initial weights on every rank
    a = model_args(str(dev))
    model = tPatchGNN(a).to(dev).train()
    fusion = FusionModel(a).to(dev).train()
    use_graph = not args.no_graph
    wire = args.grad_wire if args.grad_wire != "auto" else ("bf16" if args.precision == "bf16" else "fp32")
    trainer = FlatTrainer([list(fusion.mmf.parameters()), list(fusion.ttf.parameters()), list(model.parameters())],
                          lr=1e-3, weight_decay=0.0, max_norm=1.0, group=group, sink_buckets=(0, 1, 2), sink_exclude=[model.te_scale.weight, model.te_scale.bias, model.te_periodic.weight, model.te_periodic.bias],
                          overlap=True, device_step=use_graph, grad_wire=wire)
    comm_mode = "bucketed on a side stream" if dist_on else "none"
    cpu_batch, sum_n = synth_batch(100 + rank, B_PER_GPU)
    batch = {k: v.to(dev) for k, v in cpu_batch.items()}
    # per-variable observation counts of the GLOBAL batch: a property of the data (mask), reduced once when the batch
    # is built, so the step itself has no collective besides the gradient all-reduce
    global_cnt = batch["mask_predicted_data"].reshape(-1, C).sum(0)
    if dist_on:
        import torch.distributed as dist
        dist.all_reduce(global_cnt)

    from lib.evaluation import forecast_and_fuse
    backbone_stream = None if args.no_overlap else torch.cuda.Stream(device=dev)

    def loss_fn():
        out = forecast_and_fuse(model, fusion, batch, backbone_stream)
        return masked_mse(out, batch["data_to_predict"], batch["mask_predicted_data"], None, global_cnt)

    def eager_step():
        trainer.zero_grad()
        loss = loss_fn()
        backward_unit(loss)
        trainer.sync_grads()
        trainer.step()
        return loss

    # hipGraph replay (immtsf.train.GraphedStep): graph A = zero-grad, backbone + fusion forward, loss, backward, gradient
    # collection; [N>1: eager RCCL all-reduce of the flat gradient]; graph B = clip + Adam + device-side counters
    launch_mode = ("hipGraph replay (2 graphs/step)" if use_graph else "eager") + \
                  ("" if args.no_overlap else ", backbone on a second HIP stream beside TTF")
    if use_graph:
        step = None
        if dist_on and args.captured_comm:
            # RCCL all-reduces captured inside graph A, bucket by bucket on the communication stream while the backward
            # of the later buckets still runs.  A failed capture leaves the HIP context unusable (seen with gloo, which
            # cannot be captured), so there is no fallback: run without the flag instead
            try:
                step = GraphedStep(trainer, loss_fn, capture_collectives=True)
                comm_mode = "captured, bucketed"
            except Exception as e:      # noqa: BLE001
                raise SystemExit(f"--captured-comm: the collectives could not be captured ({type(e).__name__}); "
                                 "re-run without the flag (eager all-reduce between the two graphs)") from e
        if step is None and args.phased and not args.no_overlap and hasattr(fusion.mmf, "project_kv"):
            # two streams, six single-chain graphs, events in between (immtsf.train.PhasedStep)
            fc_args = (batch["tp_to_predict"], batch["observed_data"], batch["observed_tp"], batch["observed_mask"])

            def text_fn():
                E, M = fusion.ttf(batch["notes_embeddings"], batch["tau"], batch["tp_to_predict"])
                return (E, M) + tuple(fusion.mmf.project_kv(E))

            def head_fn(pred, E, M, kv, fold):
                out = fusion.mmf(pred, E, M, kv=(kv, fold))
                return masked_mse(out, batch["data_to_predict"], batch["mask_predicted_data"], None, global_cnt)

            if dist_on:
                comm_mode = "eager, in front of the optimizer graph"
            step = PhasedStep(trainer, text_fn, lambda: model.forecasting(*fc_args), head_fn)
            launch_mode = "hipGraph replay: 6 single-stream graphs per step on 2 HIP streams (text side | backbone), HIP events between them"
        if step is None:
            if dist_on:
                trainer.overlap, comm_mode = False, "eager, between the graphs"
            step = GraphedStep(trainer, loss_fn)
    else:
        step = eager_step

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_enqueue_s = time.perf_counter() - t0      # host time to enqueue the timed steps (no sync inside the loop)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fusion.check_nan()
    assert torch.isfinite(loss).all(), "loss is not finite"
    ms_per_step = elapsed / args.steps * 1e3
    windows_per_s = B_PER_GPU * world * args.steps / elapsed

    roofline = None
    if not args.no_roofline and rank == 0:
        # dominant kernel = the MFMA GEMM (all projections).  HIP events bracket every GEMM launch on torch's current
        # stream (the stream the library launches on) over a further block of identical steps.
        k2 = min(args.steps, 20)
        lib.immtsf_timing_enable(1)
        # rank 0 only: the other ranks are already waiting at the final barrier, so this leg must not issue collectives
        was_collective, trainer.collective = trainer.collective, False
        for _ in range(k2):
            eager_step()      # the tap records at launch time, so this leg launches eagerly (same kernels, same shapes)
        trainer.collective = was_collective
        torch.cuda.synchronize()
        cap = 16384
        meta = (ctypes.c_int32 * (10 * cap))()
        ms = (ctypes.c_float * cap)()
        n = lib.immtsf_timing_collect(cap, meta, ms)
        lib.immtsf_timing_enable(0)
        groups = {}
        for i in range(n):
            key = tuple(meta[10 * i:10 * i + 9])
            groups.setdefault(key, []).append(ms[i])
        rows = []
        for key, v in groups.items():
            layout, prec, Mm, Nn, Kk, nprob, nbatch, dyn, grid_threads = key
            if dyn == 1:
                Mm = sum_n
            elif dyn == 2:
                Kk = sum_n
            fl = 2.0 * Mm * Nn * Kk * nprob * nbatch
            rows.append(dict(key=key, launches=len(v), avg_us=1e3 * float(np.mean(v)), total_ms=float(np.sum(v)), flops=fl))
        rows.sort(key=lambda r: -r["total_ms"])
        if os.environ.get("IMMTSF_BENCH_GEMM_TABLE"):
            for r in rows:
                k = r["key"]
                print(f"# gemm {['NT','NN','TN'][k[0]]} M={k[2]:6d} N={k[3]:5d} K={k[4]:6d} prob={k[5]} batch={k[6]:4d} dyn={k[7]} "
                      f"launches/step={r['launches']/k2:5.1f} avg_us={r['avg_us']:7.1f} us/step={r['total_ms']*1e3/k2:7.1f} "
                      f"TF={r['flops']/(r['avg_us']*1e-6)/1e12:7.2f}", file=sys.stderr)
        gemm_ms = sum(r["total_ms"] for r in rows) / k2
        top = rows[0]
        # the tap brackets eager launches, so its interval also holds the launch latency between the two event records;
        # re-time the dominant instance itself: 50 back-to-back launches of exactly that GEMM captured in one hipGraph,
        # HIP events around replays on the launch stream -> mean kernel duration (this is what rocprofv3 reports)
        lay_i, _, Mm, Nn, Kk, nprob = top["key"][0], top["key"][1], top["key"][2], top["key"][3], top["key"][4], top["key"][5]
        if top["key"][7] == 1:
            Mm = sum_n
        elif top["key"][7] == 2:
            Kk = sum_n
        shapes = {0: ((Mm, Kk), (Nn, Kk)), 1: ((Mm, Kk), (Kk, Nn)), 2: ((Kk, Mm), (Kk, Nn))}[lay_i]
        Ab, Bb = torch.randn(*shapes[0], device=dev), torch.randn(*shapes[1], device=dev)
        Cb = torch.empty(Mm, Nn, device=dev)
        prec_code = 1 if args.precision == "bf16" else 0
        # in the step the weight operand of a forward / data-gradient GEMM is read from FlatTrainer's bf16 twin: same here
        twin_b = None
        if prec_code == 1 and lay_i != 2 and trainer.flat_twin is not None:
            twin_b = Bb.to(torch.bfloat16).contiguous()
            _lib.check(lib.immtsf_bf16_twin_register(_lib.ptr(Bb), _lib.ptr(twin_b), Bb.numel()), "bf16_twin_register")

        def one():
            for _ in range(nprob):
                _lib.check(lib.immtsf_gemm(lay_i, prec_code, _lib.ptr(Ab), Ab.shape[1], _lib.ptr(Bb), Bb.shape[1], _lib.ptr(Cb),
                                           Nn, None, Mm, Nn, Kk, 1.0, 0, 0, _lib.stream_ptr()), "gemm")
        side2 = torch.cuda.Stream(device=dev)
        side2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side2):
            one()
        torch.cuda.current_stream().wait_stream(side2)
        gg = torch.cuda.CUDAGraph()
        reps = 50
        with torch.cuda.graph(gg):
            for _ in range(reps):
                one()
        gg.replay()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10):
            gg.replay()
        ev1.record()
        torch.cuda.synchronize()
        kernel_us = ev0.elapsed_time(ev1) / (10 * reps * nprob) * 1e3
        if twin_b is not None:
            lib.immtsf_bf16_twin_unregister(_lib.ptr(Bb))
        ach = (top["flops"] / nprob) / (kernel_us * 1e-6) / 1e12
        allfl = sum(r["flops"] * r["launches"] for r in rows) / sum(r["total_ms"] for r in rows) / 1e9
        lay = {0: "NT", 1: "NN", 2: "TN"}[top["key"][0]]
        # HBM traffic of that kernel instance from the committed rocprofv3 PMC passes (tools/pmc_summary.py): matched by
        # kernel template + launch grid
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            tag = {0: "<true, false, false", 1: "<true, false, true", 2: "<true, true, true"}[top["key"][0]]
            for kr in pmc["kernels"]:
                if "gemm_kernel" + tag in kr["kernel"] and kr["grid_threads"] == top["key"][8]:
                    traffic = kr["fetch_bytes_per_launch"] + (kr["write_bytes_per_launch"] or 0)
                    break
        except Exception:
            traffic = None
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS if args.precision == "bf16" else 157.3,
                    "unit": "TFLOP/s", "frac": round(ach / (PEAK_BF16_TFLOPS if args.precision == "bf16" else 157.3), 5),
                    "traffic": traffic, "traffic_unit": "bytes/launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_traffic.json)",
                    "algorithmic_bytes": 4 * Mm * Kk + (2 if twin_b is not None else 4) * Nn * Kk + 4 * Mm * Nn,
                    "b_operand": "bf16 twin of the weights" if twin_b is not None else "fp32",
                    "kernel": f"gemm_kernel {lay} M={Mm} N={Nn} K={Kk} (x{top['key'][5] * top['key'][6]} problems per launch in the step; per-problem figures here)",
                    "avg_launch_us": round(kernel_us, 2), "avg_launch_us_eager_tap": round(top["avg_us"], 2),
                    "launches_per_step": top["launches"] // k2,
                    "all_gemm_ms_per_step": round(gemm_ms, 4), "all_gemm_tflops": round(allfl, 2),
                    "gemm_launches_per_step": sum(r["launches"] for r in rows) // k2}

    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(cpu_batch)

    if rank == 0:
        fl_win = fusion_flops_per_window(sum_n, B_PER_GPU)
        line = {
            "metric": "forecast windows/sec (train fwd+bwd) on ragged 64-entity batch", "value": round(windows_per_s, 1),
            "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "launch": launch_mode, "host_enqueue_ms_per_step": round(host_enqueue_s / args.steps * 1e3, 4),
            "config": {"workload": "cfg2: tPatchGNN + TTF_T2V_XAttn + MMF_XAttn_Add, GPT2 dims (d_m=d_txt=768, H=1), "
                                   f"{B_PER_GPU} ragged windows per GPU (N_b~U{{1..32}}, T=32, C=8, M=2 patches, L<=32), dropout 0.1",
                       "step": "backbone fwd + fusion fwd + masked MSE + backward + grad all-reduce (N>1) + clip + Adam",
                       "global_batch": B_PER_GPU * world, "parallelism": f"dp{world}", "sum_notes_rank0": sum_n,
                       "fusion_algorithmic_gflop_per_window": round(fl_win / 1e9, 4),
                       "fusion_algorithmic_tflops_at_step_time": round(fl_win * B_PER_GPU * world / (ms_per_step * 1e-3) / 1e12, 2),
                       "grad_bytes": trainer.grad_bytes(),
                       "grad_allreduce": comm_mode + (f", {wire} on the wire" if dist_on else "")},
            "roofline": roofline, "cpu_baseline": cpu}
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()      # RCCL may log on teardown: keep the JSON line the last thing rank 0 prints
    if rank == 0:
        sys.stderr.flush()
        print(json.dumps(line), file=json_out, flush=True)
    # native libraries (RCCL prints its path on unload) must not write to stdout behind the JSON line
    json_out.flush()
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)


if __name__ == "__main__":
    main()
