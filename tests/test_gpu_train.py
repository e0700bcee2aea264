"""GPU tests of the training-step plumbing: hipGraph replay (immtsf.train.GraphedStep) must train exactly like the
eager step -- including the autograd-owned backbone gradients, which have to be collected INSIDE the graph -- and the
two-stream backbone/TTF overlap must not change results."""
import os
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _setup(dev, dropout, seed=0, trainer=True):
    import bench
    from fusions.FusionModel import FusionModel
    from fusions.load_llm import register_d_model
    from immtsf import config
    from immtsf.train import FlatTrainer
    from models.tPatchGNN import tPatchGNN
    register_d_model("TOY48", 48)
    config.precision = "fp32"
    config.nan_check = "deferred"
    config.manual_seed(77)
    torch.manual_seed(seed)
    a = types.SimpleNamespace(
        device=str(dev), hid_dim=16, C=bench.C, npatch=bench.M_PATCH, nlayer=1, te_dim=6, n_heads=1, tf_layer=1, node_dim=5,
        hop=1, outlayer="Linear", TTF_module="TTF_T2V_XAttn", MMF_module="MMF_XAttn_Add", llm_model_fusion="TOY48",
        llm_layers_fusion=6, max_length=1024, use_text_embeddings=True, recency_sigma=1.0, n_heads_fusion=2,
        dropout=dropout, d_txt=32, kappa=0.5, batch_size=8)
    model = tPatchGNN(a).to(dev).train()
    fusion = FusionModel(a).to(dev).train()
    for m in model.modules():           # the stock transformer layer's dropout draws from torch's generator
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    for lyr in model.transformer_encoder:
        for l in lyr.layers:
            l.self_attn.dropout = 0.0
    trainer = None if not trainer else FlatTrainer([list(fusion.mmf.parameters()), list(fusion.ttf.parameters()), list(model.parameters())],
                          lr=1e-2, eps=1e-3, max_norm=1.0, sink_buckets=(0, 1), overlap=False, device_step=True)   # eps: Adam's sign-like first
    # steps would otherwise turn 1e-9 atomic-order noise on ~zero gradients into +-lr parameter differences
    cpu_batch, _ = bench.synth_batch(5, 8)
    batch = {k: v.to(dev) for k, v in cpu_batch.items()}
    batch["notes_embeddings"] = batch["notes_embeddings"][..., :48].contiguous()
    return model, fusion, trainer, batch


def _loss_fn(model, fusion, batch, stream=None):
    from immtsf.ops import masked_mse
    from lib.evaluation import forecast_and_fuse

    def f():
        out = forecast_and_fuse(model, fusion, batch, stream)
        return masked_mse(out, batch["data_to_predict"], batch["mask_predicted_data"])
    return f


def test_graph_replay_trains_like_eager_incl_backbone():
    dev = _dev()
    from immtsf.train import GraphedStep
    steps = 4
    # eager
    model, fusion, tr, batch = _setup(dev, 0.0)
    f = _loss_fn(model, fusion, batch)
    for _ in range(steps):              # GraphedStep's warm-up steps are undone: replay 1 is step 1
        tr.zero_grad()
        f().backward()
        tr.sync_grads()
        tr.step()
    ref = tr.flat_param.clone()
    lo, hi = tr.ranges[2]               # the autograd-owned backbone bucket
    # graphs
    model, fusion, tr, batch = _setup(dev, 0.0)
    g = GraphedStep(tr, _loss_fn(model, fusion, batch))
    after_warm = tr.flat_param.clone()
    losses = [float(g().detach()) for _ in range(steps)]
    torch.cuda.synchronize()
    assert losses[-1] < losses[0]
    assert float((tr.flat_param[lo:hi] - after_warm[lo:hi]).abs().max()) > 1e-3     # the backbone keeps training
    err = float((tr.flat_param - ref).abs().max() / ref.abs().max())
    assert err < 2e-4, err


def test_two_stream_overlap_same_result_and_dropout_advances():
    dev = _dev()
    from immtsf.train import GraphedStep
    model, fusion, tr, batch = _setup(dev, 0.0)
    a = _loss_fn(model, fusion, batch)()
    b = _loss_fn(model, fusion, batch, torch.cuda.Stream(device=dev))()
    torch.cuda.synchronize()
    assert abs(float(a) - float(b)) <= 1e-6 * abs(float(a))
    # with dropout on, every replay must draw new masks (device-side seed counter): frozen weights -> different losses
    model, fusion, tr, batch = _setup(dev, 0.3)
    tr.lr = 0.0
    g = GraphedStep(tr, _loss_fn(model, fusion, batch, torch.cuda.Stream(device=dev)))
    ls = [float(g().detach()) for _ in range(4)]
    assert len({round(v, 7) for v in ls}) == 4, ls


def test_bf16_twin_operand_is_bit_identical():
    """a GEMM whose weight operand has a registered bf16 twin must return exactly what the fp32-operand GEMM returns
    (the same values are rounded to bf16 either way), for NT (forward) and NN (data gradient), incl. a ragged K tail"""
    dev = _dev()
    from immtsf import _lib
    lib = _lib.load()
    torch.manual_seed(0)
    for layout, M, N, K in [(0, 300, 192, 160), (1, 300, 160, 192), (0, 2048, 768, 768), (1, 2048, 768, 768), (0, 70, 64, 100)]:
        W = torch.randn((N, K) if layout == 0 else (K, N), device=dev)
        A = torch.randn(M, K, device=dev)
        bias = torch.randn(N, device=dev)
        outs = []
        for use_twin in (False, True):
            if use_twin:
                twin = W.to(torch.bfloat16).contiguous()
                _lib.check(lib.immtsf_bf16_twin_register(_lib.ptr(W), _lib.ptr(twin), W.numel()), "reg")
            Cm = torch.empty(M, N, device=dev)
            _lib.check(lib.immtsf_gemm(layout, 1, _lib.ptr(A), K, _lib.ptr(W), W.shape[1], _lib.ptr(Cm), N, _lib.ptr(bias), M, N, K,
                                       1.0, 0, 0, _lib.stream_ptr()), "gemm")
            torch.cuda.synchronize()
            outs.append(Cm)
        lib.immtsf_bf16_twin_unregister(_lib.ptr(W))
        assert torch.equal(outs[0], outs[1]), (layout, M, N, K)
        ref = (A @ W.t() if layout == 0 else A @ W) + bias
        assert float((outs[1] - ref).abs().max() / ref.abs().max()) < 2e-2


def test_training_with_twins_matches_without():
    """bf16 mode, dropout off: FlatTrainer's bf16 parameter twin (written by the fused Adam kernel every step) must not
    change the trajectory -- a stale twin would."""
    dev = _dev()
    from immtsf import config
    res = []
    for twin in (False, True):
        model, fusion, tr, batch = _setup(dev, 0.0)
        tr.close()
        from immtsf.train import FlatTrainer
        tr = FlatTrainer([list(fusion.mmf.parameters()), list(fusion.ttf.parameters()), list(model.parameters())],
                         lr=1e-2, eps=1e-3, max_norm=1.0, sink_buckets=(0, 1), overlap=False, device_step=True, bf16_twin=twin)
        config.precision = "bf16"
        f = _loss_fn(model, fusion, batch)
        ls = []
        for _ in range(5):
            tr.zero_grad()
            l = f()
            l.backward()
            tr.sync_grads()
            tr.step()
            ls.append(float(l.detach()))
        res.append((ls, tr.flat_param.clone()))
        assert (tr.flat_twin is not None) == twin
        if twin:
            assert torch.equal(tr.flat_twin, tr.flat_param.to(torch.bfloat16))
        tr.close()
    config.precision = "fp32"
    assert res[0][0][-1] < res[0][0][0]
    assert max(abs(a - b) for a, b in zip(res[0][0], res[1][0])) < 2e-4 * abs(res[0][0][0])


def test_evaluation_metrics_match_reference_formula():
    """lib.evaluation.evaluation (device-side accumulation, one host transfer) against the reference's definition
    (lib/evaluation.py:192-283) evaluated with plain torch over the same predictions"""
    dev = _dev()
    import bench
    from lib.evaluation import evaluation
    model, fusion, tr, _ = _setup(dev, 0.0)
    tr.close()
    model.eval()
    fusion.eval()
    batches = []
    for seed in (11, 12, 13):
        cpu_batch, _ = bench.synth_batch(seed, 8)
        b = {k: v.to(dev) for k, v in cpu_batch.items()}
        b["notes_embeddings"] = b["notes_embeddings"][..., :48].contiguous()
        batches.append(b)
    got = evaluation(model, fusion, batches)
    se = ae = ape = cnt = cnt_ape = 0
    with torch.no_grad():
        for b in batches:
            pred = fusion(b["notes_embeddings"], b["tau"], b["tp_to_predict"],
                          model.forecasting(b["tp_to_predict"], b["observed_data"], b["observed_tp"], b["observed_mask"]))
            t, m = b["data_to_predict"], b["mask_predicted_data"]
            C = t.shape[-1]
            se = se + (((t - pred) ** 2) * m).reshape(-1, C).sum(0)
            ae = ae + ((t - pred).abs() * m).reshape(-1, C).sum(0)
            m2 = (t != 0) * m
            ape = ape + ((t - pred).abs() / (t + (t == 0) * 1e-8) * m2).reshape(-1, C).sum(0)
            cnt = cnt + m.reshape(-1, C).sum(0)
            cnt_ape = cnt_ape + m2.reshape(-1, C).sum(0)
    mse = float(((se / (cnt + 1e-8)).sum() / torch.count_nonzero(cnt)))
    mae = float(((ae / (cnt + 1e-8)).sum() / torch.count_nonzero(cnt)))
    mape = float(((ape / (cnt_ape + 1e-8)).sum() / torch.count_nonzero(cnt_ape)))
    assert set(got) == {"loss", "mse", "mae", "rmse", "mape"} and all(isinstance(v, float) for v in got.values())
    for k, v in (("loss", mse), ("mse", mse), ("mae", mae), ("rmse", mse ** 0.5), ("mape", mape)):
        assert abs(got[k] - v) <= 1e-5 * max(1.0, abs(v)), (k, got[k], v)


def _two_rank_worker(rank, world, port, q, wire):
    """one rank of the data-parallel graph step: both ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one
    device) -- what is under test is the N>1 control flow of GraphedStep / FlatTrainer, not the transport"""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "imm-tsf_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from immtsf.ops import masked_mse
    from immtsf.train import FlatTrainer, GraphedStep, shard_range
    from lib.evaluation import forecast_and_fuse
    model, fusion, _, batch = _setup(dev, 0.0)
    lo, hi = shard_range(8, rank, world)
    shard = {k: v[lo:hi].contiguous() for k, v in batch.items()}
    cnt = shard["mask_predicted_data"].reshape(-1, shard["mask_predicted_data"].shape[-1]).sum(0)
    dist.all_reduce(cnt)                # observation counts of the GLOBAL batch: data only, reduced once
    tr = FlatTrainer([list(fusion.mmf.parameters()), list(fusion.ttf.parameters()), list(model.parameters())], lr=1e-2, eps=1e-3,
                     max_norm=1.0, sink_buckets=(0, 1), overlap=True, device_step=True, group=dist.group.WORLD, grad_wire=wire)
    tr.overlap = False                  # what bench.py does for graph mode: one eager all-reduce between the two graphs

    def f():
        out = forecast_and_fuse(model, fusion, shard, None)
        return masked_mse(out, shard["data_to_predict"], shard["mask_predicted_data"], None, cnt)
    g = GraphedStep(tr, f)              # (its 3 warm-up steps are undone)
    for _ in range(5):
        g()
    torch.cuda.synchronize()
    if rank == 0:
        q.put(tr.gather(tr.flat_param).cpu().numpy())        # by value: the process exits before the parent reads it
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wire,tol", [("fp32", 3e-4), ("bf16", 3e-2)])
def test_two_rank_graph_step_equals_single_process(wire, tol):
    """bench.py's N>1 path (graph A -> eager all-reduce -> graph B) on two ranks trains like one process on the full batch;
    the bucket hooks must NOT issue collectives inside the captured backward when overlap is off (they would be captured
    and then repeated eagerly: gradients x world)."""
    dev = _dev()
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q, wire)) for r in range(2)]
    for p in procs:
        p.start()
    got = torch.from_numpy(q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model, fusion, tr, batch = _setup(dev, 0.0)
    f = _loss_fn(model, fusion, batch)
    for _ in range(5):
        tr.zero_grad()
        f().backward()
        tr.sync_grads()
        tr.step()
    ref = tr.gather(tr.flat_param).cpu()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < tol, err


def _two_rank_eager_worker(rank, world, port, q, sharded):
    """the EAGER data-parallel step with overlap=True (bucket hooks + communication stream), all-reduce or sharded optimizer"""
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "imm-tsf_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from immtsf.ops import masked_mse
    from immtsf.train import FlatTrainer, shard_range
    from lib.evaluation import forecast_and_fuse
    model, fusion, _, batch = _setup(dev, 0.0)
    lo, hi = shard_range(8, rank, world)
    shard = {k: v[lo:hi].contiguous() for k, v in batch.items()}
    cnt = shard["mask_predicted_data"].reshape(-1, shard["mask_predicted_data"].shape[-1]).sum(0)
    dist.all_reduce(cnt)
    tr = FlatTrainer([list(fusion.mmf.parameters()), list(fusion.ttf.parameters()), list(model.parameters())], lr=1e-2, eps=1e-3,
                     max_norm=1.0, sink_buckets=(0, 1), overlap=True, device_step=False, group=dist.group.WORLD, grad_wire="fp32",
                     shard_optimizer=sharded)
    assert tr.overlap == (not sharded)          # the sharded optimizer never all-reduces buckets from the backward hooks
    for _ in range(4):
        tr.zero_grad()
        out = forecast_and_fuse(model, fusion, shard, None)
        masked_mse(out, shard["data_to_predict"], shard["mask_predicted_data"], None, cnt).backward()
        tr.sync_grads()
        tr.step()
    torch.cuda.synchronize()
    if rank == 0:
        q.put(tr.gather(tr.flat_param).cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sharded", [False, True])
def test_two_rank_eager_step_equals_single_process(sharded):
    """eager N>1 step with overlap=True -- what bench.py runs for the configurations it cannot graph and under --no-graph: the
    sharded optimizer must not have its sink buckets all-reduced by the backward hooks before the reduce-scatter sums them again
    (round-2 advisor finding: gradients x world), and must not leave the communication stream unjoined."""
    dev = _dev()
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_eager_worker, args=(r, 2, port, q, sharded)) for r in range(2)]
    for p in procs:
        p.start()
    got = torch.from_numpy(q.get(timeout=300))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model, fusion, tr, batch = _setup(dev, 0.0)
    f = _loss_fn(model, fusion, batch)
    for _ in range(4):
        tr.zero_grad()
        f().backward()
        tr.sync_grads()
        tr.step()
    ref = tr.gather(tr.flat_param).cpu()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 3e-4, err


def _named_flat(model, fusion):
    """every parameter, concatenated in NAME order (a trainer's own gather() follows its bucket layout, which differs between trainers)"""
    named = sorted([("m." + k, p) for k, p in model.named_parameters()] + [("f." + k, p) for k, p in fusion.named_parameters()])
    return torch.cat([p.detach().reshape(-1) for _, p in named])


def _two_rank_flag_worker(rank, world, port, q, wire, layout="phases", inject=False):
    """one rank of the data-parallel FLAG step (bench.py's N > 1 default): ONE graph per step, every bucket rounded to the wire image
    where it completes and announced by a counting flag, the all-reduces on the communication stream beside the backward, clip + Adam of
    step k at the head of replay k + 1 reading the reduced wire.  Both ranks share cuda:0 over gloo: the control flow is under test,
    not the transport.  layout "phases": bench.py's buckets (FusionModel.grad_buckets: MMF, TTF's three backward phases, the backbone;
    Adam split over the branches); "blocks": one bucket per block, everything on the text branch.  inject: after two good steps rank 1
    waits for a flag nobody sets -- BOTH ranks must drop that step."""
    try:
        _two_rank_flag_body(rank, world, port, q, wire, layout, inject)
    except BaseException:       # noqa: BLE001 -- the parent must not sit out its queue time-out when a rank dies
        import traceback
        q.put(("error", rank, traceback.format_exc()))
        raise


def _two_rank_flag_body(rank, world, port, q, wire, layout, inject):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "imm-tsf_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from immtsf import _lib
    from immtsf.train import FlagStep, FlatTrainer, shard_range
    model, fusion, tr0, batch = _setup(dev, 0.0)
    tr0.close()
    lo, hi = shard_range(8, rank, world)
    shard = {k: v[lo:hi].contiguous() for k, v in batch.items()}
    cnt = shard["mask_predicted_data"].reshape(-1, shard["mask_predicted_data"].shape[-1]).sum(0)
    dist.all_reduce(cnt)
    te = [model.te_scale.weight, model.te_scale.bias, model.te_periodic.weight, model.te_periodic.bias]
    kw = {}
    if layout == "phases":
        fb, names = fusion.grad_buckets(shard["tp_to_predict"].shape[1])
        buckets = fb + [list(model.parameters())]
        names = list(names) + ["backbone"]
        kw = {"adam_split": ([i for i, n in enumerate(names) if n.startswith("ttf")], [len(names) - 1], [0]), "backbone_buckets": [len(names) - 1]}
    else:
        buckets = [list(fusion.mmf.parameters()), list(fusion.ttf.parameters()), list(model.parameters())]
    tr = FlatTrainer(buckets, lr=1e-2, eps=1e-3, max_norm=0.05, sink_buckets=tuple(range(len(buckets))), sink_shared=te, overlap=True,
                     device_step=True, group=dist.group.WORLD, grad_wire=wire)
    fc = (shard["tp_to_predict"], shard["observed_data"], shard["observed_tp"], shard["observed_mask"])

    def text_fn():
        if layout != "blocks":      # the fused pair (proj_out folded into MMF's projection: its gradient leaves MMF's chain, MMF's bucket holds it)
            E, M, kv = fusion.text_side(shard["notes_embeddings"], shard["tau"], shard["tp_to_predict"])
            return (E, M) + tuple(kv)
        E, M = fusion.ttf(shard["notes_embeddings"], shard["tau"], shard["tp_to_predict"])      # one bucket per block: the blocks as written
        return (E, M) + tuple(fusion.mmf.project_kv(E))

    def head_fn(pred, E, M, kv, fold):
        return fusion.mmf.forward_loss(pred, E, M, shard["data_to_predict"], shard["mask_predicted_data"], cnt, kv=(kv, fold))

    st = FlagStep(tr, text_fn, lambda: model.forecasting(*fc).contiguous(), head_fn, timeout_ms=20 if inject else 50, **kw)
    assert st.dist and len(st.segments) >= (2 if layout == "blocks" else 3)
    if not inject:
        for _ in range(3):
            st()
        tr.flush()
        torch.cuda.synchronize()
        st.check()
        if rank == 0:
            q.put((_named_flat(model, fusion).cpu().numpy(), len(st.segments), [g["branch"] for g in st.segments]))
    else:
        for _ in range(2):
            st()
        tr.flush()
        torch.cuda.synchronize()
        st.check()
        p2 = tr.gather(tr.flat_param).clone()
        keep = st.segments[0]["flags"][0]
        if rank == 1:
            st.segments[0]["flags"][0] = st.flags.data_ptr() + 4 * 47   # a word nobody bumps: the communication stream's spin gives up
        st()
        tr.flush()
        torch.cuda.synchronize()
        p3 = tr.gather(tr.flat_param).clone()
        raised = False
        try:
            st.check()
        except _lib.ImmtsfError:
            raised = True
        own = bool(st.timed_out())
        st.segments[0]["flags"][0] = keep
        st.clear_error()
        st()
        tr.flush()
        torch.cuda.synchronize()
        st.check()
        p4 = tr.gather(tr.flat_param).clone()
        q.put((rank, p2.cpu().numpy(), p3.cpu().numpy(), p4.cpu().numpy(), raised, own))
    dist.barrier()
    dist.destroy_process_group()


def _spawn2(target, args):
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + tuple(args)) for r in range(2)]
    for p in procs:
        p.start()
    return q, procs


@pytest.mark.parametrize("layout", ["phases", "blocks"])
@pytest.mark.parametrize("wire,tol", [("fp32", 3e-4), ("bf16", 3e-2)])
def test_two_rank_flag_step_equals_single_process(wire, tol, layout):
    """bench.py's N > 1 default -- FlagStep with the bucketed all-reduce beside the backward, one graph per step, the optimizer at the
    head of the next replay -- on two ranks (half batches) trains like one process on the full batch: 3 steps, clip active (max_norm
    0.05).  SURVEY 8e."""
    dev = _dev()
    q, procs = _spawn2(_two_rank_flag_worker, (wire, layout))
    res = q.get(timeout=300)
    assert not (isinstance(res[0], str) and res[0] == "error"), res[-1]
    got, nseg, branches = res
    got = torch.from_numpy(got)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if layout == "phases":
        # MMF's bucket and TTF's first two phases (one burst, one collective) leave the parameter branch, TTF's last phase the text
        # side, the backbone its own branch: the communication stream's order
        assert branches == ["P", "PP", "T", "B"], branches
    model, fusion, tr, batch = _setup_sinks(dev, 0.0)
    tr.max_norm = 0.05
    f = _loss_fn(model, fusion, batch)
    for _ in range(3):
        tr.zero_grad()
        f().backward()
        tr.sync_grads()
        tr.step()
    ref = _named_flat(model, fusion).cpu()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < tol, (err, nseg, branches)


def test_two_rank_flag_step_drops_a_timed_out_step_on_every_rank():
    """Cross-rank guard coherence (round-4 review): a device-flag wait that gives up on ONE rank (here: rank 1's communication stream
    spins on a word nobody bumps) must make EVERY rank drop that step -- the guard words ride with the step's last collective and the
    step decision reads their sum -- so the replicas stay bit-identical; check() raises on both; after clear_error() training goes on."""
    _dev()
    q, procs = _spawn2(_two_rank_flag_worker, ("bf16", "phases", True))
    res = {}
    for _ in range(2):
        r = q.get(timeout=300)
        assert r[0] != "error", r[-1]
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    import numpy as np
    for rank in (0, 1):
        p2, p3, p4, raised, own = res[rank]
        assert np.array_equal(p2, p3), f"rank {rank} applied a step the other rank's time-out should have dropped"
        assert raised, f"check() did not raise on rank {rank}"
        assert own == (rank == 1)                      # only rank 1's own guard word was set: rank 0 learnt it from the collective
        assert not np.array_equal(p3, p4)              # training continues after clear_error()
    for i in range(3):
        assert np.array_equal(res[0][i], res[1][i]), "replicas diverged"


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("tail", [True, False])
def test_t2v_backward_in_phases_equals_one_call(precision, tol, tail):
    """TTF_T2V_XAttn's backward as three calls (immtsf_fusion_cfg.bwd_phase: what a data-parallel step uses to hand finished gradient
    buckets to the all-reduce early) writes the gradients the single call writes, and every bucket's hook fires behind the phase that
    completes it -- in order A, B, C.  tail=False: the form in which the consumer owns proj_out."""
    dev = _dev()
    from fusions.TTF_T2V_XAttn import TTF_T2V_XAttn
    from fusions.load_llm import register_d_model
    from immtsf import config
    from immtsf.train import FlatTrainer
    register_d_model("TOY64", 64)
    config.precision = precision
    config.nan_check = "deferred"
    try:
        def build():
            torch.manual_seed(11)
            return TTF_T2V_XAttn("TOY64", 6, n_heads_fusion=2, dropout=0.0, d_txt=32).to(dev).train()
        g = torch.Generator().manual_seed(2)
        B, N, T = 12, 9, 7
        notes = torch.randn(B, N, 64, generator=g)
        lengths = torch.randint(1, N + 1, (B,), generator=g)
        tau = torch.rand(B, N, generator=g) * 24
        for b in range(B):
            notes[b, int(lengths[b]):] = 0
            tau[b, int(lengths[b]):] = 0
        notes, tau = notes.to(dev), tau.to(dev)
        t_hat = torch.sort(torch.rand(B, T, generator=g), 1).values.to(dev)
        up = torch.randn(B, T, 32, generator=g).to(dev)

        def run(m, tr):
            tr.zero_grad()
            E, _ = m(notes, tau, t_hat, tail=tail)
            (E * up).sum().backward()
            torch.cuda.synchronize()
            return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

        a = build()
        ta = FlatTrainer([list(a.parameters())], sink_buckets=(0,), overlap=False)
        ga = run(a, ta)
        b = build()
        tb = FlatTrainer(b.grad_phases(tail=tail) + ([[b.proj_out.weight, b.proj_out.bias]] if not tail else []), sink_buckets=(0, 1, 2),
                         overlap=False)
        fired = []
        for i, ps in enumerate(tb.buckets[:3]):
            ps[0]._immtsf_bwd_hook = (lambda i=i: fired.append(i))
        gb = run(b, tb)
        assert fired == [0, 1, 2], fired
        for k in ga:
            if not tail and k.startswith("proj_out"):
                continue
            ref = ga[k]
            err = float((gb[k] - ref).abs().max() / max(float(ref.abs().max()), 1e-6))
            assert err < tol, (k, err)
        ta.close(); tb.close()
    finally:
        config.precision = "fp32"


def test_adjacent_projection_weights_take_the_single_gemm_path():
    """inside FlatTrainer's flat buffer proj_k.weight and proj_v.weight of MMF_XAttn_Add are adjacent, which lets the
    backward form dE = [dK0 | dV0] [W_k ; W_v] as ONE GEMM: same gradients as with separately allocated parameters."""
    dev = _dev()
    from fusions.MMF_XAttn_Add import MMF_XAttn_Add
    from immtsf import config
    from immtsf.train import FlatTrainer
    config.precision = "fp32"
    torch.manual_seed(3)
    a = MMF_XAttn_Add(d_txt=32, C=5, d_attn=32, n_heads_fusion=2, dropout=0.0, kappa=0.5).to(dev).train()
    b = MMF_XAttn_Add(d_txt=32, C=5, d_attn=32, n_heads_fusion=2, dropout=0.0, kappa=0.5).to(dev).train()
    b.load_state_dict(a.state_dict())
    tr = FlatTrainer([list(b.parameters())], sink_buckets=(), overlap=False)
    assert b.proj_v.weight.data_ptr() == b.proj_k.weight.data_ptr() + 4 * b.proj_k.weight.numel()
    g = torch.Generator().manual_seed(1)
    Y, E = torch.randn(6, 7, 5, generator=g).to(dev), torch.randn(6, 7, 32, generator=g).to(dev)
    M = torch.tensor([1, 1, 0, 1, 1, 1], dtype=torch.bool, device=dev).view(6, 1)
    up = torch.randn(6, 7, 5, generator=g).to(dev)
    Ea, Eb = E.clone().requires_grad_(True), E.clone().requires_grad_(True)
    (a(Y, Ea, M) * up).sum().backward()
    tr.zero_grad()
    (b(Y, Eb, M) * up).sum().backward()
    torch.cuda.synchronize()
    assert float((Ea.grad - Eb.grad).abs().max() / Ea.grad.abs().max()) < 1e-5
    for (k, p), q in zip(a.named_parameters(), b.parameters()):
        assert float((p.grad - q.grad).abs().max() / max(float(p.grad.abs().max()), 1e-6)) < 1e-5, k
    tr.close()


def _setup_sinks(dev, dropout, seed=0):
    """like _setup, with the backbone bucket on gradient sinks too; the time-embedding parameters, shared by the patch encoder
    and the decoder, are accumulating sinks (both backward ops add into the zero-filled slice)"""
    model, fusion, tr, batch = _setup(dev, dropout, seed)
    from immtsf.train import FlatTrainer
    tr.close()
    te = [model.te_scale.weight, model.te_scale.bias, model.te_periodic.weight, model.te_periodic.bias]
    tr2 = FlatTrainer([list(fusion.mmf.parameters()), list(fusion.ttf.parameters()), list(model.parameters())],
                      lr=1e-2, eps=1e-3, max_norm=1.0, sink_buckets=(0, 1, 2), sink_shared=te, overlap=False, device_step=True)
    return model, fusion, tr2, batch


def test_two_writers_of_an_undeclared_sink_raise():
    """tPatchGNN's LearnableTE parameters get gradients from the patch encoder AND the decoder (models/tPatchGNN.py:176-195, :283-295): on
    gradient sinks both ops must accumulate, which they only do for parameters declared `sink_shared` -- a backbone bucket on sinks
    without the declaration raises in the forward instead of silently keeping one op's share"""
    dev = _dev()
    from immtsf import _lib
    from immtsf.train import FlatTrainer
    model, fusion, tr, batch = _setup(dev, 0.0)
    tr.close()
    tr2 = FlatTrainer([list(fusion.mmf.parameters()), list(fusion.ttf.parameters()), list(model.parameters())],
                      lr=1e-2, eps=1e-3, max_norm=1.0, sink_buckets=(0, 1, 2), overlap=False, device_step=True)
    try:
        with pytest.raises(_lib.ImmtsfError, match="sink_shared"):
            _loss_fn(model, fusion, batch)()
    finally:
        tr2.close()


def test_backbone_gradient_sinks_equal_autograd_accumulation():
    """the backbone ops writing their parameter gradients straight into the flat buffer (no fills, no per-parameter adds,
    no collection copy) must give the flat gradient autograd + collect_grads gives"""
    dev = _dev()
    model, fusion, tr, batch = _setup(dev, 0.0)
    tr.zero_grad()
    _loss_fn(model, fusion, batch)().backward()
    tr.collect_grads()
    ref = tr.gather(tr.flat_grad).clone()
    tr.close()
    model, fusion, tr, batch = _setup_sinks(dev, 0.0)
    assert len(tr._autograd_owned) == 0      # every parameter is a sink; the four shared time-embedding ones accumulate
    tr.zero_grad()
    _loss_fn(model, fusion, batch)().backward()
    tr.collect_grads()
    got = tr.gather(tr.flat_grad)
    torch.cuda.synchronize()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
    tr.close()


@pytest.mark.parametrize("kind", ["phased", "flags", "flags_fused_loss"])
def test_phased_step_trains_like_eager(kind):
    """immtsf.train.PhasedStep (six single-stream graphs on two streams, the query half's parameter gradients deferred
    behind the text-side backward) and immtsf.train.FlagStep (the same decomposition as ONE graph whose branches synchronise
    through device flags instead of graph edges, csrc/sync.hip) must train exactly like the eager step; no spin may time out"""
    dev = _dev()
    from immtsf.ops import masked_mse
    from immtsf.train import FlagStep, PhasedStep
    steps = 4
    model, fusion, tr, batch = _setup_sinks(dev, 0.0)
    f = _loss_fn(model, fusion, batch)
    for _ in range(steps):
        tr.zero_grad()
        f().backward()
        tr.sync_grads()
        tr.step()
    ref = tr.flat_param.clone()
    tr.close()
    model, fusion, tr, batch = _setup_sinks(dev, 0.0)
    fc = (batch["tp_to_predict"], batch["observed_data"], batch["observed_tp"], batch["observed_mask"])

    def text_fn():
        E, M = fusion.ttf(batch["notes_embeddings"], batch["tau"], batch["tp_to_predict"])
        return (E, M) + tuple(fusion.mmf.project_kv(E))

    cnt = batch["mask_predicted_data"].reshape(-1, batch["mask_predicted_data"].shape[-1]).sum(0)

    def head_fn(pred, E, M, kv, fold):
        if kind == "flags_fused_loss":       # MMF_XAttn_Add's head + the loss + their backward as one launch (csrc/xrank.hip)
            return fusion.mmf.forward_loss(pred, E, M, batch["data_to_predict"], batch["mask_predicted_data"], cnt, kv=(kv, fold))
        return masked_mse(fusion.mmf(pred, E, M, kv=(kv, fold)), batch["data_to_predict"], batch["mask_predicted_data"])

    from immtsf import config
    config.head_dy_ptr = None
    # (fused loss: a contiguous forecast, as the fused decoder of the benchmark configuration returns it -- the head then publishes
    # the "dY is ready" flag itself, in the middle of its kernel, and FlagStep leaves its own flag_set out)
    bb = (lambda: model.forecasting(*fc).contiguous()) if kind == "flags_fused_loss" else (lambda: model.forecasting(*fc))
    st = (PhasedStep if kind == "phased" else FlagStep)(tr, text_fn, bb, head_fn)
    assert (config.head_dy_ptr is not None) == (kind == "flags_fused_loss")
    losses = [float(st().detach()) for _ in range(steps)]
    tr.flush()               # (FlagStep: the last step's clip + Adam would run at the head of the next replay)
    torch.cuda.synchronize()
    if kind != "phased":
        assert not st.timed_out()
    assert losses[-1] < losses[0]
    err = float((tr.flat_param - ref).abs().max() / ref.abs().max())
    assert err < 2e-4, err
    tr.close()


def test_split_k_gemm_zero_fill_survives_graph_replay():
    """A split-K product (few output tiles, long reduction: 128 x 16 x 2048 in exact fp32) zero-fills its output in front of the atomic
    accumulation.  As a hipMemsetAsync node inside a captured hipGraph that fill left the output dirty on the second and later
    replays (ROCm 7.2): replay 0 exact, replays 1.. off by 1.0 -- found through the graph-cached drop-in seam, whose transformer-layer
    GEMMs are not pre-zeroed gradient sinks.  Every zero-fill of the library is a kernel now; this pins it."""
    dev = _dev()
    from immtsf import _lib
    lib = _lib.load()
    torch.manual_seed(0)
    M, N, K = 128, 16, 2048
    A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    Cc = torch.empty(M, N, device=dev)

    def run():
        _lib.check(lib.immtsf_gemm(0, 0, _lib.ptr(A), K, _lib.ptr(B), K, _lib.ptr(Cc), N, None, M, N, K, 1.0, 0, 0, _lib.stream_ptr()), "gemm")
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    ref = A @ B.t()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    for i in range(4):
        g.replay()
        torch.cuda.synchronize()
        assert float((Cc - ref).abs().max()) < 1e-3, i


def test_dropin_seam_graph_replay_equals_eager():
    """The zero-edit seam of an unmodified main.py (lib.evaluation.compute_all_losses -> loss.backward() -> clip_grad_norm_ ->
    torch.optim.Adam.step(), main.py:1093-1101) served from a replayed hipGraph (IMMTSF_NAN_CHECK=deferred; the first call of a batch
    shape is eager, the second captures, later ones replay) trains exactly like the eager seam; gradients ACCUMULATE into .grad like
    autograd's when the caller does not zero them; a second batch of the same shape goes through the same graph."""
    dev = _dev()
    from immtsf import config
    from lib import evaluation as ev
    from lib.evaluation import compute_all_losses
    import bench

    def train(seam):
        ev._graphs.clear(); ev._seen.clear()
        model, fusion, _, batch = _setup(dev, 0.0, trainer=False)          # (plain parameters: what an unmodified main.py has)
        cpu2, _ = bench.synth_batch(6, 8)
        batch2 = {k: v.to(dev) for k, v in cpu2.items()}
        batch2["notes_embeddings"] = batch2["notes_embeddings"][..., :48].contiguous()
        params = [p for p in list(model.parameters()) + list(fusion.parameters())]
        opt = torch.optim.Adam(params, lr=1e-2, eps=1e-3)
        config.seam_graph = seam
        losses = []
        for i in range(6):
            opt.zero_grad()
            res = compute_all_losses(model, fusion, batch if i % 2 == 0 else batch2)
            res["loss"].backward()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            losses.append(float(res["loss"]))
        # accumulation: two backward passes without zero_grad
        opt.zero_grad()
        compute_all_losses(model, fusion, batch)["loss"].backward()
        g1 = [p.grad.clone() for p in params if p.grad is not None]
        compute_all_losses(model, fusion, batch)["loss"].backward()
        g2 = [p.grad.clone() for p in params if p.grad is not None]
        torch.cuda.synchronize()
        n_graphs = sum(len(v) for v in ev._graphs.values())
        return torch.cat([p.detach().reshape(-1) for p in params]), losses, g1, g2, n_graphs

    old = config.seam_graph
    try:
        p_e, l_e, _, _, n_e = train(False)
        p_g, l_g, g1, g2, n_g = train(True)
    finally:
        config.seam_graph = old
        ev._graphs.clear(); ev._seen.clear()
    assert n_e == 0 and n_g == 1              # one shape -> one graph, used by both batches
    import gc
    gc.collect()
    assert sum(len(v) for v in ev._graphs.values()) == 0          # the cache died with the models (weak keys): no stale graph for a recycled id
    for a, b in zip(l_e, l_g):
        assert abs(a - b) <= 1e-5 * abs(a), (l_e, l_g)
    assert float((p_e - p_g).abs().max() / p_e.abs().max()) < 2e-4
    for a, b in zip(g1, g2):
        # (1e-4: the time-embedding scalars are atomic sums of cancelling terms -- their order differs between the two passes)
        assert float((b - 2 * a).abs().max()) <= 1e-4 * max(float(a.abs().max()), 1e-6)


@pytest.mark.parametrize("seam", [False, True])
def test_optim_shim_trains_like_torch_adam(seam):
    """immtsf.optim (what `optim.Adam(...)` / `clip_grad_norm_(...)` of an unmodified main.py resolve to once lib.evaluation is imported,
    main.py:1024,1098-1101): FusedAdam on flat buffers + the clip folded into its step train like torch.optim.Adam + torch's clip -- with
    weight decay, over the eager seam and the graph-replayed one; state_dict() / load_state_dict() keep torch's format (a checkpoint of
    one loads into the other); anything FusedAdam does not take falls back to torch's Adam."""
    dev = _dev()
    from immtsf import config, optim
    from lib import evaluation as ev
    from lib.evaluation import compute_all_losses
    old = config.seam_graph
    config.seam_graph = seam

    def train(fused, steps=5, resume=None):
        ev._graphs.clear(); ev._seen.clear()
        model, fusion, _, batch = _setup(dev, 0.0, trainer=False)
        params = [p for p in list(model.parameters()) + list(fusion.parameters())]
        mk = optim.Adam if fused else optim._torch_adam
        clip = optim.clip_grad_norm_ if fused else optim._torch_clip
        opt = mk(params, lr=1e-2, eps=1e-3, weight_decay=1e-3)
        assert isinstance(opt, optim.FusedAdam) == fused
        if resume is not None:
            opt.load_state_dict(resume[0])
            with torch.no_grad():
                for p, v in zip(params, resume[1]):
                    p.copy_(v)
        norms = []
        for _ in range(steps):
            opt.zero_grad()
            compute_all_losses(model, fusion, batch)["loss"].backward()
            norms.append(float(clip(params, 0.05)))
            opt.step()
        torch.cuda.synchronize()
        return torch.cat([p.detach().reshape(-1) for p in params]).clone(), norms, opt.state_dict(), [p.detach().clone() for p in params]

    try:
        p_t, n_t, sd_t, w_t = train(False)
        p_f, n_f, sd_f, w_f = train(True)
        assert float((p_t - p_f).abs().max() / p_t.abs().max()) < 2e-4
        for a, b in zip(n_t, n_f):
            assert abs(a - b) <= 1e-4 * abs(a), (n_t, n_f)                 # the norm clip_grad_norm_ returns
        # a torch checkpoint resumes in the fused optimizer and the other way round: 2 more steps land on the same parameters
        p_a = train(True, steps=2, resume=(sd_t, w_t))[0]
        p_b = train(False, steps=2, resume=(sd_f, w_f))[0]
        assert float((p_a - p_b).abs().max() / p_a.abs().max()) < 2e-4
        assert isinstance(optim.Adam([torch.nn.Parameter(torch.zeros(3))]), optim._torch_adam)              # CPU parameter: torch's
        assert isinstance(optim.Adam([torch.nn.Parameter(torch.zeros(3, device=dev))], amsgrad=True), optim._torch_adam)
    finally:
        config.seam_graph = old
        ev._graphs.clear(); ev._seen.clear()


def test_deferred_nan_guards_raise_at_the_next_call():
    """immtsf.config.nan_check = "deferred" (the default): the reference's ValueErrors (fusions/TTF_T2V_XAttn.py:116-117,
    lib/evaluation.py:158-160) are raised by the NEXT compute_all_losses() call instead of inside the step that met the NaN -- no host
    sync inside the step; a clean step raises nothing."""
    dev = _dev()
    from immtsf import config
    from lib import evaluation as ev
    from lib.evaluation import compute_all_losses
    config.nan_check = "deferred"
    config.seam_graph = False
    try:
        model, fusion, _, batch = _setup(dev, 0.0, trainer=False)
        ev._probe.clear()
        compute_all_losses(model, fusion, batch)["loss"].backward()
        torch.cuda.synchronize()
        compute_all_losses(model, fusion, batch)                               # clean: nothing raised
        bad = dict(batch)
        bad["notes_embeddings"] = batch["notes_embeddings"].clone()
        bad["notes_embeddings"][1, 0, 3] = float("nan")
        torch.cuda.synchronize()
        compute_all_losses(model, fusion, bad)                                  # the step itself goes through ...
        torch.cuda.synchronize()                                                # (main.py's own loss.item())
        with pytest.raises(ValueError, match="Input embeddings V contain NaN"):
            compute_all_losses(model, fusion, batch)                            # ... the next call raises
        torch.cuda.synchronize()
        bad2 = dict(batch)
        bad2["data_to_predict"] = batch["data_to_predict"].clone()
        bad2["data_to_predict"][0, 0, 0] = float("nan")
        bad2["mask_predicted_data"] = batch["mask_predicted_data"].clone()
        bad2["mask_predicted_data"][0, 0, 0] = 1.0
        ev._probe.clear()
        compute_all_losses(model, fusion, bad2)
        torch.cuda.synchronize()
        with pytest.raises(ValueError, match="MSE is NaN"):
            compute_all_losses(model, fusion, batch)
    finally:
        config.seam_graph = True
        ev._probe.clear()


def test_dropin_seam_graph_guards():
    """the seam graph's guards (round-4 advisor findings): a loss whose graph was replayed again before its backward() raises instead of
    handing on the other batch's gradients; a parameter frozen after the capture gets a graph of its own (the key holds every parameter's
    storage address and requires_grad); building a graph leaves module buffers and the dropout counter as they were."""
    dev = _dev()
    from immtsf import config
    from lib import evaluation as ev
    from lib.evaluation import compute_all_losses
    old = config.seam_graph
    config.seam_graph = True
    try:
        model, fusion, _, batch = _setup(dev, 0.1, trainer=False)
        drop_dev = config.enable_device_counters(dev)[1]
        compute_all_losses(model, fusion, batch)["loss"].backward()            # first sighting: eager
        d0 = int(drop_dev.item())
        bufs0 = [b.clone() for b in list(model.buffers()) + list(fusion.buffers())]
        l1 = compute_all_losses(model, fusion, batch)["loss"]                     # second: captured (two warm-up runs + capture) and replayed
        assert sum(len(v) for v in ev._graphs.values()) == 1
        assert int(drop_dev.item()) == d0 + 1                                     # one step's worth, not three
        for a, b in zip(bufs0, list(model.buffers()) + list(fusion.buffers())):
            if not b.dtype.is_floating_point:                                     # (e.g. BatchNorm's num_batches_tracked: ONE step, not three)
                assert int((b - a).abs().max()) <= 1
        l2 = compute_all_losses(model, fusion, batch)["loss"]
        with pytest.raises(RuntimeError, match="before this loss's backward"):
            l1.backward()
        l2.backward()
        p0 = next(fusion.mmf.parameters())
        p0.requires_grad_(False)
        compute_all_losses(model, fusion, batch)["loss"].backward()              # new key: eager
        compute_all_losses(model, fusion, batch)["loss"].backward()              # ... then a second graph
        assert sum(len(v) for v in ev._graphs.values()) == 2
    finally:
        config.seam_graph = old
        ev._graphs.clear(); ev._seen.clear()


def test_load_state_dict_refreshes_the_bf16_twin():
    """bf16 mode reads the GEMM weights from FlatTrainer's bf16 twin: after trainer.watch(module), module.load_state_dict
    must refresh it (the forward then uses the NEW weights); without the hook the twin would be stale"""
    dev = _dev()
    from immtsf import config
    model, fusion, tr, batch = _setup(dev, 0.0)
    assert tr.flat_twin is not None
    tr.watch(model, fusion)
    config.precision = "bf16"
    try:
        f = _loss_fn(model, fusion, batch)
        with torch.no_grad():
            l0 = float(f())
        sd = {k: v.clone() for k, v in fusion.state_dict().items()}
        for k in sd:
            if sd[k].dtype.is_floating_point and sd[k].dim() == 2:
                sd[k] = sd[k] * 1.5
        fusion.load_state_dict(sd)
        with torch.no_grad():
            l1 = float(f())
        # reference point: the same weights in a trainer built AFTER the load (twin derived from them at construction)
        tr.refresh_twins()
        with torch.no_grad():
            l2 = float(f())
        torch.cuda.synchronize()
    finally:
        config.precision = "fp32"
        tr.close()
    assert abs(l1 - l0) > 1e-3 * abs(l0), (l0, l1)          # the new weights are in effect
    assert l1 == l2, (l1, l2)                               # and the hook left nothing stale


@pytest.mark.parametrize("engine", ["flags", "flags_packed", "flags_fold", "graphed", "graphed_fold"])
@pytest.mark.parametrize("precision,tol_loss,tol_param,tol_delta", [("fp32", 1e-4, 3e-4, 5e-3), ("bf16", 3e-2, 3e-3, 2.5e-1)])
def test_cfg2_step_vs_oracle(precision, tol_loss, tol_param, tol_delta, engine):
    """The BENCHMARKED composition against the oracle, not against itself: three cfg2 training steps (tPatchGNN -> TTF_T2V_XAttn
    -> MMF_XAttn_Add -> masked MSE -> clip 1.0 -> Adam) at B = 64, d = 768 through bench.Workload + bench.build_step -- exactly what
    bench.py times: "flags" = immtsf.train.FlagStep (one hipGraph, three branches synchronised by device flags, the head publishing
    the dY flag mid-kernel, MMF_XAttn_Add's fold and parameter chain on the parameter branch, grouped weight gradients), "flags_packed"
    = the same with the notes handed over as PackedNotes (bench.py's default hand-over), "graphed" = GraphedStep (graph edges);
    "flags_fold" / "graphed_fold" = the same two engines with TTF_T2V_XAttn forced into its folded form (what the size rule picks from
    256 windows on; the scheduling hint between its backward and the patch encoder's is live) -- vs oracle/tpatchgnn_ref.py +
    oracle/fusion_ref.py +
    torch.optim.Adam on the CPU from identical weights (reference: lib/evaluation.py:72-164, main.py:1093-1101).  Dropout 0.
    Bars: loss of every step `tol_loss` relative; final parameters `tol_param` relative L2; the three-step UPDATE (p_final -
    p_init) `tol_delta` relative L2 -- in bf16 mode the update of weakly driven parameters carries the operands' 2^-9 rounding."""
    dev = _dev()
    sys.path.insert(0, ROOT)
    import bench
    from immtsf import config
    from oracle import fusion_ref as R
    from oracle import tpatchgnn_ref as TP
    old_drop = bench.P_DROP
    bench.P_DROP = 0.0
    config.nan_check = "deferred"       # no host syncs inside the captured step (what bench.py sets)
    old_form = config.t2v_form
    config.t2v_form = "fold" if engine.endswith("_fold") else "auto"
    try:
        w = bench.Workload("cfg2", dev, 64, precision, device_step=True, packed_notes=engine == "flags_packed")
        # Adam eps 1e-3 on both sides: with 1e-8 the sign-like first steps turn 1e-9 summation-order noise on gradients that are
        # zero in exact arithmetic (the softmax's key bias) into +-lr parameter differences
        w.trainer.eps = 1e-3
        a = bench.model_args("cpu")
        ref_model = TP.build(a).train()

        def no_dropout(model):          # the stock transformer layer of the backbone is built with torch's default dropout 0.1
            for mm in model.modules():
                if isinstance(mm, torch.nn.Dropout):
                    mm.p = 0.0
            for lyr in model.transformer_encoder:
                for l_ in lyr.layers:
                    l_.self_attn.dropout = 0.0
        no_dropout(w.model)
        no_dropout(ref_model)
        ref_model.load_state_dict({k: v.detach().cpu() for k, v in w.model.state_dict().items()})
        params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in w.fusion.state_dict().items()}
        p0 = torch.cat([v.detach().reshape(-1) for v in params.values()] + [q.detach().reshape(-1) for q in ref_model.parameters()])
        opt = torch.optim.Adam(list(ref_model.parameters()) + list(params.values()), lr=1e-3, eps=1e-3)
        b = w.cpu_batch
        ref_losses = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            pred = ref_model.forecasting(b["tp_to_predict"], b["observed_data"], b["observed_tp"], b["observed_mask"])
            out = R.fusion_forward("TTF_T2V_XAttn", "MMF_XAttn_Add", params, b["notes_embeddings"], b["tau"], b["tp_to_predict"], pred,
                                   H=bench.H, kappa=bench.KAPPA, expand_T=False)
            loss = R.masked_mse(b["data_to_predict"], out, b["mask_predicted_data"])
            loss.backward()
            torch.nn.utils.clip_grad_norm_(list(ref_model.parameters()) + list(params.values()), 1.0)
            opt.step()
            ref_losses.append(float(loss))
        eng = "graphed" if engine.startswith("graphed") else "flags"
        step, info = bench.build_step(w, eng)          # (warm-up / trial steps are undone)
        assert info["engine"] == eng and not info["flag_step_rejected"], info
        got_losses = [float(step()) for _ in range(3)]
        w.trainer.flush()                                 # (FlagStep applies step k's clip + Adam at the head of replay k + 1)
        torch.cuda.synchronize()
        if eng != "graphed":
            step.check()                                  # no spin gave up: no step was dropped
        for g_, r_ in zip(got_losses, ref_losses):
            assert abs(g_ - r_) <= tol_loss * abs(r_), (got_losses, ref_losses)
        sd_f = {k: v.detach().cpu() for k, v in w.fusion.state_dict().items()}
        names = [n for n, _ in ref_model.named_parameters()]
        sd_m = dict(w.model.named_parameters())
        p_gpu = torch.cat([sd_f[k].reshape(-1) for k in params] + [sd_m[n].detach().cpu().reshape(-1) for n in names])
        p_ref = torch.cat([v.detach().reshape(-1) for v in params.values()] + [q.detach().reshape(-1) for q in ref_model.parameters()])
        e_param = float((p_gpu - p_ref).norm() / p_ref.norm())
        e_delta = float(((p_gpu - p0) - (p_ref - p0)).norm() / (p_ref - p0).norm())
        # per-tensor update errors, worst first (diagnostics of a failure)
        worst, off = [], 0
        for k, v in list(params.items()) + [(n, q) for n, q in ref_model.named_parameters()]:
            n_ = v.numel()
            d_ref = (p_ref - p0)[off:off + n_]
            d_gpu = (p_gpu - p0)[off:off + n_]
            worst.append((float((d_gpu - d_ref).norm() / (p_ref - p0).norm()), float((d_gpu - d_ref).norm() / max(float(d_ref.norm()), 1e-12)), k))
            off += n_
        worst.sort(reverse=True)
        assert e_param < tol_param and e_delta < tol_delta, (e_param, e_delta, worst[:6])
        w.close()
    finally:
        bench.P_DROP = old_drop
        config.precision = "fp32"
        config.t2v_form = old_form


def test_timesnet_graph_step_trains_like_eager():
    """The cfg4 composition (TimesNet + TTF_RecAvg + MMF_XAttn_Add) replayed from a hipGraph -- possible because TimesNet's period
    selection, a host read in the reference (models/TimesNet.py:9-18), stays on the device (models/TimesNet.py fft_for_period_device,
    csrc/conv.hip conv2d_period_*) -- trains exactly like the eager step, also across a change of the batch that changes the selected
    periods: the SAME graph serves, the new periods reach its kernels as device numbers."""
    dev = _dev()
    sys.path.insert(0, ROOT)
    import bench
    from immtsf import config
    from immtsf.train import GraphedStep
    config.nan_check = "deferred"
    old_drop = bench.P_DROP
    bench.P_DROP = 0.0

    def run(graph):
        w = bench.Workload("cfg4", dev, 16, "fp32", device_step=True)
        w.trainer.eps = 1e-3
        for mm in w.model.modules():
            if isinstance(mm, torch.nn.Dropout):
                mm.p = 0.0
        step = GraphedStep(w.trainer, w.loss_fn) if graph else w.eager_step
        b = w.batch
        losses = []
        for i in range(8):
            if i == 4:                         # another signal: another spectrum, another top-k
                L = b["observed_data"].shape[1]
                t = torch.arange(L, device=dev, dtype=torch.float32).view(1, L, 1)
                b["observed_data"].copy_(torch.sin(t * 2.3) * 3.0 + torch.cos(t * 0.37) + 0.01 * b["observed_data"])
            losses.append(float(step()))
        torch.cuda.synchronize()
        p = torch.cat([q.detach().reshape(-1) for q in list(w.model.parameters()) + list(w.fusion.parameters())]).clone()
        w.close()
        return p, losses

    try:
        p_e, l_e = run(False)
        p_g, l_g = run(True)
    finally:
        bench.P_DROP = old_drop
        config.precision = "fp32"
    for a, b_ in zip(l_e, l_g):
        assert abs(a - b_) <= 1e-4 * abs(a), (l_e, l_g)
    err = float((p_e - p_g).abs().max() / p_e.abs().max())
    assert err < 2e-4, err


@pytest.mark.parametrize("cfg,precision", [("cfg3", "fp32"), ("cfg3", "bf16"), ("cfg4", "bf16")])
def test_cfg3_flag_step_trains_like_eager(cfg, precision):
    """The cfg3 composition (PatchTST + TTF_T2V_XAttn + MMF_GR_Add) on immtsf.train.FlagStep -- possible since MMF_GR_Add has a text-only
    half and a one-launch head (fusions/MMF_GR_Add.py project_kv / forward_loss, csrc/gr_train.hip), with the weight gradients of
    PatchTST's large linear layers deferred to the parameter branch in bf16 mode (immtsf.ops.LinearBf16Fn) -- trains like the eager
    step: the same losses, the same parameters after eight steps.  cfg4 in bf16 mode: TimesNet's Inception convolutions in the implicit
    form with their kernel gradients on the parameter branch (immtsf.ops.InceptionPeriodsFn).  reference: main.py:1093-1101 over
    models/PatchTST.py, models/TimesNet.py, fusions/MMF_GR_Add.py:31-61."""
    dev = _dev()
    sys.path.insert(0, ROOT)
    import bench
    from immtsf import config
    config.nan_check = "deferred"
    old_drop = bench.P_DROP
    bench.P_DROP = 0.0

    def run(flags):
        w = bench.Workload(cfg, dev, 16, precision, device_step=True)
        w.trainer.eps = 1e-3
        for mm in w.model.modules():
            if isinstance(mm, torch.nn.Dropout):
                mm.p = 0.0
        if flags:
            step = bench.flag_step(w)
            assert step is not None, "FlagStep was rejected for the %s workload" % cfg
        else:
            step = w.eager_step
        losses = [float(step()) for _ in range(8)]
        if flags:
            step.flush()
        torch.cuda.synchronize()
        p = torch.cat([q.detach().reshape(-1) for q in list(w.model.parameters()) + list(w.fusion.parameters())]).clone()
        w.close()
        return p, losses

    try:
        p_e, l_e = run(False)
        p_g, l_g = run(True)
    finally:
        bench.P_DROP = old_drop
        config.precision = "fp32"
    tol = 1e-4 if precision == "fp32" else 2e-2
    for a, b_ in zip(l_e, l_g):
        assert abs(a - b_) <= tol * abs(a), (l_e, l_g)
    err = float((p_e - p_g).abs().max() / p_e.abs().max())
    assert err < (2e-4 if precision == "fp32" else 3e-2), err


def _cfg2_reference(w, bench):
    """the oracle side of a cfg2 workload: (reference backbone, fusion parameters as leaf tensors, the CPU batch), from w's weights"""
    from oracle import tpatchgnn_ref as TP
    ref_model = TP.build(bench.model_args("cpu")).train()

    def no_dropout(model):          # the stock transformer layer of the backbone is built with torch's default dropout 0.1
        for mm in model.modules():
            if isinstance(mm, torch.nn.Dropout):
                mm.p = 0.0
        for lyr in model.transformer_encoder:
            for l_ in lyr.layers:
                l_.self_attn.dropout = 0.0
    no_dropout(w.model)
    no_dropout(ref_model)
    ref_model.load_state_dict({k: v.detach().cpu() for k, v in w.model.state_dict().items()})
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in w.fusion.state_dict().items()}
    return ref_model, params, w.cpu_batch


def test_cfg2_flag_step_gradient_vs_oracle_elementwise():
    """The benchmarked engine's PRE-ADAM gradient against the oracle's, element by element, at the default Adam eps (round-4 review:
    the three-step parameter comparison needs eps = 1e-3 to keep Adam's sign-like first steps from amplifying 1e-9 noise -- this is the
    evidence that does not): one FlagStep replay of the cfg2 workload (fp32 parity mode, dropout 0) leaves the step's gradient in the
    flat buffer (clip + Adam run at the head of the NEXT replay); every parameter's slice is compared with autograd's gradient of the
    oracle (oracle/tpatchgnn_ref.py + oracle/fusion_ref.py) at 2e-4 of the tensor's largest entry (north_star: 1e-4 outputs; gradients
    of d x d x 2048-row reductions carry the fp32 summation order)."""
    dev = _dev()
    sys.path.insert(0, ROOT)
    import bench
    from immtsf import config
    from oracle import fusion_ref as R
    old_drop = bench.P_DROP
    bench.P_DROP = 0.0
    config.nan_check = "deferred"
    try:
        w = bench.Workload("cfg2", dev, 64, "fp32", device_step=True, packed_notes=True)
        ref_model, params, b = _cfg2_reference(w, bench)
        pred = ref_model.forecasting(b["tp_to_predict"], b["observed_data"], b["observed_tp"], b["observed_mask"])
        out = R.fusion_forward("TTF_T2V_XAttn", "MMF_XAttn_Add", params, b["notes_embeddings"], b["tau"], b["tp_to_predict"], pred,
                               H=bench.H, kappa=bench.KAPPA, expand_T=False)
        loss = R.masked_mse(b["data_to_predict"], out, b["mask_predicted_data"])
        loss.backward()
        step, info = bench.build_step(w, "flags")
        assert info["engine"] == "flags" and not info["flag_step_rejected"], info
        got = float(step())
        torch.cuda.synchronize()
        step.check()
        assert abs(got - float(loss)) <= 1e-4 * abs(float(loss))
        worst = []
        named = [("f." + k, p_, params[k].grad) for k, p_ in w.fusion.named_parameters()] + \
                [("m." + k, p_, q.grad) for (k, p_), q in zip(w.model.named_parameters(), ref_model.parameters())]
        for k, p_, g_ref in named:
            g = p_.grad.detach().cpu()
            g_ref = torch.zeros_like(g) if g_ref is None else g_ref
            den = max(float(g_ref.abs().max()), 1e-7)
            worst.append((float((g - g_ref).abs().max()) / den, k, den))
        worst.sort(reverse=True)
        # tensors whose true gradient is zero in exact arithmetic (the softmax's key bias) are compared on the scale of the largest gradient
        gmax = max(d_ for _, _, d_ in worst)
        bad = [(e, k) for e, k, d_ in worst if e > 2e-4 and e * d_ > 2e-6 * gmax]
        assert not bad, bad[:6]
        w.close()
    finally:
        bench.P_DROP = old_drop
        config.precision = "fp32"


def test_cfg2_flag_step_with_dropout_vs_oracle():
    """The benchmarked engine WITH the benchmark's dropout (p = 0.1 at TTF_T2V_XAttn's attention weights and output, MMF_XAttn_Add's
    attention weights and output): the Philox keys advance on the device once per replay inside ONE hipGraph of three branches -- so
    the masks of every replay are exported per site (immtsf_dropout_mask with the key the replay used: the module's seed + the device
    counter) and fed to the oracle, step by step; loss of each of three steps at 1e-4, final parameters like test_cfg2_step_vs_oracle
    (fp32).  The round-4 review's gap: the dropout sites were only covered block by block, never through FlagStep."""
    dev = _dev()
    sys.path.insert(0, ROOT)
    import bench
    import numpy as np
    from immtsf import config, ops
    from oracle import fusion_ref as R
    pd = bench.P_DROP
    assert pd > 0
    config.nan_check = "deferred"
    try:
        w = bench.Workload("cfg2", dev, 64, "fp32", device_step=True, packed_notes=True)
        w.trainer.eps = 1e-3
        ref_model, params, b = _cfg2_reference(w, bench)
        plist = list(ref_model.parameters()) + list(params.values())
        p0 = torch.cat([v.detach().reshape(-1) for v in params.values()] + [q.detach().reshape(-1) for q in ref_model.parameters()])
        from immtsf import optim as _o
        opt = _o._torch_adam(plist, lr=1e-3, eps=1e-3)
        step, info = bench.build_step(w, "flags")
        assert info["engine"] == "flags" and not info["flag_step_rejected"], info
        B, N, T, Cc, d, H = 64, bench.N_MAX, bench.T_MAX, bench.C, bench.D_TXT, bench.H
        ttf, mmf = w.fusion.ttf, w.fusion.mmf

        def keep(seed, site, shape):
            return ops.dropout_keep_mask(seed & 0xFFFFFFFFFFFFFFFF, site, int(np.prod(shape)), pd, dev).cpu().view(*shape).float()
        for k in range(3):
            got = float(step())
            torch.cuda.synchronize()
            step.check()        # (a spin that timed out -- a stalled GPU -- drops the optimizer step by design: say so instead of a parity failure)
            ctr = int(w.trainer.drop_dev.item())               # the counter value this replay's kernels added to the modules' seeds
            drop = {"ttf": {"attn": keep(ttf.last_seed + ctr, 1, (B, T, H, N)), "out": keep(ttf.last_seed + ctr, 2, (B, T, d))},
                    "mmf": {"attn": keep(mmf.last_seed + ctr, 4, (B, H, T, T)), "out": keep(mmf.last_seed + ctr, 5, (B, T, Cc))}}
            opt.zero_grad(set_to_none=True)
            pred = ref_model.forecasting(b["tp_to_predict"], b["observed_data"], b["observed_tp"], b["observed_mask"])
            out = R.fusion_forward("TTF_T2V_XAttn", "MMF_XAttn_Add", params, b["notes_embeddings"], b["tau"], b["tp_to_predict"], pred,
                                   H=H, kappa=bench.KAPPA, drop=drop, p_drop=pd, expand_T=True)
            loss = R.masked_mse(b["data_to_predict"], out, b["mask_predicted_data"])
            loss.backward()
            _o._torch_clip(plist, 1.0)
            opt.step()
            assert abs(got - float(loss)) <= 1e-4 * abs(float(loss)), (k, got, float(loss))
        w.trainer.flush()
        torch.cuda.synchronize()
        step.check()
        sd_f = {k: v.detach().cpu() for k, v in w.fusion.state_dict().items()}
        names = [n for n, _ in ref_model.named_parameters()]
        sd_m = dict(w.model.named_parameters())
        p_gpu = torch.cat([sd_f[k].reshape(-1) for k in params] + [sd_m[n].detach().cpu().reshape(-1) for n in names])
        p_ref = torch.cat([v.detach().reshape(-1) for v in params.values()] + [q.detach().reshape(-1) for q in ref_model.parameters()])
        e_param = float((p_gpu - p_ref).norm() / p_ref.norm())
        e_delta = float(((p_gpu - p0) - (p_ref - p0)).norm() / (p_ref - p0).norm())
        assert e_param < 3e-4 and e_delta < 5e-3, (e_param, e_delta)
        w.close()
    finally:
        config.precision = "fp32"


@pytest.mark.gpu
def test_device_flags_and_their_trace():
    """immtsf_flag_set / immtsf_flag_wait across two streams (csrc/sync.hip) and the trace the flag kernels keep when asked
    (immtsf_flag_trace / immtsf_flag_trace_read, tools/flag_timeline.py): the consumer's wait is entered before the producer's set and
    left after it, on one clock; a wait on a flag nobody sets gives up and reports it."""
    import ctypes as C
    from immtsf import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    flags = torch.zeros(16, dtype=torch.int32, device=dev)
    fp = flags.data_ptr()
    x0 = torch.randn(2048, 2048, device=dev)
    torch.cuda.synchronize()
    # HIP streams are dealt round-robin onto a handful of hardware queues, and two streams on ONE queue run in order: a spin on b would
    # then sit in front of a's kernels until it gives up (seen at the end of the full suite, where the process has made dozens of
    # streams; FlagStep's own answer to that is the time-out + guard word + bench.flag_step's trial replays).  So: a fresh pair per
    # attempt, a short time-out, and the pair that does run concurrently is the one the assertions look at.
    for attempt in range(8):
        a, b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        flags.zero_()
        torch.cuda.synchronize()
        _lib.check(lib.immtsf_flag_trace(1), "flag_trace")
        _lib.check(lib.immtsf_flag_wait(fp, fp + 32, 300, b.cuda_stream), "flag_wait")        # spins until stream a gets there
        with torch.cuda.stream(a):
            x = x0
            for _ in range(4):
                x = x @ x * 1e-3
            _lib.check(lib.immtsf_flag_set(fp, a.cuda_stream), "flag_set")
        torch.cuda.synchronize()
        if int(flags[8]) == 0:
            break
    assert int(flags[8]) == 0, "no pair of streams out of 8 ran concurrently"
    buf = (C.c_int64 * (3 * 16))()
    n = lib.immtsf_flag_trace_read(buf, 16)
    assert n == 3
    ev = {int(buf[3 * i + 1]): int(buf[3 * i + 2]) for i in range(n)}
    assert all(int(buf[3 * i]) == fp for i in range(n))
    assert ev[1] <= ev[0] <= ev[2], ev                  # entered <= set <= left
    assert int(flags[0]) == 1 and int(flags[8]) == 0
    _lib.check(lib.immtsf_flags_clear(fp, 4, a.cuda_stream), "flags_clear")
    _lib.check(lib.immtsf_flag_wait(fp + 4, fp + 32, 1, a.cuda_stream), "flag_wait")        # nobody sets flag 1: gives up after 1 ms
    torch.cuda.synchronize()
    assert int(flags[8]) == 1
    assert lib.immtsf_flag_trace_read(buf, 16) == 6     # + clear, wait entered, wait left
    _lib.check(lib.immtsf_flag_trace(0), "flag_trace")
    _lib.check(lib.immtsf_flag_set(fp, a.cuda_stream), "flag_set")
    assert lib.immtsf_flag_trace_read(buf, 16) == 0
