#!/usr/bin/env python3
"""Where the branches of immtsf.train.FlagStep wait for each other inside a replayed step: the flag kernels' own trace
(immtsf_flag_trace, 100 MHz device wall clock -- no profiler, nothing serialised).  Prints, per step, the time of every flag event
after the previous step's flags_clear, and how long each wait spun.  usage: flag_timeline.py [windows] [steps]
DIST=1: the data-parallel step on a 1-rank RCCL group (bucket announcements = when each gradient bucket is final); CFG=cfg3|cfg4: another
configuration that runs on FlagStep."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "imm-tsf_amd")]
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    from immtsf import _lib, config
    W = int(sys.argv[1]) if len(sys.argv) > 1 else bench.B_PER_GPU
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    config.nan_check = "deferred"
    config.manual_seed(1234)
    group = None
    if os.environ.get("DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        group = dist.group.WORLD
    prec = os.environ.get("PREC", "bf16")
    w = bench.Workload(os.environ.get("CFG", "cfg2"), dev, W, prec, packed_notes=os.environ.get("PADDED", "0") != "1", group=group,
                       wire="bf16" if prec == "bf16" else "fp32")
    st = bench.flag_step(w)
    if st is None:
        raise SystemExit("FlagStep not available for this workload")
    for _ in range(20):
        st()
    base = st.flags.data_ptr()
    names = {0: "backbone forward done", 4: "head: dY published", 8: "backbone backward done", 12: "fold done", 16: "parameter tail: inputs ready", 20: "parameter branch done",
             4 * st._COMM_DONE: "collectives of the step done"}
    names[4 * st._TTF] = "TTF phase B data path done"
    for i, g in enumerate(st.segments):
        names[g["flag"] - base] = "bucket %s final (%s, %.2f MB bf16)" % ("+".join(w.bucket_names[b] for b in g["buckets"]), g["branch"], (g["hi"] - g["lo"]) * 2 / 1e6)
    _lib.check(lib.immtsf_flag_trace(1), "flag_trace")
    for _ in range(steps):
        st()
    buf = (C.c_int64 * (3 * 1024))()
    n = lib.immtsf_flag_trace_read(buf, 1024)
    lib.immtsf_flag_trace(0)
    ev = sorted(((buf[3 * i + 2], buf[3 * i] - base, buf[3 * i + 1]) for i in range(n)))
    kinds = {0: "set", 1: "wait entered", 2: "wait left", 3: "flags cleared (optimizer follows)"}
    t_clear, step = None, 0
    entered = {}
    for t, off, kind in ev:
        if kind == 3:
            if t_clear is not None:
                print("  step %d: %.1f us from clear to clear" % (step, (t - t_clear) / 100.0))
            t_clear, step = t, step + 1
            continue
        if t_clear is None:
            continue
        rel = (t - t_clear) / 100.0
        extra = ""
        if kind == 1:
            entered[off] = t
        if kind == 2 and off in entered:
            extra = "  (spun %.1f us)" % ((t - entered.pop(off)) / 100.0)
        print("    +%7.1f us  %-26s %s%s" % (rel, names.get(off, "flag %d" % off), kinds[kind], extra))
    if group is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
