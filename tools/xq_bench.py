#!/usr/bin/env python3
"""Q half of MMF_XAttn_Add's low-rank form as the training step runs it (forward + counted masked MSE + backward in one launch,
xrank_q_train_kernel): us per call at a given number of windows, back-to-back launches.  usage: xq_bench.py [windows] [T]
(run under `rocprofv3 --kernel-trace --stats` for the kernel's own duration)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "imm-tsf_amd")]
import torch  # noqa: E402


def main():
    from fusions.MMF_XAttn_Add import MMF_XAttn_Add
    from immtsf import config
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    C = 8
    dev = torch.device("cuda", 0)
    config.precision = os.environ.get("PREC", "bf16")
    config.manual_seed(1)
    torch.manual_seed(0)
    mmf = MMF_XAttn_Add(768, C, 768, n_heads_fusion=1, dropout=0.1, kappa=0.5).to(dev).train()
    E = torch.randn(B, T, 768, device=dev)
    with torch.no_grad():
        P, bHO = mmf.project_kv(E)
    P = P.detach().requires_grad_(True)
    bHO = bHO.detach().requires_grad_(True)
    Y = torch.randn(B, T, C, device=dev, requires_grad=True)
    M = torch.ones(B, 1, device=dev)
    truth = torch.randn(B, T, C, device=dev)
    mask = (torch.rand(B, T, C, device=dev) < 0.7).float()
    cnt = mask.reshape(-1, C).sum(0)

    def call():
        return mmf.forward_loss(Y, E, M, truth, mask, cnt, kv=(P, bHO))

    for _ in range(10):
        call()
    torch.cuda.synchronize()
    n = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    import ctypes as Ct
    from immtsf import _lib
    lib = _lib.load()
    if hasattr(lib, "immtsf_debug_xq_times"):
        buf = (Ct.c_longlong * 16)()
        lib.immtsf_debug_xq_times(buf)
        t = list(buf)
        print("ticks (10 ns) since entry of workgroup 0:", [t[i] - t[0] for i in range(11)])
    print("windows %d T %d: %.1f us per head+loss+backward call (back-to-back eager launches)" % (B, T, e0.elapsed_time(e1) * 1e3 / n))


if __name__ == "__main__":
    main()
