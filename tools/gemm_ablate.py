#!/usr/bin/env python3
"""Ablation of the GEMM main loop (which part of a K-step costs what): skip MFMA / skip global loads / skip LDS stores.
Timed inside one captured hipGraph of 50 launches so the CPU launch rate does not mask GPU time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imm-tsf_amd"))
import torch  # noqa: E402

from immtsf import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")


def timed(layout, M, N, K, cfgword, splitk=0, reps=50):
    if layout == 0:
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    elif layout == 1:
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    Cm = torch.empty(M, N, device=dev)
    lib.immtsf_debug_gemm_config(cfgword, splitk)

    def run():
        _lib.check(lib.immtsf_gemm(layout, 1, _lib.ptr(A), A.shape[1], _lib.ptr(B), B.shape[1], _lib.ptr(Cm), N, None, M, N, K, 1.0,
                                   0, 0, _lib.stream_ptr()), "gemm")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    lib.immtsf_debug_gemm_config(0, 0)
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


for layout, M, N, K in [(0, 2048, 768, 768), (1, 2048, 768, 768), (2, 768, 768, 2048), (0, 4096, 4096, 4096), (0, 512, 32, 64)]:
    for v in (1, 4, 5):
        if M * N < 100000 and v != 1:
            continue
        row = []
        for dbg, name in ((0, "full"), (1, "no-mfma"), (2, "no-gload"), (4, "no-ldsstore"), (7, "skeleton")):
            row.append(f"{name} {timed(layout, M, N, K, v | (dbg << 9), 1):7.1f}us")
        print(f"{['NT','NN','TN'][layout]} {M}x{N}x{K} v{v}: " + " | ".join(row), flush=True)
