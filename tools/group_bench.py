#!/usr/bin/env python3
"""The text side's grouped weight-gradient launch (csrc/gemm2.hip immtsf_launch_gemm2_group_tn) through the C ABI at the cfg2 shapes:
correctness of both tile sizes against an fp32 product of the same bf16 operands, then 20 launches per hipGraph.
usage: group_bench.py [M,N,K ...]   (default: 768,768,2048 1536,768,1117 768,1152,1117 768,768,1117)"""
import ctypes as C
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "imm-tsf_amd"))
import torch
from immtsf import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
ms = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(768, 768, 2048), (1536, 768, 1117), (768, 1152, 1117), (768, 768, 1117)]
torch.manual_seed(0)
As = [torch.randn(k, m, device=dev).bfloat16() for m, n, k in ms]
Bs = [torch.randn(k, n, device=dev).bfloat16() for m, n, k in ms]
Cs = [torch.zeros(m, n, device=dev) for m, n, k in ms]
ref = [a.float().t() @ b.float() for a, b in zip(As, Bs)]
k_ = len(ms)
vp = lambda ts: (C.c_void_p * k_)(*[t.data_ptr() for t in ts])      # noqa: E731
i32 = lambda xs: (C.c_int32 * k_)(*xs)                              # noqa: E731
pa, pb, pc = vp(As), vp(Bs), vp(Cs)
la, lb, lc = i32([m for m, n, k in ms]), i32([n for m, n, k in ms]), i32([n for m, n, k in ms])
mm, nn, kk = i32([m for m, n, k in ms]), i32([n for m, n, k in ms]), i32([k for m, n, k in ms])
flops = sum(2.0 * m * n * k for m, n, k in ms)
def one():
    _lib.check(lib.immtsf_gemm_bf16_group_tn(k_, pa, la, pb, lb, pc, lc, mm, nn, kk, _lib.stream_ptr()), "group")
def timed(n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): one()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): one()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
for code in (128, 64):
    lib.immtsf_debug_gemm2_config(1000 + code, 0, -1)
    one(); torch.cuda.synchronize()
    err = max(float((c - r).abs().max() / r.abs().max()) for c, r in zip(Cs, ref))
    us = timed()
    print(f"tile {code:3d}: {us:6.1f} us  {flops / us / 1e6:6.1f} TFLOP/s  max rel err {err:.2e}")
lib.immtsf_debug_gemm2_config(1000, 0, -1)
