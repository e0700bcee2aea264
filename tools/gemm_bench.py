#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM (through the C ABI) at the shapes the fusion step launches.
usage: python tools/gemm_bench.py [bf16|fp32] [sweep]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imm-tsf_amd"))
import torch  # noqa: E402

from immtsf import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
prec = 1 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else 0
sweep = len(sys.argv) > 2 and sys.argv[2] == "sweep"
SHAPES = [  # layout, M, N, K
    (0, 2048, 768, 768), (1, 2048, 768, 768), (2, 768, 768, 2048),
    (0, 1117, 768, 1152), (0, 1117, 1536, 768), (2, 1536, 768, 1117), (1, 1117, 1152, 768),
    (0, 2048, 8, 768), (0, 2048, 768, 8), (2, 8, 768, 2048), (0, 4096, 4096, 4096), (1, 4096, 4096, 4096), (2, 4096, 4096, 4096),
]
VARIANTS = {1: "64x64x64", 2: "64x64x128", 3: "128x64x64", 4: "128x128x64", 5: "32x64x64", 6: "64x128x64", 7: "64x64x32",
            11: "64x64x64w8", 14: "64x64x128w8", 15: "64x64x64spec", 16: "64x64x32spec"}


def bench(layout, M, N, K, variant, splitk):
    if layout == 0:
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    elif layout == 1:
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    Cm = torch.empty(M, N, device=dev)
    lda, ldb = A.shape[1], B.shape[1]
    lib.immtsf_debug_gemm_config(variant, splitk)

    def run():
        _lib.check(lib.immtsf_gemm(layout, prec, _lib.ptr(A), lda, _lib.ptr(B), ldb, _lib.ptr(Cm), N, None, M, N, K, 1.0, 0, 0,
                                   _lib.stream_ptr()), "gemm")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    ref = (A @ B.t()) if layout == 0 else (A @ B) if layout == 1 else (A.t() @ B)
    err = float((Cm - ref).abs().max() / ref.abs().max())
    lib.immtsf_debug_gemm_config(0, 0)
    return us, err


for layout, M, N, K in SHAPES:
    tag = f"{['NT','NN','TN'][layout]} M={M:5d} N={N:5d} K={K:5d}"
    us, err = bench(layout, M, N, K, 0, 0)
    print(f"{tag}  auto       {us:9.1f} us  {2.0*M*N*K/us/1e6:8.2f} TFLOP/s  relerr {err:.1e}", flush=True)
    us, err = bench(layout, M, N, K, 0x100, 0)
    print(f"{tag}  auto-noxcd {us:9.1f} us  {2.0*M*N*K/us/1e6:8.2f} TFLOP/s  relerr {err:.1e}", flush=True)
    if sweep and prec == 1:
        for v, name in VARIANTS.items():
            for sk in (1,):
                us, err = bench(layout, M, N, K, v, sk)
                print(f"{tag}  v{v} {name:10s} splitk={sk}  {us:9.1f} us  {2.0*M*N*K/us/1e6:8.2f} TFLOP/s  relerr {err:.1e}", flush=True)
