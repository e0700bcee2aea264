#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 counter collection).  usage: gemm_one.py layout M N K [variant] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imm-tsf_amd"))
import torch  # noqa: E402

from immtsf import _lib  # noqa: E402

lib = _lib.load()
layout, M, N, K = [int(a) for a in sys.argv[1:5]]
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = torch.device("cuda:0")
A = torch.randn((M, K) if layout < 2 else (K, M), device=dev)
B = torch.randn((N, K) if layout == 0 else (K, N), device=dev)
Cm = torch.empty(M, N, device=dev)
lib.immtsf_debug_gemm_config(variant, 0)
for _ in range(reps):
    _lib.check(lib.immtsf_gemm(layout, 1, _lib.ptr(A), A.shape[1], _lib.ptr(B), B.shape[1], _lib.ptr(Cm), N, None, M, N, K, 1.0, 0, 0,
                               _lib.stream_ptr()), "gemm")
torch.cuda.synchronize()
