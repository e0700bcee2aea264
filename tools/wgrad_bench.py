#!/usr/bin/env python3
"""weight-gradient GEMM (+ bias gradient) through immtsf_linear_backward, 50 launches per hipGraph"""
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "imm-tsf_amd"))
import torch
from immtsf import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
print("library:", os.path.basename(_lib.LIB_PATH))
for rows, N, K, bias in [(2048, 768, 768, True), (2048, 768, 768, False), (1117, 768, 1152, True), (1117, 1536, 768, True), (2048, 8, 768, True), (16384, 32, 32, True)]:
    x, dy = torch.randn(rows, K, device=dev), torch.randn(rows, N, device=dev)
    dW, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    def run():
        lib.immtsf_linear_backward(1, _lib.ptr(x), None, _lib.ptr(dy), rows, N, K, None, None, _lib.ptr(dW), _lib.ptr(db) if bias else None, 0, _lib.stream_ptr())
    for _ in range(3): run()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    ref_w, ref_b = dy.t() @ x, dy.sum(0)
    print(f"wgrad rows={rows} N={N} K={K} bias={bias}: {e0.elapsed_time(e1)/50*1e3:6.1f} us   errW {float((dW-ref_w).abs().max()/ref_w.abs().max()):.1e} errb {float((db-ref_b).abs().max()/ref_b.abs().max()) if bias else 0:.1e}")
