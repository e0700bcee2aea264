#!/usr/bin/env python3
"""tPatchGNN patch encoder (LearnableTE + TTCN) forward / backward through the C ABI, 20 launches per hipGraph.
usage: ttcn_bench.py [P L te_dim K]   (default: the benchmark shape 1024 32 10 31)"""
import ctypes as C
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "imm-tsf_amd"))
import torch
from immtsf import _lib
from immtsf.ops import TTCNParams, _struct
lib = _lib.load(); dev = torch.device("cuda:0"); ptr = _lib.ptr
P, L, te_dim, K = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (1024, 32, 10, 31)
F = 1 + te_dim
torch.manual_seed(0)
x, tt = torch.randn(P, L, device=dev), torch.rand(P, L, device=dev)
mask = (torch.arange(L, device=dev)[None, :] < torch.randint(0, L + 1, (P, 1), device=dev)).float()
shapes = [(1, 1), (1,), (1, te_dim - 1), (te_dim - 1,), (K, F), (K,), (K, K), (K,), (F * K, K), (F * K,), (K,)]
params = [torch.randn(*s, device=dev) * 0.3 for s in shapes]
grads = [torch.zeros_like(p) for p in params]
out, dout = torch.empty(P, K + 1, device=dev), torch.randn(P, K + 1, device=dev)
ws = torch.empty(lib.immtsf_ttcn_workspace_bytes(P, L, te_dim, K), dtype=torch.uint8, device=dev)
sc = torch.empty(lib.immtsf_ttcn_scratch_bytes(P, L, te_dim, K), dtype=torch.uint8, device=dev)
ps, gs = _struct(TTCNParams, params), _struct(TTCNParams, grads)
def fwd():
    assert lib.immtsf_ttcn_forward(P, L, te_dim, K, 1, ptr(x), ptr(tt), ptr(mask), C.byref(ps), ptr(out), K + 1, K, ptr(ws), ws.numel(), _lib.stream_ptr()) == 0
def bwd():
    assert lib.immtsf_ttcn_backward(P, L, te_dim, K, 1, ptr(x), ptr(tt), ptr(mask), C.byref(ps), ptr(out), ptr(dout), K + 1, ptr(ws), ws.numel(), ptr(sc), sc.numel(), C.byref(gs), 0, _lib.stream_ptr()) == 0
def timed(fn, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fwd(); bwd(); torch.cuda.synchronize()
print(f"P={P} L={L} F={F} K={K}: forward {timed(fwd):6.1f} us   backward {timed(bwd):6.1f} us  (pack / memset / unpack included)")
