#!/usr/bin/env python3
"""tPatchGNN forecast decoder (one kernel per direction) through the C ABI, 20 launches per hipGraph.
usage: decoder_bench.py [B N Lp D E]   (default: the benchmark shape 64 8 32 32 10; H = 32)"""
import ctypes as C
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "imm-tsf_amd"))
import torch
from immtsf import _lib
from immtsf.ops import DecoderParams, _struct
lib = _lib.load(); dev = torch.device("cuda:0"); ptr = _lib.ptr
B, N, Lp, D, E = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else (64, 8, 32, 32, 10)
H = 32
torch.manual_seed(0)
h, te, dout = torch.randn(B, N, D, device=dev), torch.randn(B, Lp, E, device=dev), torch.randn(B, Lp, N, device=dev)
params = [torch.randn(*s, device=dev) * 0.2 for s in [(H, D + E), (H,), (H, H), (H,), (1, H), (1,)]]
grads = [torch.zeros_like(p) for p in params]
out, dh, dte = torch.empty(B, Lp, N, device=dev), torch.empty_like(h), torch.empty_like(te)
ps, gs = _struct(DecoderParams, params), _struct(DecoderParams, grads)
PREC = 0
def fwd():
    assert lib.immtsf_tpatchgnn_decoder_forward_p(B, N, Lp, D, E, H, PREC, ptr(h), ptr(te), C.byref(ps), ptr(out), _lib.stream_ptr()) == 0
def bwd():
    assert lib.immtsf_tpatchgnn_decoder_backward_p(B, N, Lp, D, E, H, PREC, ptr(h), ptr(te), C.byref(ps), ptr(dout), ptr(dh), ptr(dte), C.byref(gs), _lib.stream_ptr()) == 0
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
for PREC in (0, 1):
    fwd(); bwd(); torch.cuda.synchronize()
    print(f"B={B} N={N} Lp={Lp} D={D} E={E} H={H} precision {PREC} ({'exact fp32' if PREC == 0 else 'bf16 MFMA'}): forward {timed(fwd):6.1f} us   backward {timed(bwd):6.1f} us")
