"""time the embedding kernels alone (hipGraph of 20 raw C-ABI launches): PatchTST cfg3 shape"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "imm-tsf_amd"))
import torch
from immtsf import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
R, L, K, stride, pad, D = 1152, 32, 16, 8, 8, 512
P = (L + pad - K) // stride + 1
x = torch.randn(R, L, device=dev)
W = torch.randn(D, K, device=dev)
pe = torch.randn(64, D, device=dev)
out = torch.empty(R, P, D, device=dev)
dout = torch.randn(R, P, D, device=dev)
dW = torch.zeros(D, K, device=dev)


def timeit(f, name):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(gr):
            for _ in range(20):
                f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us", flush=True)


for p in (0.0, 0.1):
    timeit(lambda: _lib.check(lib.immtsf_embed_forward(0, _lib.ptr(x), R, L, 1, P, K, stride, D, _lib.ptr(W), _lib.ptr(pe), _lib.ptr(out), p, 7, 56,
                                                       None, _lib.stream_ptr()), "f"), f"fwd p={p}")
    timeit(lambda: _lib.check(lib.immtsf_embed_backward(0, _lib.ptr(x), R, L, 1, P, K, stride, D, _lib.ptr(W), _lib.ptr(dout), _lib.ptr(dW), 0, None,
                                                        p, 7, 56, None, _lib.stream_ptr()), "b"), f"bwd dW p={p}")
