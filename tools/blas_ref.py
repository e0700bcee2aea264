#!/usr/bin/env python3
"""What do the vendor libraries reach at the fusion GEMM shapes?  (reference point only; the product never calls them)"""
import torch
dev = torch.device("cuda:0")


def timed(fn, n=50):
    for _ in range(3):
        fn()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in [(2048, 768, 768), (768, 768, 2048), (1117, 1536, 768), (4096, 4096, 4096), (512, 32, 64)]:
    for dt in (torch.bfloat16, torch.float32):
        A, B = torch.randn(M, K, device=dev, dtype=dt), torch.randn(N, K, device=dev, dtype=dt)
        Bt = B.t().contiguous()
        out = torch.empty(M, N, device=dev, dtype=dt)
        nt = timed(lambda: torch.mm(A, B.t(), out=out))
        nn = timed(lambda: torch.mm(A, Bt, out=out))
        print(f"{M}x{N}x{K} {str(dt)[6:]:9s} NT {nt:7.1f} us ({2*M*N*K/nt/1e6:6.1f} TF)   NN {nn:7.1f} us ({2*M*N*K/nn/1e6:6.1f} TF)")
x = torch.randn(1 << 20, device=dev)
print("elementwise add 4 MB:", round(timed(lambda: x.add_(1.0)), 2), "us;  tiny (256 elems):", round(timed(lambda: x[:256].add_(1.0)), 2), "us")
