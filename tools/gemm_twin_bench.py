#!/usr/bin/env python3
"""Graph-timed GEMM launches (50 per hipGraph) with the weight operand read as fp32 vs from a registered bf16 twin, for
the tile variants the launcher can pick (v0 = its own choice).  profiles/ and DESIGN.md section 8 quote these numbers."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imm-tsf_amd"))
import torch  # noqa: E402

from immtsf import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")


def bench(layout, M, N, K, cfg, twin=False):
    A = torch.randn((M, K) if layout < 2 else (K, M), device=dev)
    B = torch.randn((N, K) if layout == 0 else (K, N), device=dev)
    Cm = torch.empty(M, N, device=dev)
    if twin:
        Bt = B.to(torch.bfloat16).contiguous()
        lib.immtsf_bf16_twin_register(_lib.ptr(B), _lib.ptr(Bt), B.numel())
    lib.immtsf_debug_gemm_config(cfg, 0)

    def run():
        lib.immtsf_gemm(layout, 1, _lib.ptr(A), A.shape[1], _lib.ptr(B), B.shape[1], _lib.ptr(Cm), N, None, M, N, K, 1.0, 0, 0,
                        _lib.stream_ptr())
    for _ in range(3):
        run()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    lib.immtsf_debug_gemm_config(0, 0)
    if twin:
        lib.immtsf_bf16_twin_unregister(_lib.ptr(B))
    return e0.elapsed_time(e1) / 50 * 1e3


print("library:", os.path.basename(_lib.LIB_PATH))
shapes = [(0, 2048, 768, 768), (1, 2048, 768, 768), (0, 2048, 1536, 768), (1, 2048, 1152, 768), (0, 1117, 1536, 768), (1, 1117, 1152, 768)]
if len(sys.argv) >= 5:          # gemm_twin_bench.py LAYOUT M N K [LAYOUT M N K ...]
    a = [int(v) for v in sys.argv[1:]]
    shapes = [tuple(a[i:i + 4]) for i in range(0, len(a) - 3, 4)]
for layout, M, N, K in shapes:
    row = [f"{['NT','NN','TN'][layout]} {M}x{N}x{K}:"]
    for v in (0, 4, 11, 14, 17):
        row.append(f"v{v} {bench(layout, M, N, K, v):7.1f} / twin {bench(layout, M, N, K, v, True):7.1f}")
    print("  ".join(row))
