import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "imm-tsf_amd"))
import torch
from immtsf import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
def bench(layout, M, N, K, cfg):
    A = torch.randn((M, K) if layout < 2 else (K, M), device=dev); B = torch.randn((N, K) if layout == 0 else (K, N), device=dev)
    Cm = torch.empty(M, N, device=dev)
    lib.immtsf_debug_gemm_config(cfg, 0)
    def run():
        lib.immtsf_gemm(layout, 1, _lib.ptr(A), A.shape[1], _lib.ptr(B), B.shape[1], _lib.ptr(Cm), N, None, M, N, K, 1.0, 0, 0, _lib.stream_ptr())
    for _ in range(3): run()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(50): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    lib.immtsf_debug_gemm_config(0, 0)
    return e0.elapsed_time(e1) / 50 * 1e3
for layout, M, N, K in [(0, 2048, 768, 768), (1, 2048, 768, 768), (2, 768, 768, 2048), (0, 4096, 4096, 4096)]:
    for v in (1, 4):
        full = bench(layout, M, N, K, v); half = bench(layout, M, N, K, v | (8 << 9)); nog = bench(layout, M, N, K, v | (2 << 9))
        print(f"{['NT','NN','TN'][layout]} {M}x{N}x{K} v{v}: full {full:7.1f}  half-bytes {half:7.1f}  no-gload {nog:7.1f}")
