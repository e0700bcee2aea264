import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(R, "imm-tsf_amd"))
import torch
from immtsf import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
def bench(layout, M, N, K, v, sk):
    A = torch.randn((M, K) if layout < 2 else (K, M), device=dev); B = torch.randn((N, K) if layout == 0 else (K, N), device=dev)
    Cm = torch.zeros(M, N, device=dev)
    lib.immtsf_debug_gemm_config(v, sk)
    def run():
        lib.immtsf_gemm(layout, 1, _lib.ptr(A), A.shape[1], _lib.ptr(B), B.shape[1], _lib.ptr(Cm), N, None, M, N, K, 1.0, 0, 0, _lib.stream_ptr())
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
    lib.immtsf_debug_gemm_config(0, 0)
    return e0.elapsed_time(e1) / 50 * 1e3
for (M, N, K) in [(768, 768, 2048), (1536, 768, 1117), (768, 1152, 1117), (768, 768, 1117)]:
    for v in (1, 5, 11, 14, 15):
        print(f"TN {M}x{N}x{K} v{v}: " + "  ".join(f"s{sk} {bench(2, M, N, K, v, sk):6.1f}" for sk in (1, 2, 3, 4, 6)))
