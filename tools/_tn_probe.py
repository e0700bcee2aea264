import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imm-tsf_amd"))
import torch
from immtsf import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
M, N, K = [int(x) for x in sys.argv[1:4]]
what = sys.argv[4]
A = (torch.randn(K, M, device=dev)).bfloat16(); B = (torch.randn(K, N, device=dev)).bfloat16()
C = torch.zeros(M, N, device=dev); bg = torch.zeros(M, device=dev)
print("start", what, flush=True)
t0 = time.time()
if len(sys.argv) > 5:
    nb = lib.immtsf_gemm3_tn_workspace_bytes(M, N, K)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = {"vendor": lambda: torch.matmul(A.t(), B, out=o),
         "g2": lambda: lib.immtsf_gemm_bf16(2, _lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(C), N, None, N, None, _lib.ptr(bg), M, N, K, 1.0, 0, 0, None, 0, None, _lib.stream_ptr()),
         "g3": lambda: lib.immtsf_gemm3_tn_bf16(_lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(C), N, _lib.ptr(bg), M, N, K, 1.0, 0, None, _lib.ptr(ws), nb, _lib.stream_ptr())}[what]
    for _ in range(2): f()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): f()
    torch.cuda.synchronize()
    print("capture", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    print("replay", flush=True)
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print("us", e0.elapsed_time(e1) / 20 * 1e3, flush=True)
    sys.exit(0)
if what == "vendor":
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(A.t(), B, out=o)
elif what == "g2":
    for _ in range(3):
        rc = lib.immtsf_gemm_bf16(2, _lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(C), N, None, N, None, _lib.ptr(bg), M, N, K, 1.0, 0, 0, None, 0, None, _lib.stream_ptr())
    print("rc", rc)
else:
    nb = lib.immtsf_gemm3_tn_workspace_bytes(M, N, K)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    for _ in range(3):
        rc = lib.immtsf_gemm3_tn_bf16(_lib.ptr(A), M, _lib.ptr(B), N, _lib.ptr(C), N, _lib.ptr(bg), M, N, K, 1.0, 0, None, _lib.ptr(ws), nb, _lib.stream_ptr())
    print("rc", rc, nb)
torch.cuda.synchronize()
print("done", what, time.time() - t0, flush=True)
