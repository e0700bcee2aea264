#!/usr/bin/env python3
"""Correctness + graph-timed micro-benchmark of the bf16-in-memory GEMM (csrc/gemm2.hip, through immtsf_gemm_bf16) at
the shapes the fusion step launches, per tile variant, against the round-1 kernel (fp32 activations + bf16 weight twin)
and the vendor GEMM (torch.matmul on bf16 = hipBLASLt) on the same box.
usage: python tools/gemm2_bench.py [quick|full]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "imm-tsf_amd"))
import torch  # noqa: E402

from immtsf import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
VAR = {1: "64x64 s4", 2: "64x96 s4", 3: "128x64 s4", 4: "64x128 s4", 5: "128x128 s3", 6: "128x128w8 s4", 7: "256x128w8 s3",
       8: "64x64 s3", 9: "64x64w8 s4", 10: "96x64 s4", 11: "64x96 s6", 12: "128x96 s4", 13: "256x256w8 s2", 14: "64x96 k2 s3",
       15: "64x64 k2 s3", 16: "64x96 k4 s2", 17: "64x64 k4 s2", 18: "128x128 k2 s2", 19: "96x64 k2 s3", 20: "128x64 k2 s3",
       21: "64x128 k2 s3", 22: "128x96 k2 s2", 23: "64x64 k2 s4", 24: "256x128w8 s2", 25: "96x96 k2 s3", 26: "64x96 k3 s2",
       27: "96x64 k4 s2"}
SMALL = (8, 9, 14, 16, 17, 20, 22, 25, 26, 27)
BIG = (5, 6, 7, 13, 18, 24)


def graph_time(run, n=50):
    for _ in range(3):
        run()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            run()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def operands(layout, M, N, K):
    A = torch.randn((M, K) if layout < 2 else (K, M), device=dev)
    B = torch.randn((N, K) if layout == 0 else (K, N), device=dev)
    return A, B


def ref_of(layout, Ah, Bh):
    a, b = Ah.float(), Bh.float()
    return (a @ b.t()) if layout == 0 else (a @ b) if layout == 1 else (a.t() @ b)


def run2(layout, Ah, Bh, C, Ch, M, N, K, bias=None, bgrad=None, act=0, dyn=None, dyn_which=0, rowmap=None, alpha=1.0, acc=0):
    return lib.immtsf_gemm_bf16(layout, _lib.ptr(Ah), Ah.shape[1], _lib.ptr(Bh), Bh.shape[1], _lib.ptr(C), N, _lib.ptr(Ch), N,
                                _lib.ptr(bias), _lib.ptr(bgrad), M, N, K, alpha, acc, act, _lib.ptr(dyn), dyn_which, _lib.ptr(rowmap),
                                _lib.stream_ptr())


def check(layout, M, N, K, variant, splitk=0):
    A, B = operands(layout, M, N, K)
    Ah, Bh = A.bfloat16(), B.bfloat16()
    C = torch.zeros(M, N, device=dev)
    Ch = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    lib.immtsf_debug_gemm2_config(variant, splitk, -1)
    rc = run2(layout, Ah, Bh, C, Ch if splitk <= 1 else None, M, N, K)
    torch.cuda.synchronize()
    lib.immtsf_debug_gemm2_config(0, 0, -1)
    if rc != 0:
        return None
    ref = ref_of(layout, Ah, Bh)
    e = float((C - ref).abs().max() / ref.abs().max())
    eh = float((Ch.float() - ref).abs().max() / ref.abs().max()) if splitk <= 1 else 0.0
    return e, eh


def bench2(layout, M, N, K, variant, splitk=0, out="f32", xcd=-1):
    A, B = operands(layout, M, N, K)
    Ah, Bh = A.bfloat16(), B.bfloat16()
    C = torch.zeros(M, N, device=dev) if out in ("f32", "both") else None
    Ch = torch.zeros(M, N, device=dev, dtype=torch.bfloat16) if out in ("bf16", "both") else None
    lib.immtsf_debug_gemm2_config(variant, splitk, xcd)
    if run2(layout, Ah, Bh, C, Ch, M, N, K) != 0:
        lib.immtsf_debug_gemm2_config(0, 0, -1)
        return None
    us = graph_time(lambda: run2(layout, Ah, Bh, C, Ch, M, N, K))
    lib.immtsf_debug_gemm2_config(0, 0, -1)
    return us


def bench_old(layout, M, N, K):
    A, B = operands(layout, M, N, K)
    Cm = torch.empty(M, N, device=dev)
    Bt = B.to(torch.bfloat16).contiguous()
    if layout != 2:
        lib.immtsf_bf16_twin_register(_lib.ptr(B), _lib.ptr(Bt), B.numel())
    us = graph_time(lambda: lib.immtsf_gemm(layout, 1, _lib.ptr(A), A.shape[1], _lib.ptr(B), B.shape[1], _lib.ptr(Cm), N, None,
                                            M, N, K, 1.0, 0, 0, _lib.stream_ptr()))
    if layout != 2:
        lib.immtsf_bf16_twin_unregister(_lib.ptr(B))
    return us


def bench_vendor(layout, M, N, K):
    A, B = operands(layout, M, N, K)
    Ah, Bh = A.bfloat16(), B.bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    if layout == 0:
        f = lambda: torch.matmul(Ah, Bh.t(), out=out)
    elif layout == 1:
        f = lambda: torch.matmul(Ah, Bh, out=out)
    else:
        f = lambda: torch.matmul(Ah.t(), Bh, out=out)
    return graph_time(f)


# ---------------------------------------------------------------- correctness (edge shapes, epilogues)
print("== correctness (rel. max error vs fp32 matmul of the bf16 operands; fp32 out / bf16 out)")
bad = 0
for layout, M, N, K in [(0, 2048, 768, 768), (1, 2048, 768, 768), (2, 768, 768, 2048), (0, 1117, 1536, 768), (0, 1117, 768, 1152),
                        (1, 1117, 1152, 768), (2, 1536, 768, 1117), (2, 768, 1152, 1117), (0, 100, 72, 40), (1, 70, 72, 40),
                        (2, 72, 136, 100), (0, 513, 200, 1000), (2, 64, 64, 8)]:
    for v in VAR:
        r = check(layout, M, N, K, v)
        if r is None:
            continue
        ok = r[0] < 2e-5 and r[1] < 1e-2
        bad += 0 if ok else 1
        if not ok or v in (1, 13, 16, 18):
            print(f"{['NT','NN','TN'][layout]} {M}x{N}x{K} v{v:2d} {VAR[v]:13s} err {r[0]:.1e} / {r[1]:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    if layout == 2:
        for sk in (2, 3):
            r = check(layout, M, N, K, 1, sk)
            ok = r is not None and r[0] < 2e-5
            bad += 0 if ok else 1
            print(f"TN {M}x{N}x{K} v1 splitk={sk} err {r[0] if r else -1:.1e} {'ok' if ok else 'FAIL'}", flush=True)
# epilogue options + dyn + rowmap + bias gradient
M, N, K = 300, 200, 136
A, B = operands(0, M, N, K)
Ah, Bh = A.bfloat16(), B.bfloat16()
bias = torch.randn(N, device=dev)
C = torch.zeros(M, N, device=dev)
Ch = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
run2(0, Ah, Bh, C, Ch, M, N, K, bias=bias, act=1, alpha=0.5)
ref = torch.relu(0.5 * ref_of(0, Ah, Bh) + bias)
e = float((C - ref).abs().max() / ref.abs().max())
print(f"NT bias+relu+alpha err {e:.1e} {'ok' if e < 2e-5 else 'FAIL'}")
bad += 0 if e < 2e-5 else 1
dyn = torch.tensor([173], device=dev, dtype=torch.int32)
rowmap = torch.randperm(M, device=dev).int()
C.fill_(-7.0)
run2(0, Ah, Bh, C, None, M, N, K, dyn=dyn, dyn_which=0, rowmap=rowmap)
ref = ref_of(0, Ah[rowmap.long()[:173]], Bh)
e = float((C[:173] - ref).abs().max() / ref.abs().max())
untouched = bool((C[173:] == -7.0).all())
print(f"NT dyn M + rowmap err {e:.1e} rows past M untouched {untouched} {'ok' if e < 2e-5 and untouched else 'FAIL'}")
bad += 0 if (e < 2e-5 and untouched) else 1
Kt, Mt, Nt = 500, 136, 200
At, Bt = operands(2, Mt, Nt, Kt)
Ath, Bth = At.bfloat16(), Bt.bfloat16()
dynk = torch.tensor([333], device=dev, dtype=torch.int32)
Ct = torch.zeros(Mt, Nt, device=dev)
bg = torch.zeros(Mt, device=dev)
run2(2, Ath, Bth, Ct, None, Mt, Nt, Kt, bgrad=bg, dyn=dynk, dyn_which=1)
ref = ref_of(2, Ath[:333], Bth[:333])
e = float((Ct - ref).abs().max() / ref.abs().max())
eb = float((bg - Ath[:333].float().sum(0)).abs().max() / Ath[:333].float().sum(0).abs().max())
print(f"TN dyn K + bias grad err {e:.1e} / {eb:.1e} {'ok' if e < 2e-5 and eb < 2e-5 else 'FAIL'}")
bad += 0 if (e < 2e-5 and eb < 2e-5) else 1
print("correctness failures:", bad, flush=True)

# ---------------------------------------------------------------- timing
print("== timing (us per launch, 50 launches per hipGraph, best of 3 replays)")
shapes = [(0, 2048, 768, 768), (1, 2048, 768, 768), (2, 768, 768, 2048), (0, 2048, 1536, 768), (1, 2048, 768, 1536),
          (2, 1536, 768, 2048), (0, 1117, 1536, 768), (2, 1536, 768, 1117), (0, 768, 768, 768), (2, 768, 768, 768)]
if mode == "bigm":
    # many rows, short K (forward / data-gradient GEMMs at 1024 windows per GPU)
    for (lay, M, N, K) in ((0, 32768, 768, 768), (1, 32768, 768, 768), (0, 32768, 1536, 768), (1, 32768, 768, 1536)):
        fl = 2.0 * M * N * K
        print(f"{['NT','NN','TN'][lay]} {M}x{N}x{K}: vendor {bench_vendor(lay, M, N, K):8.1f} us   heuristic {bench2(lay, M, N, K, 0, 0, 'both'):8.1f} us (fp32 + bf16 result)", flush=True)
        for v in (5, 6, 7, 9, 12, 13, 24):
            u = bench2(lay, M, N, K, v, 1, "both")
            print(f"   v{v:2d} {VAR[v]:13s} {u:8.1f} us {fl/u/1e6:7.1f} TF" if u else f"   v{v:2d} -", flush=True)
    sys.exit(0)
if mode == "longk5":
    M, N, K = 768, 4096, 145408
    print(f"TN {M}x{N}x{K}: vendor {bench_vendor(2, M, N, K):8.1f} us   heuristic {bench2(2, M, N, K, 0, 0, 'f32'):8.1f} us", flush=True)
    for v in (18, 22, 7, 6):
        row = []
        for sk in (1, 2, 3, 4):
            u = bench2(2, M, N, K, v, sk, "f32")
            row.append(f"sk{sk} {u:7.1f}" if u else f"sk{sk}    -")
        print(f"   v{v:2d} {VAR[v]:13s} " + " ".join(row), flush=True)
    sys.exit(0)
if mode == "longk":
    # long reductions (weight gradients at >= 512 windows per GPU): tile variants x cross-workgroup split-K
    for K in (8192, 32768):
        M = N = 768
        print(f"TN {M}x{N}x{K}: vendor {bench_vendor(2, M, N, K):8.1f} us   heuristic {bench2(2, M, N, K, 0, 0, 'f32'):8.1f} us", flush=True)
        for v in (9, 17, 6, 18, 22):
            row = []
            for sk in (1, 4, 7, 14):
                u = bench2(2, M, N, K, v, sk, "f32")
                row.append(f"sk{sk} {u:7.1f}" if u else f"sk{sk}    -")
            print(f"   v{v:2d} {VAR[v]:13s} " + " ".join(row), flush=True)
    sys.exit(0)
if mode != "small":
    shapes += [(0, 4096, 4096, 4096), (1, 4096, 4096, 4096), (2, 4096, 4096, 4096)]
if mode == "full":
    shapes += [(0, 32768, 768, 768), (1, 32768, 768, 768), (2, 768, 768, 32768), (0, 8192, 8192, 8192)]
for layout, M, N, K in shapes:
    fl = 2.0 * M * N * K
    tag = f"{['NT','NN','TN'][layout]} {M}x{N}x{K}"
    old, ven = bench_old(layout, M, N, K), bench_vendor(layout, M, N, K)
    print(f"{tag}: r01 kernel {old:7.1f} us ({fl/old/1e6:6.1f} TF)  vendor bf16 {ven:7.1f} us ({fl/ven/1e6:6.1f} TF)", flush=True)
    big = M * N >= 4096 * 4096
    for v in (BIG if big else SMALL):
        for out in ("f32", "bf16"):
            us = bench2(layout, M, N, K, v, 1, out)
            if us is None:
                continue
            extra = ""
            if layout == 2 and out == "f32" and not big:
                parts = []
                for sk in (1, 2, 3, 4):
                    u = bench2(layout, M, N, K, v, sk, out)
                    parts.append(f"sk{sk} {u:6.1f}")
                extra = "  [" + " ".join(parts) + "]"
            print(f"   v{v:2d} {VAR[v]:13s} out={out:4s} {us:7.1f} us  {fl/us/1e6:7.1f} TF{extra}", flush=True)
