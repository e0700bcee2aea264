#!/usr/bin/env python3
"""P half of MMF_XAttn_Add's low-rank form alone (fold + P projection; backward: dE, dW_fold, chain rule): us per call at a given
number of windows, for A/B runs of IMMTSF_XRANK_ROWS (streaming kernels vs the GEMM path).  usage: xrank_bench.py [windows] [T]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "imm-tsf_amd")]
import torch  # noqa: E402


def main():
    from fusions.MMF_XAttn_Add import MMF_XAttn_Add
    from immtsf import config
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda", 0)
    config.precision = os.environ.get("PREC", "bf16")
    torch.manual_seed(0)
    mmf = MMF_XAttn_Add(768, 8, 768, n_heads_fusion=1, dropout=0.1, kappa=0.5).to(dev).train()
    E = torch.randn(B, T, 768, device=dev, requires_grad=True)
    P, bHO = mmf.project_kv(E)
    gP, gb = torch.randn_like(P), torch.randn_like(bHO)

    def fwd():
        return mmf.project_kv(E)

    def fwdbwd():
        P, bHO = mmf.project_kv(E)
        torch.autograd.backward([P, bHO], [gP, gb])

    for name, fn in (("forward", fwd), ("forward+backward", fwdbwd)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / n * 1e3:.1f} us  (B={B}, T={T}, rows={os.environ.get('IMMTSF_XRANK_ROWS', '1')})")


if __name__ == "__main__":
    main()
