"""The bf16-in-memory GEMM entry points of include/immtsf.h against torch on the same bf16 operands (fp32 accumulation either way, so the
bar is round-off: 2e-5 of the result's norm).  The blocks' tests exercise these kernels at the benchmark's shapes; this file pins the
shapes they do not reach -- a single output tile with a split reduction, ragged reduction lengths, partial tiles."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "imm-tsf_amd")]
pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.mark.parametrize("M,N,K,dynk", [(64, 256, 8192, 6000), (128, 64, 8192, 6000), (64, 96, 8192, 5), (64, 64, 8192, None), (256, 256, 16384, 9000),
                                       (264, 256, 8192, 7000), (768, 768, 8192, 6000), (776, 4480, 8192, 8000), (128, 64, 2048, 1500), (64, 256, 4096, None)])
def test_tn_long_reduction_vs_torch(M, N, K, dynk):
    """C (M, N) = A (K, M)^T B (K, N), the weight-gradient shape: immtsf_gemm_bf16 (layout 2) and, where it applies (K >= 8192), the
    persistent split-reduction kernel immtsf_gemm3_tn_bf16 -- incl. ONE output tile (M, N <= 256: every workgroup a split of the same
    tile), a reduction length read from the device, column sums of A."""
    dev = _dev()
    from immtsf import _lib
    from immtsf.ops import ptr, stream_ptr
    lib = _lib.load()
    torch.manual_seed(M + N)
    A = torch.randn(K, M, device=dev).bfloat16().contiguous()
    Bm = torch.randn(K, N, device=dev).bfloat16().contiguous()
    kk = dynk or K
    ref = A[:kk].float().t() @ Bm[:kk].float()
    ref_b = A[:kk].float().sum(0)
    dyn = torch.tensor([kk], dtype=torch.int32, device=dev) if dynk else None
    out = torch.full((M, N), 7.0, device=dev)
    bg = torch.full((M,), 3.0, device=dev)
    _lib.check(lib.immtsf_gemm_bf16(2, ptr(A), M, ptr(Bm), N, ptr(out), N, None, 0, None, ptr(bg), M, N, K, 1.0, 0, 0, ptr(dyn) if dynk else None, 1,
                                    None, stream_ptr()), "gemm_bf16")
    assert float((out - ref).norm() / ref.norm()) < 2e-5
    assert float((bg - ref_b).norm() / ref_b.norm()) < 2e-5
    wsb = lib.immtsf_gemm3_tn_workspace_bytes(M, N, K)
    assert (wsb > 0) == (K >= 8192 and M * N < 256 * 256 * 128)
    if wsb:
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        out2 = torch.full((M, N), 7.0, device=dev)
        bg2 = torch.full((M,), 3.0, device=dev)
        _lib.check(lib.immtsf_gemm3_tn_bf16(ptr(A), M, ptr(Bm), N, ptr(out2), N, ptr(bg2), M, N, K, 1.0, 0, ptr(dyn) if dynk else None, ptr(ws), wsb,
                                            stream_ptr()), "gemm3_tn_bf16")
        assert float((out2 - ref).norm() / ref.norm()) < 2e-5
        assert float((bg2 - ref_b).norm() / ref_b.norm()) < 2e-5


@pytest.mark.parametrize("layout,M,N,K", [(0, 2048, 768, 4480), (1, 2048, 4480, 768), (0, 100, 72, 136), (1, 37, 200, 64), (0, 6000, 768, 1152), (1, 6000, 1152, 768)])
def test_nt_nn_vs_torch(layout, M, N, K):
    """C = A B^T (layout 0, B (N, K)) and C = A B (layout 1, B (K, N)) through immtsf_gemm_bf16, fp32 and bf16 results, with a bias."""
    dev = _dev()
    from immtsf import _lib
    from immtsf.ops import ptr, stream_ptr
    lib = _lib.load()
    torch.manual_seed(M + K)
    A = torch.randn(M, K, device=dev).bfloat16().contiguous()
    Bm = (torch.randn(N, K, device=dev) if layout == 0 else torch.randn(K, N, device=dev)).bfloat16().contiguous()
    bias = torch.randn(N, device=dev)
    ref = A.float() @ (Bm.float().t() if layout == 0 else Bm.float()) + bias
    out = torch.full((M, N), 7.0, device=dev)
    outh = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.immtsf_gemm_bf16(layout, ptr(A), K, ptr(Bm), K if layout == 0 else N, ptr(out), N, ptr(outh), N, ptr(bias), None, M, N, K, 1.0, 0, 0,
                                    None, 0, None, stream_ptr()), "gemm_bf16")
    assert float((out - ref).norm() / ref.norm()) < 2e-5
    assert float((outh.float() - ref).norm() / ref.norm()) < 6e-3
