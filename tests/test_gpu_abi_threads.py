"""include/immtsf.h's concurrency contract on the GPU: two host threads, each on its own HIP stream with its own buffers,
call the C ABI at the same time (ctypes releases the GIL for the duration of a call) -- the fusion blocks' forward
(TTF_T2V_XAttn + MMF_XAttn_Add, bf16 dataflow, so every weight GEMM looks its operand up in the twin registry), raw GEMMs
against a registered bf16 twin, and a linear layer's backward -- while the main thread registers and unregisters unrelated
twin ranges and flips nothing else.  Every thread must reproduce, bit for bit, what the same work gives when it runs
alone (forward kernels and the one-wave-per-tile GEMMs are deterministic; the weight and bias gradients, whose partial
sums meet in fp32 atomics, are compared at 1e-5)."""
import threading
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(d_m_name, d_txt, H, C):
    return types.SimpleNamespace(TTF_module="TTF_T2V_XAttn", MMF_module="MMF_XAttn_Add", llm_model_fusion=d_m_name,
                                 llm_layers_fusion=6, max_length=1024, device="cuda", use_text_embeddings=True, recency_sigma=1.3,
                                 n_heads_fusion=H, dropout=0.0, d_txt=d_txt, C=C, kappa=0.5)


def _job(seed, stream, iters, out):
    """everything one thread does; results of the last iteration go to out[seed]"""
    from fusions.FusionModel import FusionModel
    from immtsf import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    try:
        with torch.cuda.stream(stream), torch.no_grad():
            g = torch.Generator().manual_seed(seed)
            B, N, T, C, d_m, d = 6 + seed, 9, 7, 5, 64, 32
            m = FusionModel(_args("THR64", d, 2, C)).to(dev).eval()
            for p in m.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
            notes = torch.randn(B, N, d_m, generator=g)
            notes[1, 4:] = 0
            tau = torch.sort(torch.rand(B, N, generator=g) * 24, dim=1).values
            t_hat = torch.sort(torch.rand(B, T, generator=g), dim=1).values
            Y = torch.randn(B, T, C, generator=g)
            notes, tau, t_hat, Y = [t.to(dev) for t in (notes, tau, t_hat, Y)]
            M, Nn, K = 200 + 8 * seed, 96, 160
            A = torch.randn(M, K, generator=g).to(dev)
            W = torch.randn(Nn, K, generator=g).to(dev)
            dy = torch.randn(M, Nn, generator=g).to(dev)
            twin = W.bfloat16().contiguous()
            sp = _lib.stream_ptr()          # torch's current stream is per thread: this thread's `stream`
            for _ in range(iters):
                fused = m(notes, tau, t_hat, Y)
                _lib.check(lib.immtsf_bf16_twin_register(_lib.ptr(W), _lib.ptr(twin), W.numel()), "register")
                Cm = torch.empty(M, Nn, device=dev)
                _lib.check(lib.immtsf_gemm(0, 1, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(Cm), Nn, None, M, Nn, K, 1.0, 0, 0, sp), "gemm")
                Ah = A.bfloat16()
                C2 = torch.empty(M, Nn, device=dev)
                _lib.check(lib.immtsf_gemm_bf16(0, _lib.ptr(Ah), K, _lib.ptr(twin), K, _lib.ptr(C2), Nn, None, Nn, None, None, M, Nn, K,
                                                1.0, 0, 0, None, 0, None, sp), "gemm_bf16")
                dx, dW, db = torch.empty(M, K, device=dev), torch.empty(Nn, K, device=dev), torch.empty(Nn, device=dev)
                _lib.check(lib.immtsf_linear_backward(0, _lib.ptr(A), _lib.ptr(W), _lib.ptr(dy), M, Nn, K, _lib.ptr(dx), None, _lib.ptr(dW),
                                                      _lib.ptr(db), 0, sp), "linear_backward")
                _lib.check(lib.immtsf_bf16_twin_unregister(_lib.ptr(W)), "unregister")
            stream.synchronize()
            out[seed] = dict(fused=fused.cpu(), C=Cm.cpu(), C2=C2.cpu(), dx=dx.cpu(), dW=dW.cpu(), db=db.cpu())
    except Exception as e:       # noqa: BLE001 -- reported by the asserting thread
        out[seed] = e


def test_two_threads_two_streams_match_solo_runs():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from fusions.load_llm import register_d_model
    from immtsf import _lib, config
    lib = _lib.load()
    register_d_model("THR64", 64)
    config.precision = "bf16"
    try:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        solo = {}
        for s in (0, 1):
            _job(s, streams[s], 2, solo)
            assert not isinstance(solo[s], Exception), solo[s]
        both = {}
        th = [threading.Thread(target=_job, args=(s, streams[s], 40, both)) for s in (0, 1)]
        for t in th:
            t.start()
        # the main thread churns the registry meanwhile: lookups in the workers must see whole ranges or none
        junk = torch.zeros(4096, device="cuda")
        junk_h = torch.zeros(4096, device="cuda", dtype=torch.bfloat16)
        n = 0
        while any(t.is_alive() for t in th):
            _lib.check(lib.immtsf_bf16_twin_register(_lib.ptr(junk), _lib.ptr(junk_h), junk.numel()), "register")
            _lib.check(lib.immtsf_bf16_twin_unregister(_lib.ptr(junk)), "unregister")
            n += 1
        for t in th:
            t.join()
        assert n > 0
        for s in (0, 1):
            assert not isinstance(both[s], Exception), both[s]
            for k in ("fused", "C", "C2", "dx"):
                assert torch.equal(both[s][k], solo[s][k]), (s, k, float((both[s][k] - solo[s][k]).abs().max()))
            for k in ("dW", "db"):          # split-K / column-sum atomics: summation order varies run to run
                ref = solo[s][k]
                assert float((both[s][k] - ref).abs().max()) <= 1e-5 * float(ref.abs().max()), (s, k)
            assert torch.isfinite(both[s]["fused"]).all()
    finally:
        config.precision = "fp32"
        torch.cuda.synchronize()
