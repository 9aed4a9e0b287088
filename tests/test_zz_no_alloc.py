"""vae_gemm_noalloc (include/vae_amd.h): vae_gemm with its split-K slabs in a static device array, for callers that
cannot allocate.  Through the C ABI, against vae_gemm (caller workspace) on the same inputs: bit-identical, on two calls
in a row, the second with a smaller input (the shared array keeps the first call's slabs); an oversize shape is
rejected with a clear error and leaves the output untouched.  Collected last (file name) so that a failure here cannot
hide another test."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, layout, precision, M, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a_shape = (K, M) if layout == 2 else (M, K)
    b_shape = (N, K) if layout == 0 else (K, N)
    A = torch.randn(a_shape, generator=g, device="cuda")
    Bm = torch.randn(b_shape, generator=g, device="cuda")
    bias = torch.randn(N, generator=g, device="cuda")
    need = lib.vae_gemm_workspace_bytes(M, N, K)
    ws = torch.empty(max(need // 4, 4), dtype=torch.float32, device="cuda")
    ref = torch.full((M, N), 7.0, device="cuda")
    out = torch.full((M, N), -3.0, device="cuda")
    torch.cuda.synchronize()          # every allocation of the test is done before the entry point under test runs
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.vae_gemm(layout, precision, M, N, K, A.data_ptr(), A.stride(0), Bm.data_ptr(), Bm.stride(0), ref.data_ptr(), N,
                      bias.data_ptr(), None, 0, ws.data_ptr(), need, st)
    assert rc == 0, lib.vae_last_error()
    before = torch.cuda.memory_allocated()
    rc = lib.vae_gemm_noalloc(layout, precision, M, N, K, A.data_ptr(), A.stride(0), Bm.data_ptr(), Bm.stride(0),
                              out.data_ptr(), N, bias.data_ptr(), None, 0, st)
    assert rc == 0, lib.vae_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    return need, ref, out, A, Bm, bias


def test_noalloc_gemm_matches_vae_gemm_twice_then_smaller():
    from vae_amd import _lib
    lib = _lib.load()
    assert lib.vae_gemm_noalloc_max_bytes() == 40 << 20
    assert lib.vae_gemm_noalloc_init() == 0 and lib.vae_gemm_noalloc_init() == 0      # eager address lookup, idempotent
    # 1) the encoder forward of BASELINE configs[1]: 42 split-K slabs of 512 x 384 (33 MB of scratch), fp32-grade
    need, ref, out, A, Bm, bias = _both(lib, 0, 0, 512, 384, 20000, seed=1)
    assert need > 30 << 20
    assert torch.equal(ref, out)
    truth = A.double() @ Bm.double().t() + bias.double()
    assert float((out.double() - truth).abs().max() / truth.abs().max()) < 1e-5
    # 2) right after it, a SMALLER product whose slabs overlay the first call's (stale contents must not leak): wgrad layout
    need2, ref2, out2, A2, B2, bias2 = _both(lib, 2, 0, 128, 1000, 4096, seed=2)
    assert 0 < need2 < need
    assert torch.equal(ref2, out2)
    truth2 = A2.double().t() @ B2.double() + bias2.double()
    assert float((out2.double() - truth2).abs().max() / truth2.abs().max()) < 1e-5
    # 3) bf16 operands, and a shape that needs no scratch at all
    need3, ref3, out3, *_ = _both(lib, 0, 1, 512, 40000, 256, seed=3)
    assert need3 == 0 and torch.equal(ref3, out3)


def test_noalloc_gemm_rejects_what_does_not_fit():
    from vae_amd import _lib
    lib = _lib.load()
    M, N, K = 4096, 384, 20000                                   # the same product at batch 4096: 6+ slabs of 6.3 MB each
    need = lib.vae_gemm_workspace_bytes(M, N, K)
    if need <= lib.vae_gemm_noalloc_max_bytes():
        M, need = 8192, lib.vae_gemm_workspace_bytes(8192, N, K)
    assert need > lib.vae_gemm_noalloc_max_bytes()
    A = torch.zeros(M, K, device="cuda")
    Bm = torch.zeros(N, K, device="cuda")
    out = torch.full((M, N), 5.0, device="cuda")
    rc = lib.vae_gemm_noalloc(0, 0, M, N, K, A.data_ptr(), K, Bm.data_ptr(), K, out.data_ptr(), N, None, None, 0,
                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 1 and b"static scratch holds" in lib.vae_last_error()
    torch.cuda.synchronize()
    assert float(out.min()) == 5.0 and float(out.max()) == 5.0   # nothing was launched
