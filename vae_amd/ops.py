"""Tensor-level wrappers over the C-ABI HIP kernels (include/vae_amd.h).

torch is used for device memory and the current stream only; every computation below is a
hand-written gfx950 kernel.  All tensors must be fp32 CUDA (ROCm) tensors whose last
dimension is contiguous; there is no CPU fallback -- CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib

NT, NN, TN = 0, 1, 2
FP32, BF16, BF16X3, F16X2 = 0, 1, 2, 3

_workspaces = {}


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk2d(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"vae_amd.ops: {name} must live on the GPU (no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"vae_amd.ops: {name} must be float32, got {t.dtype}")
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"vae_amd.ops: {name} must be 2-D with a contiguous last dim, got {tuple(t.shape)} / {t.stride()}")


def _chk_logits(t: torch.Tensor, name: str) -> bool:
    """Logits may be float32 or bfloat16 (what a bf16-mode step stores); -> True for bfloat16."""
    if t.dtype != torch.bfloat16:
        _chk2d(t, name)
        return False
    if not t.is_cuda or t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"vae_amd.ops: {name} must be a 2-D GPU tensor with a contiguous last dim")
    return True


def _chk_counts(t: torch.Tensor, name: str = "x") -> bool:
    """Counts may be float32 or uint8 (the reference's on-disk format); -> True for uint8."""
    if t.dtype != torch.uint8:
        _chk2d(t, name)
        return False
    if not t.is_cuda:
        raise RuntimeError(f"vae_amd.ops: {name} must live on the GPU (no CPU fallback)")
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"vae_amd.ops: {name} must be 2-D with a contiguous last dim, got {tuple(t.shape)} / {t.stride()}")
    return True


def _chk1d(t: Optional[torch.Tensor], n: int, name: str) -> None:
    if t is None:
        return
    if not t.is_cuda or t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
        raise ValueError(f"vae_amd.ops: {name} must be a contiguous float32 GPU tensor of {n} elements")


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


def workspace(nbytes: int, device) -> torch.Tensor:
    """Per-device scratch for split-K slabs, grown on demand (torch = allocator only)."""
    key = torch.device(device).index
    ws = _workspaces.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max(nbytes // 4 + 4, 1 << 20), dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


def gemm(layout: int, a: torch.Tensor, b: torch.Tensor, *, out: Optional[torch.Tensor] = None,
         bias: Optional[torch.Tensor] = None, x_mean: Optional[torch.Tensor] = None,
         accumulate: bool = False, precision: int = FP32) -> torch.Tensor:
    """NT: a(M,K) b(N,K) | NN: a(M,K) b(K,N) | TN: a(K,M) b(K,N)  ->  out(M,N).

    ``x_mean`` normalises the count operand on the fly (a in NT, b in TN): x/mean - 1.
    """
    lib = _lib.load()
    _chk2d(a, "a")
    _chk2d(b, "b")
    if layout == NT:
        M, K = a.shape
        N, K2 = b.shape
    elif layout == NN:
        M, K = a.shape
        K2, N = b.shape
    elif layout == TN:
        K, M = a.shape
        K2, N = b.shape
    else:
        raise ValueError("layout")
    if K != K2:
        raise ValueError(f"vae_amd.ops.gemm: contraction mismatch {tuple(a.shape)} vs {tuple(b.shape)} (layout {layout})")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _chk2d(out, "out")
    if tuple(out.shape) != (M, N):
        raise ValueError(f"vae_amd.ops.gemm: out must be {(M, N)}, got {tuple(out.shape)}")
    _chk1d(bias, N, "bias")
    if x_mean is not None:
        _chk1d(x_mean, M if layout == NT else K, "x_mean")
    nbytes = lib.vae_gemm_workspace_bytes(M, N, K)
    ws = workspace(nbytes, a.device) if nbytes else None
    rc = lib.vae_gemm(layout, precision, M, N, K, _p(a), _ld(a), _p(b), _ld(b), _p(out), _ld(out), _p(bias), _p(x_mean),
                      int(accumulate), _p(ws), nbytes, _stream())
    _lib.check(rc, "gemm")
    return out


def absmax_scale(x: torch.Tensor) -> torch.Tensor:
    """-> 1-element tensor: the power of two s with max|x| * s in [2^13, 2^14) (``gemm_scaled`` operand scale)."""
    lib = _lib.load()
    _chk2d(x, "x")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    nb = lib.vae_absmax_workspace_bytes()
    ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
    _lib.check(lib.vae_absmax_scale(_p(x), x.shape[0], x.shape[1], _ld(x), _p(out), _p(ws), nb, _stream()), "absmax_scale")
    return out


def gemm_scaled(layout: int, a: torch.Tensor, b: torch.Tensor, scale_a: Optional[torch.Tensor] = None,
                scale_b: Optional[torch.Tensor] = None, *, out: Optional[torch.Tensor] = None,
                bias: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """``gemm`` on two f16 terms per element and three products (include/vae_amd.h, vae_gemm_scaled); the scales
    default to ``absmax_scale`` of each operand."""
    lib = _lib.load()
    _chk2d(a, "a"); _chk2d(b, "b")
    if layout == NT:
        (M, K), (N, K2) = a.shape, b.shape
    elif layout == NN:
        (M, K), (K2, N) = a.shape, b.shape
    elif layout == TN:
        (K, M), (K2, N) = a.shape, b.shape
    else:
        raise ValueError("layout must be NT, NN or TN")
    if K != K2:
        raise ValueError(f"vae_amd.ops.gemm_scaled: inner dimensions differ ({K} vs {K2})")
    sa = absmax_scale(a) if scale_a is None else scale_a
    sb = absmax_scale(b) if scale_b is None else scale_b
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _chk2d(out, "out")
    _chk1d(bias, N, "bias")
    nbytes = lib.vae_gemm_workspace_bytes(M, N, K)
    ws = workspace(nbytes, a.device) if nbytes else None
    rc = lib.vae_gemm_scaled(layout, M, N, K, _p(a), _ld(a), _p(sa), _p(b), _ld(b), _p(sb), _p(out), _ld(out), _p(bias),
                             int(accumulate), _p(ws), nbytes, _stream())
    _lib.check(rc, "gemm_scaled")
    return out


def gemm_rowscaled(a: torch.Tensor, b: torch.Tensor, col_max: Optional[torch.Tensor] = None, b_max: Optional[torch.Tensor] = None, *,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a(K, M)^T b(K, N) -> out(M, N) on two f16 terms with one power-of-two scale per OUTPUT ROW (include/vae_amd.h,
    vae_gemm_rowscaled: the step's decoder-head weight gradient).  ``col_max`` (chunks, M): maxima of |a| per column over
    chunks of its rows (default: one chunk, computed here); ``b_max``: partial maxima of |b| (default: computed here)."""
    lib = _lib.load()
    _chk2d(a, "a"); _chk2d(b, "b")
    (K, M), (K2, N) = a.shape, b.shape
    if K != K2:
        raise ValueError(f"vae_amd.ops.gemm_rowscaled: inner dimensions differ ({K} vs {K2})")
    if not lib.vae_gemm_rowscaled_ok(M, N, K):
        raise ValueError(f"vae_amd.ops.gemm_rowscaled: shape {(M, N, K)} is outside the tile pipeline")
    if col_max is None:
        col_max = a.abs().amax(dim=0, keepdim=True).contiguous()
    if b_max is None:
        b_max = b.abs().amax().expand(4).contiguous()
    _chk2d(col_max, "col_max")
    if col_max.shape[1] < M or b_max.dim() != 1 or b_max.dtype != torch.float32 or not b_max.is_cuda or not b_max.is_contiguous():
        raise ValueError("vae_amd.ops.gemm_rowscaled: col_max must be (chunks, >= M), b_max a contiguous 1-D float32 GPU tensor")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _chk2d(out, "out")
    nbytes = lib.vae_gemm_workspace_bytes(M, N, K)
    ws = workspace(nbytes, a.device) if nbytes else None
    rc = lib.vae_gemm_rowscaled(M, N, K, _p(a), _ld(a), _p(col_max), col_max.shape[0], _ld(col_max), _p(b), _ld(b), _p(b_max),
                                b_max.numel(), _p(out), _ld(out), _p(ws), nbytes, _stream())
    _lib.check(rc, "gemm_rowscaled")
    return out


def row_stats(x: torch.Tensor, w: torch.Tensor):
    """-> (mean (B,), absmax (B*4,), flags (B*4,), wsum (HE,)): what the count-operand products need (include/vae_amd.h,
    vae_row_stats): per-cell mean of the raw counts, per-wave maxima of |x|, per-wave "needs a lo term" flags, and the
    row sums of the (HE, G) first-layer weight."""
    lib = _lib.load()
    _chk2d(x, "x"); _chk2d(w, "w")
    B, G = x.shape
    if w.shape[1] != G:
        raise ValueError(f"vae_amd.ops.row_stats: w must be (HE, {G}), got {tuple(w.shape)}")
    mean = torch.empty(B, dtype=torch.float32, device=x.device)
    absmax, flags = torch.empty(B * 4, dtype=torch.float32, device=x.device), torch.empty(B * 4, dtype=torch.float32, device=x.device)
    wsum = torch.empty(w.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(lib.vae_row_stats(_p(x), _ld(x), B, G, _p(mean), _p(absmax), _p(flags), _p(w), _ld(w), w.shape[0], _p(wsum), _stream()),
               "row_stats")
    return mean, absmax, flags, wsum


def gemm_counts(layout: int, a: torch.Tensor, b: torch.Tensor, max_a: torch.Tensor, max_b: torch.Tensor, flags: torch.Tensor, *,
                row_sub: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``gemm_scaled`` whose count operand (``a`` for NT, ``b`` for TN) runs on ONE f16 term when ``flags`` (from
    ``row_stats``) are all zero (include/vae_amd.h, vae_gemm_counts); ``max_a`` / ``max_b``: partial maxima of |a| / |b|;
    ``row_sub`` (TN): out[m, :] -= row_sub[m]."""
    lib = _lib.load()
    _chk2d(a, "a"); _chk2d(b, "b")
    if layout == NT:
        (M, K), (N, K2) = a.shape, b.shape
    elif layout == TN:
        (K, M), (K2, N) = a.shape, b.shape
    else:
        raise ValueError("vae_amd.ops.gemm_counts: layout must be NT or TN")
    if K != K2:
        raise ValueError(f"vae_amd.ops.gemm_counts: inner dimensions differ ({K} vs {K2})")
    for t, name in ((max_a, "max_a"), (max_b, "max_b"), (flags, "flags")):
        if t.dim() != 1 or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise ValueError(f"vae_amd.ops.gemm_counts: {name} must be a contiguous 1-D float32 GPU tensor")
    _chk1d(row_sub, M, "row_sub")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _chk2d(out, "out")
    nbytes = lib.vae_gemm_workspace_bytes(M, N, K)
    ws = workspace(nbytes, a.device) if nbytes else None
    rc = lib.vae_gemm_counts(layout, M, N, K, _p(a), _ld(a), _p(max_a), max_a.numel(), _p(b), _ld(b), _p(max_b), max_b.numel(), _p(out),
                             _ld(out), _p(flags), flags.numel(), _p(row_sub), _p(ws), nbytes, _stream())
    _lib.check(rc, "gemm_counts")
    return out


IO_A_BF16, IO_B_BF16, IO_C_BF16 = 1, 2, 4


def gemm_bf16_storage(layout: int, a: torch.Tensor, b: torch.Tensor, io: int, *, out: Optional[torch.Tensor] = None,
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``gemm(precision=BF16)`` with one matrix held as bfloat16 in HBM (include/vae_amd.h, vae_gemm_bf16_storage):
    ``io`` = IO_A_BF16 (a is a bfloat16 tensor) | IO_B_BF16 (b is) | IO_C_BF16 (the result is; NT only) -- the combinations
    ``vae_gemm_bf16_storage_ok`` accepts."""
    lib = _lib.load()
    if not 0 < io < 8:
        raise ValueError("io must be a combination of IO_A_BF16, IO_B_BF16, IO_C_BF16")
    want = tuple(torch.bfloat16 if io & bit else torch.float32 for bit in (IO_A_BF16, IO_B_BF16, IO_C_BF16))
    for t, dt, name in ((a, want[0], "a"), (b, want[1], "b")):
        if not t.is_cuda or t.dtype != dt or t.dim() != 2 or t.stride(1) != 1:
            raise TypeError(f"vae_amd.ops.gemm_bf16_storage: {name} must be a 2-D {dt} GPU tensor with a contiguous last dim")
    if layout == NT:
        (M, K), (N, K2) = a.shape, b.shape
    elif layout == NN:
        (M, K), (K2, N) = a.shape, b.shape
    elif layout == TN:
        (K, M), (K2, N) = a.shape, b.shape
    else:
        raise ValueError("layout must be NT, NN or TN")
    if K != K2:
        raise ValueError(f"vae_amd.ops.gemm_bf16_storage: inner dimensions differ ({K} vs {K2})")
    if not lib.vae_gemm_bf16_storage_ok(layout, io, M, N, K):
        raise ValueError(f"vae_amd.ops.gemm_bf16_storage: shape {(M, N, K)} / layout {layout} / io {io} is outside the bf16-storage pipeline")
    if out is None:
        out = torch.empty(M, N, dtype=want[2], device=a.device)
    if out.dtype != want[2] or tuple(out.shape) != (M, N) or out.stride(1) != 1 or not out.is_cuda:
        raise TypeError(f"vae_amd.ops.gemm_bf16_storage: out must be a {(M, N)} {want[2]} GPU tensor")
    _chk1d(bias, N, "bias")
    nbytes = lib.vae_gemm_workspace_bytes(M, N, K)
    ws = workspace(nbytes, a.device) if nbytes else None
    rc = lib.vae_gemm_bf16_storage(layout, io, M, N, K, _p(a), _ld(a), _p(b), _ld(b), _p(out), _ld(out), _p(bias), _p(ws), nbytes,
                                   _stream())
    _lib.check(rc, "gemm_bf16_storage")
    return out


def gemm_slabs(layout: int, a: torch.Tensor, b: torch.Tensor, precision: int = FP32) -> torch.Tensor:
    """The product of ``gemm`` left as split-K partial sums: -> (n_slabs, M, N), whose sum over dim 0 is
    the result (no bias).  What the fused encoder-tail / decoder-head kernels consume."""
    lib = _lib.load()
    _chk2d(a, "a"); _chk2d(b, "b")
    if layout == NT:
        (M, K), (N, K2) = a.shape, b.shape
    elif layout == NN:
        (M, K), (K2, N) = a.shape, b.shape
    elif layout == TN:
        (K, M), (K2, N) = a.shape, b.shape
    else:
        raise ValueError("layout must be NT, NN or TN")
    if K != K2:
        raise ValueError(f"gemm_slabs: inner dimensions differ ({K} vs {K2})")
    nbytes = lib.vae_gemm_slabs_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=a.device)
    n = C.c_int(0)
    _lib.check(lib.vae_gemm_slabs(layout, precision, M, N, K, _p(a), _ld(a), _p(b), _ld(b), _p(ws), nbytes, C.byref(n),
                                  _stream()), "gemm_slabs")
    return ws[:n.value * M * N].view(n.value, M, N)


def dropout_apply(x: torch.Tensor, p: float, seed: int, step: int, site: int) -> torch.Tensor:
    """In place ``x[b, c] *= keep / (1 - p)`` with the library's documented Philox stream
    (include/vae_amd.h, "dropout"): nn.Dropout of FCLayers (modules.py:115).  -> x"""
    lib = _lib.load()
    _chk2d(x, "x")
    B, W = x.shape
    _lib.check(lib.vae_dropout_apply(_p(x), _ld(x), B, W, float(p), int(seed) & (2 ** 64 - 1), int(step) & 0xFFFFFFFF,
                                     int(site), _stream()), "dropout_apply")
    return x


def row_mean(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _chk2d(x, "x")
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(lib.vae_row_mean(_p(x), _ld(x), x.shape[0], x.shape[1], _p(out), _stream()), "row_mean")
    return out


def normalize_rows(x: torch.Tensor):
    """-> (mean (B,), xn = x / mean - 1 (B,G)): Encoder.forward's input normalisation (modules.py:264)."""
    lib = _lib.load()
    u8 = _chk_counts(x)
    mean = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    xn = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check((lib.vae_normalize_rows_u8 if u8 else lib.vae_normalize_rows)(_p(x), _ld(x), x.shape[0], x.shape[1], _p(mean), _p(xn), _ld(xn), _stream()),
               "normalize_rows")
    return mean, xn


def gelu_ln_fwd(pre: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    lib = _lib.load()
    _chk2d(pre, "pre")
    B, W = pre.shape
    _chk1d(gamma, W, "gamma")
    _chk1d(beta, W, "beta")
    if out is None:
        out = torch.empty(B, W, dtype=torch.float32, device=pre.device)
    _chk2d(out, "out")
    mean = torch.empty(B, dtype=torch.float32, device=pre.device)
    rstd = torch.empty(B, dtype=torch.float32, device=pre.device)
    _lib.check(lib.vae_gelu_ln_fwd(_p(pre), _ld(pre), _p(gamma), _p(beta), B, W, _p(out), _ld(out), _p(mean), _p(rstd),
                                   _stream()), "gelu_ln_fwd")
    return out, mean, rstd


def gelu_ln_bwd(dout: torch.Tensor, pre: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, gamma: torch.Tensor,
                dgamma: Optional[torch.Tensor], dbeta: Optional[torch.Tensor], dbias: Optional[torch.Tensor],
                dpre: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dgamma / dbeta / dbias are ACCUMULATED into (float atomics): zero them beforehand."""
    lib = _lib.load()
    _chk2d(dout, "dout")
    _chk2d(pre, "pre")
    B, W = pre.shape
    if tuple(dout.shape) != (B, W):
        raise ValueError("vae_amd.ops.gelu_ln_bwd: dout shape mismatch")
    _chk1d(mean, B, "mean"); _chk1d(rstd, B, "rstd"); _chk1d(gamma, W, "gamma")
    _chk1d(dgamma, W, "dgamma"); _chk1d(dbeta, W, "dbeta"); _chk1d(dbias, W, "dbias")
    if dpre is None:
        dpre = torch.empty(B, W, dtype=torch.float32, device=pre.device)
    _chk2d(dpre, "dpre")
    _lib.check(lib.vae_gelu_ln_bwd(_p(dout), _ld(dout), _p(pre), _ld(pre), _p(mean), _p(rstd), _p(gamma), B, W, _p(dpre),
                                   _ld(dpre), _p(dgamma), _p(dbeta), _p(dbias), _stream()), "gelu_ln_bwd")
    return dpre


def reparam_kl_fwd(heads: torch.Tensor, eps: torch.Tensor, prior_mean: float, prior_var: float):
    """heads (B, L+1) = [q_m | variance pre-activation]; eps (B, L) -> q_v (B,), z (B,L), kl (B,)."""
    lib = _lib.load()
    _chk2d(heads, "heads")
    _chk2d(eps, "eps")
    B, L = eps.shape
    if heads.shape != (B, L + 1):
        raise ValueError(f"vae_amd.ops.reparam_kl_fwd: heads must be {(B, L + 1)}, got {tuple(heads.shape)}")
    qv = torch.empty(B, dtype=torch.float32, device=heads.device)
    z = torch.empty(B, L, dtype=torch.float32, device=heads.device)
    kl = torch.empty(B, dtype=torch.float32, device=heads.device)
    _lib.check(lib.vae_reparam_kl_fwd(_p(heads), _ld(heads), _p(eps), _ld(eps), B, L, float(prior_mean), float(prior_var),
                                      _p(qv), _p(z), _ld(z), _p(kl), _stream()), "reparam_kl_fwd")
    return qv, z, kl


def reparam_kl_bwd(dz: torch.Tensor, g_kl: Optional[torch.Tensor], g_kl_scalar: float, heads: torch.Tensor,
                   eps: torch.Tensor, qv: torch.Tensor, prior_mean: float, prior_var: float,
                   dbias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> dheads (B, L+1); ``dbias`` (L+1,), if given, accumulates its column sums (float atomics)."""
    lib = _lib.load()
    _chk2d(dz, "dz"); _chk2d(heads, "heads"); _chk2d(eps, "eps")
    B, L = eps.shape
    if dz.shape != (B, L) or heads.shape != (B, L + 1):
        raise ValueError("vae_amd.ops.reparam_kl_bwd: shape mismatch")
    _chk1d(qv, B, "qv"); _chk1d(g_kl, B, "g_kl"); _chk1d(dbias, L + 1, "dbias")
    dheads = torch.empty(B, L + 1, dtype=torch.float32, device=heads.device)
    _lib.check(lib.vae_reparam_kl_bwd(_p(dz), _ld(dz), _p(g_kl), float(g_kl_scalar), _p(heads), _ld(heads), _p(eps),
                                      _ld(eps), _p(qv), B, L, float(prior_mean), float(prior_var), _p(dheads),
                                      _ld(dheads), _p(dbias), _stream()), "reparam_kl_bwd")
    return dheads


def colsum_into(inp: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[n] += sum_m inp[m, n] (float atomics; zero ``out`` beforehand for a plain sum)."""
    lib = _lib.load()
    _chk2d(inp, "inp")
    _chk1d(out, inp.shape[1], "out")
    _lib.check(lib.vae_colsum(_p(inp), _ld(inp), inp.shape[0], inp.shape[1], _p(out), _stream()), "colsum")
    return out


def elbo_reduce(ll: torch.Tensor, kl_z: torch.Tensor, kl_l: torch.Tensor, kl_weight: float, coeff: float) -> torch.Tensor:
    """-> [mean(recon), mean(kl_z), mean(kl_l), gene_loss, coeff*gene_loss]."""
    lib = _lib.load()
    B = ll.numel()
    _chk1d(ll, B, "ll"); _chk1d(kl_z, B, "kl_z"); _chk1d(kl_l, B, "kl_l")
    out = torch.empty(5, dtype=torch.float32, device=ll.device)
    _lib.check(lib.vae_elbo_reduce(_p(ll), _p(kl_z), _p(kl_l), B, float(kl_weight), float(coeff), _p(out), None, _stream()),
               "elbo_reduce")
    return out


def gene_table(px_r: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(vae_gene_table_rows(), G) per-gene constants of the fused epilogue: theta = exp(px_r),
    log(theta+1e-8), 1/(theta+1e-8), lgamma(theta), digamma(theta) and, for counts k < 16,
    lgamma(theta+k)-lgamma(theta)-lgamma(k+1) and digamma(theta+k)-digamma(theta)."""
    lib = _lib.load()
    G = px_r.numel()
    _chk1d(px_r, G, "px_r")
    rows = lib.vae_gene_table_rows()
    if out is None:
        out = torch.empty(rows, G, dtype=torch.float32, device=px_r.device)
    if out.shape != (rows, G) or not out.is_contiguous():
        raise ValueError(f"vae_amd.ops.gene_table: out must be contiguous {(rows, G)}")
    _lib.check(lib.vae_gene_table(_p(px_r), G, _p(out), _stream()), "gene_table")
    return out


def nbzinb_fwd(zinb: bool, logit_scale: torch.Tensor, logit_dropout: Optional[torch.Tensor], x: torch.Tensor,
               library: torch.Tensor, table: torch.Tensor, streaming: bool = False, out: Optional[torch.Tensor] = None,
               have_stats: bool = False, partials_only: bool = False):
    """Fused softmax + library scale + NB/ZINB log-likelihood + row sum.

    -> (ll (B,), rowmax (B,), rowsum (B,), S (B,)); recon = -ll.  The last three are what
    ``nbzinb_bwd`` needs.  ``streaming``: the (gene strip x cell chunk) form the fused step runs
    (``vae_nbzinb_fwd_ws``) instead of the one-workgroup-per-cell kernel.  ``out``: a (4, B) buffer for the four
    results; with ``have_stats`` its rows 1 and 2 already hold rowmax / rowsum and the statistics launch is skipped;
    ``partials_only`` leaves the sum over gene strips to the caller (ll / S are then not written)."""
    lib = _lib.load()
    l16 = _chk_logits(logit_scale, "logit_scale")
    u8 = _chk_counts(x)
    B, G = x.shape
    if logit_scale.shape != (B, G):
        raise ValueError("vae_amd.ops.nbzinb_fwd: logit_scale shape mismatch")
    if zinb:
        if _chk_logits(logit_dropout, "logit_dropout") != l16 or logit_dropout.shape != (B, G):
            raise ValueError("vae_amd.ops.nbzinb_fwd: logit_dropout shape / dtype mismatch")
    if l16 and not streaming:
        raise ValueError("vae_amd.ops.nbzinb_fwd: bfloat16 logits exist in the streaming form only")
    _chk1d(library, B, "library")
    if table.shape != (lib.vae_gene_table_rows(), G) or not table.is_contiguous():
        raise ValueError("vae_amd.ops.nbzinb_fwd: table must be the gene table made by ops.gene_table")
    o = out if out is not None else torch.empty(4, B, dtype=torch.float32, device=x.device)
    if o.shape != (4, B) or o.dtype != torch.float32 or not o.is_cuda or not o.is_contiguous():
        raise ValueError("vae_amd.ops.nbzinb_fwd: out must be a contiguous float32 GPU tensor of shape (4, B)")
    if have_stats and (out is None or not streaming):
        raise ValueError("vae_amd.ops.nbzinb_fwd: have_stats needs streaming=True and out= holding rowmax / rowsum")
    if streaming:
        nbytes = lib.vae_nbzinb_fwd_workspace_bytes(B, G)
        ws = workspace(nbytes, x.device)
        _lib.check((lib.vae_nbzinb_fwd_ws_bf16 if l16 else lib.vae_nbzinb_fwd_ws)(int(zinb), _p(logit_scale), _ld(logit_scale), _p(logit_dropout if zinb else None),
                                         _ld(logit_dropout) if zinb else 0, _p(x), _lib.X_U8 if u8 else _lib.X_F32, _ld(x),
                                         _p(library), _p(table), B, G, None if partials_only else _p(o[0]), _p(o[1]), _p(o[2]),
                                         None if partials_only else _p(o[3]), int(have_stats), _p(ws), nbytes, _stream()),
                   "nbzinb_fwd_ws")
        return o[0], o[1], o[2], o[3]
    _lib.check((lib.vae_nbzinb_fwd_u8 if u8 else lib.vae_nbzinb_fwd)(int(zinb), _p(logit_scale), _ld(logit_scale), _p(logit_dropout if zinb else None),
                                  _ld(logit_dropout) if zinb else 0, _p(x), _ld(x), _p(library), _p(table), B, G,
                                  _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _stream()), "nbzinb_fwd")
    return o[0], o[1], o[2], o[3]


def px_rate(logit_scale: torch.Tensor, library: torch.Tensor, rowmax: torch.Tensor, rowsum: torch.Tensor,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The decoder mean exp(library) * softmax(logit_scale) (modules.py:401,404) from the row statistics
    ``nbzinb_fwd`` returned -- the value its likelihood used.  -> (B, G)."""
    lib = _lib.load()
    _chk2d(logit_scale, "logit_scale")
    B, G = logit_scale.shape
    for t, name in ((library, "library"), (rowmax, "rowmax"), (rowsum, "rowsum")):
        _chk1d(t, B, name)
    if out is None:
        out = torch.empty(B, G, dtype=torch.float32, device=logit_scale.device)
    _chk2d(out, "out")
    if tuple(out.shape) != (B, G):
        raise ValueError(f"vae_amd.ops.px_rate: out must be {(B, G)}, got {tuple(out.shape)}")
    _lib.check(lib.vae_px_rate(_p(logit_scale), _ld(logit_scale), _p(library), _p(rowmax), _p(rowsum), B, G, _p(out),
                               _ld(out), _stream()), "px_rate")
    return out


def nbzinb_bwd(zinb: bool, logit_scale: torch.Tensor, logit_dropout: Optional[torch.Tensor], x: torch.Tensor,
               library: torch.Tensor, table: torch.Tensor, rowmax, rowsum, S, g_ll: Optional[torch.Tensor],
               g_scalar: float, dlogit_scale: torch.Tensor, dlogit_dropout: Optional[torch.Tensor],
               dlibrary: Optional[torch.Tensor], dpx_r: torch.Tensor, dbias_scale: Optional[torch.Tensor],
               dbias_dropout: Optional[torch.Tensor], deterministic: bool = True) -> None:
    """Gradient of sum_b g_scalar*g_ll[b]*ll[b].  dlogit_* may alias logit_* (in place).
    dpx_r / dbias_* are ACCUMULATED into: through per-chunk slabs summed in a fixed order
    (deterministic=True, default) or with float atomics."""
    lib = _lib.load()
    u8 = _chk_counts(x)
    l16 = _chk_logits(logit_scale, "logit_scale")
    if (u8 or l16) and not deterministic:
        raise ValueError("vae_amd.ops.nbzinb_bwd: uint8 counts / bfloat16 logits use the deterministic (workspace) mode")
    B, G = x.shape
    if table.shape != (lib.vae_gene_table_rows(), G) or not table.is_contiguous():
        raise ValueError("vae_amd.ops.nbzinb_bwd: table must be the gene table made by ops.gene_table")
    if _chk_logits(dlogit_scale, "dlogit_scale") != l16 or (zinb and (_chk_logits(logit_dropout, "logit_dropout") != l16 or
                                                                     _chk_logits(dlogit_dropout, "dlogit_dropout") != l16)):
        raise TypeError("vae_amd.ops.nbzinb_bwd: logits and their gradients must share one dtype (float32 or bfloat16)")
    _chk1d(g_ll, B, "g_ll"); _chk1d(dlibrary, B, "dlibrary"); _chk1d(dpx_r, G, "dpx_r")
    _chk1d(dbias_scale, G, "dbias_scale"); _chk1d(dbias_dropout, G, "dbias_dropout")
    nbytes = lib.vae_nbzinb_bwd_workspace_bytes(B, G) if deterministic else 0
    ws = workspace(nbytes, x.device) if deterministic else None
    args = (int(zinb), _p(logit_scale), _ld(logit_scale), _p(logit_dropout if zinb else None),
            _ld(logit_dropout) if zinb else 0, _p(x), _ld(x), _p(library), _p(table), _p(rowmax),
            _p(rowsum), _p(S), _p(g_ll), float(g_scalar), B, G, _p(dlogit_scale), _ld(dlogit_scale),
            _p(dlogit_dropout if zinb else None), _ld(dlogit_dropout) if zinb else 0, _p(dlibrary),
            _p(dpx_r), _p(dbias_scale), _p(dbias_dropout if zinb else None), _p(ws), nbytes)
    if l16:   # bf16 logits / gradients: dpx_r and the bias gradients are OVERWRITTEN (include/vae_amd.h)
        a = list(args)
        a[5:7] = [_p(x), _lib.X_U8 if u8 else _lib.X_F32, _ld(x)]
        rc = lib.vae_nbzinb_bwd_bf16(*a, _stream())
    else:
        rc = lib.vae_nbzinb_bwd_u8(*args, 0, _stream()) if u8 else lib.vae_nbzinb_bwd(*args, _stream())
    _lib.check(rc, "nbzinb_bwd")


def poisson_fwd(x: torch.Tensor, rate: torch.Tensor) -> torch.Tensor:
    """Element-wise Poisson log-probability xlogy(x, rate) - rate - lgamma(x + 1) (task.py:167-168)."""
    lib = _lib.load()
    _chk2d(x, "x"); _chk2d(rate, "rate")
    if x.shape != rate.shape:
        raise ValueError("vae_amd.ops.poisson_fwd: x and rate must have the same shape")
    B, G = x.shape
    out = torch.empty(B, G, dtype=torch.float32, device=x.device)
    _lib.check(lib.vae_poisson_fwd(_p(x), _ld(x), _p(rate), _ld(rate), B, G, _p(out), G, _stream()), "poisson_fwd")
    return out


def poisson_bwd(gout: torch.Tensor, x: torch.Tensor, rate: torch.Tensor) -> torch.Tensor:
    """d/d rate of ``poisson_fwd`` times ``gout``."""
    lib = _lib.load()
    _chk2d(gout, "gout"); _chk2d(x, "x"); _chk2d(rate, "rate")
    B, G = x.shape
    out = torch.empty(B, G, dtype=torch.float32, device=x.device)
    _lib.check(lib.vae_poisson_bwd(_p(gout), _ld(gout), _p(x), _ld(x), _p(rate), _ld(rate), B, G, _p(out), G, _stream()),
               "poisson_bwd")
    return out


def gather_counts(store: torch.Tensor, cell_rows: torch.Tensor, gene_idx: Optional[torch.Tensor] = None,
                  n_genes: Optional[int] = None, dtype: torch.dtype = torch.uint8) -> torch.Tensor:
    """Rows ``cell_rows`` (device int64) of a device-resident uint8 cells x genes ``store``, restricted to ``gene_idx``
    (device int32) or to the first ``n_genes`` genes, as uint8 or float32 (data.py:406-415).  The indices are NOT
    validated here (they are on the device): rows / genes outside the store come back as zeros."""
    lib = _lib.load()
    if not store.is_cuda or store.dtype != torch.uint8 or store.dim() != 2 or store.stride(1) != 1:
        raise TypeError("vae_amd.ops.gather_counts: store must be a 2-D uint8 GPU tensor with a contiguous last dim")
    if not cell_rows.is_cuda or cell_rows.dtype != torch.int64 or cell_rows.dim() != 1 or not cell_rows.is_contiguous():
        raise TypeError("vae_amd.ops.gather_counts: cell_rows must be a contiguous 1-D int64 GPU tensor")
    if gene_idx is not None:
        if not gene_idx.is_cuda or gene_idx.dtype != torch.int32 or gene_idx.dim() != 1 or not gene_idx.is_contiguous():
            raise TypeError("vae_amd.ops.gather_counts: gene_idx must be a contiguous 1-D int32 GPU tensor")
        G = gene_idx.numel()
    else:
        G = store.shape[1] if n_genes is None else int(n_genes)
    if dtype not in (torch.uint8, torch.float32):
        raise TypeError("vae_amd.ops.gather_counts: dtype must be torch.uint8 or torch.float32")
    B = cell_rows.numel()
    if B == 0 or G == 0:
        raise ValueError("vae_amd.ops.gather_counts: empty batch or empty gene set")
    ld = (G + 3) // 4 * 4 if dtype == torch.uint8 else G      # uint8 rows padded to 4 bytes: the step's vector loads
    buf = torch.empty(B, ld, dtype=dtype, device=store.device)
    out = buf[:, :G]
    _lib.check(lib.vae_gather_counts(_p(store), store.shape[0], store.stride(0), _p(cell_rows), B, _p(gene_idx), G, _p(out),
                                     _lib.X_U8 if dtype == torch.uint8 else _lib.X_F32, ld, _stream()), "gather_counts")
    return out


def mixup_counts(x: torch.Tensor, partner: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """alpha[i] * x[i] + (1 - alpha[i]) * x[partner[i]] as numpy evaluates it (float64, rounded once to float32;
    data.py:473-475).  x: (B, G) uint8 or float32; partner: int32 (B); alpha: float64 (B), all on the GPU."""
    lib = _lib.load()
    u8 = _chk_counts(x, "x")
    B, G = x.shape
    if not partner.is_cuda or partner.dtype != torch.int32 or partner.shape != (B,) or not partner.is_contiguous():
        raise TypeError("vae_amd.ops.mixup_counts: partner must be a contiguous int32 GPU tensor of length B")
    if not alpha.is_cuda or alpha.dtype != torch.float64 or alpha.shape != (B,) or not alpha.is_contiguous():
        raise TypeError("vae_amd.ops.mixup_counts: alpha must be a contiguous float64 GPU tensor of length B")
    out = torch.empty(B, G, dtype=torch.float32, device=x.device)
    _lib.check(lib.vae_mixup_counts(_p(x), _lib.X_U8 if u8 else _lib.X_F32, _ld(x), B, G, _p(partner), _p(alpha), _p(out), G,
                                    _stream()), "mixup_counts")
    return out


def cutmix_counts(x: torch.Tensor, partner: torch.Tensor, take: torch.Tensor, out_dtype=None) -> torch.Tensor:
    """take[i, g] ? x[partner[i], g] : x[i, g] (data.py:431-454, the array part; include/vae_amd.h).  x: (B, G) uint8 or
    float32; partner: int32 (B); take: (B, G) uint8 (nonzero = from the partner); -> uint8 or float32 (``out_dtype``,
    default: x's)."""
    lib = _lib.load()
    u8 = _chk_counts(x, "x")
    B, G = x.shape
    if not partner.is_cuda or partner.dtype != torch.int32 or partner.shape != (B,) or not partner.is_contiguous():
        raise TypeError("vae_amd.ops.cutmix_counts: partner must be a contiguous int32 GPU tensor of length B")
    if not take.is_cuda or take.dtype != torch.uint8 or take.shape != (B, G) or take.stride(1) != 1:
        raise TypeError("vae_amd.ops.cutmix_counts: take must be a (B, G) uint8 GPU tensor with a contiguous last dim")
    out_dtype = out_dtype or x.dtype
    if out_dtype not in (torch.uint8, torch.float32) or (out_dtype == torch.uint8 and not u8):
        raise TypeError("vae_amd.ops.cutmix_counts: out_dtype must be uint8 (uint8 input only) or float32")
    out = torch.empty(B, G, dtype=out_dtype, device=x.device)
    _lib.check(lib.vae_cutmix_counts(_p(x), _lib.X_U8 if u8 else _lib.X_F32, _ld(x), B, G, _p(partner), _p(take), _ld(take), _p(out),
                                     _lib.X_U8 if out_dtype == torch.uint8 else _lib.X_F32, G, _stream()), "cutmix_counts")
    return out


def loglik_elem_fwd(zinb: bool, x, mu, theta, pi) -> torch.Tensor:
    """Element-wise log_nb_positive / log_zinb_positive; theta is (G,) or (B,G)."""
    lib = _lib.load()
    _chk2d(x, "x"); _chk2d(mu, "mu")
    B, G = x.shape
    th2 = theta if theta.dim() == 2 else theta.view(1, -1)
    _chk2d(th2, "theta")
    ld_th = 0 if th2.shape[0] == 1 and B > 1 or theta.dim() == 1 else _ld(th2)
    if zinb:
        _chk2d(pi, "pi")
    out = torch.empty(B, G, dtype=torch.float32, device=x.device)
    _lib.check(lib.vae_loglik_elem_fwd(int(zinb), _p(x), _ld(x), _p(mu), _ld(mu), _p(th2), ld_th, _p(pi if zinb else None),
                                       _ld(pi) if zinb else 0, B, G, _p(out), G, _stream()), "loglik_elem_fwd")
    return out


def loglik_elem_bwd(zinb: bool, gout, x, mu, theta, pi):
    """-> (dmu, dtheta_full (B,G), dpi|None): gradients of sum(gout * loglik)."""
    lib = _lib.load()
    B, G = x.shape
    _chk2d(gout, "gout")
    th2 = theta if theta.dim() == 2 else theta.view(1, -1)
    ld_th = 0 if th2.shape[0] == 1 and B > 1 or theta.dim() == 1 else _ld(th2)
    dmu = torch.empty(B, G, dtype=torch.float32, device=x.device)
    dth = torch.empty(B, G, dtype=torch.float32, device=x.device)
    dpi = torch.empty(B, G, dtype=torch.float32, device=x.device) if zinb else None
    _lib.check(lib.vae_loglik_elem_bwd(int(zinb), _p(gout), _ld(gout), _p(x), _ld(x), _p(mu), _ld(mu), _p(th2), ld_th,
                                       _p(pi if zinb else None), _ld(pi) if zinb else 0, B, G, _p(dmu), _p(dth), _p(dpi),
                                       _stream()), "loglik_elem_bwd")
    return dmu, dth, dpi


def softmax_rate_fwd(logit_scale: torch.Tensor, library: torch.Tensor):
    """-> (px_scale, px_rate) = (softmax(logits), exp(library) * softmax(logits)), both (B, G)."""
    lib = _lib.load()
    _chk2d(logit_scale, "logit_scale")
    B, G = logit_scale.shape
    _chk1d(library, B, "library")
    ps = torch.empty(B, G, dtype=torch.float32, device=logit_scale.device)
    pr = torch.empty(B, G, dtype=torch.float32, device=logit_scale.device)
    _lib.check(lib.vae_softmax_rate_fwd(_p(logit_scale), _ld(logit_scale), _p(library), B, G, _p(ps), _p(pr), _stream()),
               "softmax_rate_fwd")
    return ps, pr


def softmax_rate_bwd(px_scale: torch.Tensor, library: torch.Tensor, d_scale: Optional[torch.Tensor],
                     d_rate: Optional[torch.Tensor]):
    """-> (dlogits (B,G), dlibrary (B,))."""
    lib = _lib.load()
    B, G = px_scale.shape
    for t, n in ((d_scale, "d_scale"), (d_rate, "d_rate")):
        if t is not None and (t.shape != (B, G) or not t.is_contiguous() or t.dtype != torch.float32):
            raise ValueError(f"vae_amd.ops.softmax_rate_bwd: {n} must be contiguous float32 {(B, G)}")
    dl = torch.empty(B, G, dtype=torch.float32, device=px_scale.device)
    dlib = torch.empty(B, dtype=torch.float32, device=px_scale.device)
    _lib.check(lib.vae_softmax_rate_bwd(_p(px_scale), _p(library), _p(d_scale), _p(d_rate), B, G, _p(dl), G, _p(dlib),
                                        _stream()), "softmax_rate_bwd")
    return dl, dlib
