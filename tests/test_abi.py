"""CPU: the C-ABI library loads and exports every symbol include/vae_amd.h declares; argument
validation returns error codes without touching a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "vae_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vae_[a-z0-9_]+)\s*\(", src)))


def test_header_and_library_agree():
    from vae_amd import _lib
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vae_amd.h but not exported by libvae_amd.so"
    # every symbol the Python binding uses is declared in the header
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} bound in vae_amd/_lib.py but missing from include/vae_amd.h"


def test_probe_functions():
    from vae_amd import _lib
    lib = _lib.load()
    assert lib.vae_version() >= 100
    assert lib.vae_arch() == b"gfx950"
    assert lib.vae_device_count() >= 0
    assert lib.vae_gene_table_rows() == 69      # 5 per-gene rows + 2 x 32 count-indexed rows
    assert lib.vae_nbzinb_max_genes() >= 30000    # the streaming forward and the step have no limit


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any HIP call; messages come through vae_last_error."""
    from vae_amd import _lib
    lib = _lib.load()
    rc = lib.vae_gemm(7, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, None, 0, None, 0, None)
    assert rc == 1 and b"layout" in lib.vae_last_error()
    rc = lib.vae_gemm(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, None, 0, None, 0, None)
    assert rc == 1 and b"null" in lib.vae_last_error()
    rc = lib.vae_gemm(0, 4, 4, 4, 4, None, 4, None, 4, None, 4, None, None, 0, None, 0, None)
    assert rc == 1 and b"precision" in lib.vae_last_error()
    rc = lib.vae_nbzinb_fwd(1, None, 0, None, 0, None, 0, None, None, 1, 1, None, None, None, None, None)
    assert rc == 1
    with pytest.raises(RuntimeError, match="status 1"):
        _lib.check(rc, "nbzinb_fwd")
    # the input-side entry points (rows f2 / f4): null pointers, batch limit, output type, in-place mixing
    rc = lib.vae_gather_counts(None, 10, 8, None, 4, None, 8, None, 1, 8, None)
    assert rc == 1 and b"null" in lib.vae_last_error()
    fake = C.c_void_p(4096)      # never dereferenced: validation fails first
    rc = lib.vae_gather_counts(fake, 10, 8, fake, 70000, None, 8, fake, 1, 8, None)
    assert rc == 1 and b"65535" in lib.vae_last_error()
    rc = lib.vae_gather_counts(fake, 10, 8, fake, 4, None, 16, fake, 1, 16, None)
    assert rc == 1 and b"gene index" in lib.vae_last_error()
    rc = lib.vae_gather_counts(fake, 10, 8, fake, 4, None, 8, fake, 5, 8, None)
    assert rc == 1 and b"out_dtype" in lib.vae_last_error()
    rc = lib.vae_mixup_counts(fake, 0, 8, 4, 8, fake, fake, fake, 8, None)
    assert rc == 1 and b"in-place" in lib.vae_last_error()
    rc = lib.vae_mixup_counts(fake, 3, 8, 4, 8, fake, fake, C.c_void_p(8192), 8, None)
    assert rc == 1 and b"x_dtype" in lib.vae_last_error()
    # cell-property heads (row a13): head count, widths, latent size, workspace -- all checked before any launch
    h = _lib.CellHeads()
    assert lib.vae_cell_heads_workspace_bytes(64, 30, C.byref(h)) == -1            # no heads
    h.n_heads = 2
    h.width[0], h.discrete[0], h.width[1], h.discrete[1] = 3, 1, 1, 0
    n = lib.vae_cell_heads_workspace_bytes(2048, 30, C.byref(h))
    assert n >= (2048 * 4 + 2048) * 4                                              # logit gradients + per-cell losses at least
    rc = lib.vae_cell_heads_step(C.byref(h), fake, 64, 30, fake, 2, fake, 2, 1.0, None, fake, fake, n, None)
    assert rc == 1 and b"null parameter" in lib.vae_last_error()                   # head weights missing
    for i in range(2):
        h.w[i], h.b[i] = 4096, 4096
    rc = lib.vae_cell_heads_step(C.byref(h), fake, 64, 65, fake, 2, fake, 2, 1.0, None, fake, fake, n, None)
    assert rc == 1 and b"latent" in lib.vae_last_error()
    rc = lib.vae_cell_heads_step(C.byref(h), fake, 64, 30, fake, 1, fake, 2, 1.0, None, fake, fake, n, None)
    assert rc == 1 and b"one column per head" in lib.vae_last_error()
    rc = lib.vae_cell_heads_step(C.byref(h), fake, 64, 30, fake, 2, fake, 2, 1.0, None, fake, fake, 16, None)
    assert rc == 1 and b"workspace" in lib.vae_last_error()
    h.width[1] = 2                                                                  # a continuous head has one output
    rc = lib.vae_cell_heads_step(C.byref(h), fake, 64, 30, fake, 2, fake, 2, 1.0, None, fake, fake, n, None)
    assert rc == 1 and b"bad width" in lib.vae_last_error()
    # the optimiser entry point that keeps bf16 weight images: ranges and image pointers are checked before any launch
    seg = (_lib.AdamWSegment * 1)(_lib.AdamWSegment(0, 64, 0.0))
    rng = (C.c_int64 * 4)(0, 16, 16, 64)
    args = (fake, fake, fake, fake, 64, seg, 1, 1e-3, 0.9, 0.999, 1e-8, 1, 0.0, 1.0, None, 0, None)
    rc = lib.vae_adamw_step_bf16(*args, rng, None, fake, None)
    assert rc == 1 and b"null img_ranges" in lib.vae_last_error()
    rc = lib.vae_adamw_step_bf16(*args, rng, C.c_void_p(4100), fake, None)
    assert rc == 1 and b"8-byte aligned" in lib.vae_last_error()
    rc = lib.vae_adamw_step_bf16(*args, (C.c_int64 * 4)(0, 18, 20, 64), fake, fake, None)
    assert rc == 1 and b"multiples of 4" in lib.vae_last_error()
    rc = lib.vae_adamw_step_bf16(*args, (C.c_int64 * 4)(0, 32, 16, 64), fake, fake, None)
    assert rc == 1 and b"overlap" in lib.vae_last_error()


def test_workspace_queries():
    from vae_amd import _lib
    lib = _lib.load()
    assert lib.vae_gemm_workspace_bytes(4096, 40000, 256) == 0          # enough tiles: no split-K
    assert lib.vae_gemm_workspace_bytes(512, 384, 20000) > 0            # skinny: split over genes
    assert lib.vae_nbzinb_bwd_workspace_bytes(512, 20000) % (5 * 20000 * 4) == 0       # slabs [chunk][3 sums + 2 maxima][G]
    cfg = _lib.StepConfig(B=512, G=20000, H=256, Hl=128, L=30, Hd=256, n_layers=1, zinb=1, precision=0, l_mean=0.0,
                          l_var=1.0, kl_weight=1.0, gene_loss_coeff=5e-3, loss_scale=1.0)
    n = lib.vae_elbo_step_workspace_bytes(C.byref(cfg))
    assert n > 512 * 40000 * 4                                          # holds the logits at least
    offs = [lib.vae_elbo_step_ws_offset(C.byref(cfg), i) for i in range(len(_lib.WS_SLICES))]
    assert offs[_lib.WS_SLICES.index("GENE_THRESHOLD")] == -1           # GeneDropout off: no such slice
    cfg.input_dropout_rate = 0.5                                        # ... on: the workspace grows by xn_z
    assert lib.vae_elbo_step_workspace_bytes(C.byref(cfg)) >= n + 512 * 20000 * 4
    offs = [lib.vae_elbo_step_ws_offset(C.byref(cfg), i) for i in range(len(_lib.WS_SLICES))]
    assert all(o >= 0 for o in offs) and len(set(offs)) == len(offs)
    assert lib.vae_elbo_step_ws_offset(C.byref(cfg), 99) == -1
    cfg.n_layers = 9
    assert lib.vae_elbo_step_workspace_bytes(C.byref(cfg)) == -1


def test_counts_path_query(monkeypatch):
    """vae_elbo_step_counts_path (a host-side decision, no launch): how float counts reach the first-layer encoder products."""
    from vae_amd import _lib
    if os.environ.get("VAE_NO_XOPT") or os.environ.get("VAE_NO_XDIRECT"):
        pytest.skip("A/B knob set")
    lib = _lib.load()
    base = dict(B=512, G=20000, H=256, Hl=128, L=30, Hd=256, n_layers=1, zinb=1, l_mean=0.0, l_var=1.0, kl_weight=1.0,
                gene_loss_coeff=5e-3, loss_scale=1.0)
    path = lambda **kw: lib.vae_elbo_step_counts_path(C.byref(_lib.StepConfig(**{**base, **kw})))  # noqa: E731
    assert path(precision=3) == 2                              # cfg2, two f16 terms: optimistic, no launch ahead of the product
    assert path(precision=3, fractional_counts=1) == 1         # a mixup loader's hint: behind the statistics launch
    assert path(precision=3, G=20008) == 2                     # (a partial last K step is fine)
    assert path(precision=3, G=20006) == 0                     # rows that are not 16-byte multiples: the exact fp32 mode altogether
    assert path(precision=3, x_dtype=1) == 2                   # uint8 counts: the same (the products read the uint8 rows)
    assert path(precision=3, input_dropout_rate=0.5) == 0      # GeneDropout: the z-encoder reads its own masked counts
    assert path(precision=3, n_layers=2) == 0                  # the count operand belongs to the fused middle (one hidden layer)
    assert path(precision=0) == 0 and path(precision=1) == 0   # exact fp32 / bf16 modes
    assert lib.vae_elbo_step_counts_path(None) == -1


def test_ops_refuse_cpu_tensors():
    import torch
    from vae_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.row_mean(torch.ones(2, 8))
