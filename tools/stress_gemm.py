#!/usr/bin/env python3
"""Random-shape stress of the tile-pipeline GEMMs (pipelined K loop, K tails, ragged tiles, row strides, split K) against
fp64: python tools/stress_gemm.py [n_cases] [seed].  Not part of the test suite (minutes of fp64 reference work)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from vae_amd import ops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = {}
for case in range(n_cases):
    layout = int(rng.integers(0, 3))
    M = int(rng.choice([128, 132, 252, 384, 512, 700, 1000])) if rng.random() < 0.7 else 4 * int(rng.integers(40, 600))
    N = int(rng.choice([128, 160, 256, 384, 644, 1280])) if rng.random() < 0.7 else 4 * int(rng.integers(40, 500))
    K = int(rng.choice([256, 512, 1024, 4096, 20000])) if rng.random() < 0.5 else 4 * int(rng.integers(16, 3000))
    pad_a, pad_b = 4 * int(rng.integers(0, 3)), 4 * int(rng.integers(0, 3))
    g = torch.Generator().manual_seed(case)
    shape_a = (M, K) if layout != ops.TN else (K, M)
    shape_b = (N, K) if layout == ops.NT else (K, N)
    A = torch.randn(shape_a[0], shape_a[1] + pad_a, generator=g)[:, :shape_a[1]] * 10 ** float(rng.uniform(-4, 2))
    Bm = torch.randn(shape_b[0], shape_b[1] + pad_b, generator=g)[:, :shape_b[1]] * 10 ** float(rng.uniform(-4, 2))
    a64, b64 = A.double(), Bm.double()
    ref = (a64 @ b64.T) if layout == ops.NT else (a64 @ b64) if layout == ops.NN else (a64.T @ b64)
    absref = (a64.abs() @ b64.abs().T) if layout == ops.NT else (a64.abs() @ b64.abs()) if layout == ops.NN else (a64.abs().T @ b64.abs())
    Ad, Bd = A.cuda(), Bm.cuda()   # keeps the row padding (non-contiguous rows)
    Ad = torch.empty(shape_a[0], shape_a[1] + pad_a, device="cuda")[:, :shape_a[1]].copy_(A)
    Bd = torch.empty(shape_b[0], shape_b[1] + pad_b, device="cuda")[:, :shape_b[1]].copy_(Bm)
    for name, fn, tol in (("f16x2", lambda: ops.gemm_scaled(layout, Ad, Bd), 1e-6),
                          ("bf16x3", lambda: ops.gemm(layout, Ad, Bd, precision=ops.FP32), 1e-6),
                          ("bf16", lambda: ops.gemm(layout, Ad, Bd, precision=ops.BF16), 2e-2)):
        out = fn().double().cpu()
        err = float(((out - ref).abs() / absref.clamp_min(1e-300)).max())   # relative to sum |a||b|: the GEMM's own scale
        worst[name] = max(worst.get(name, 0.0), err)
        status = "ok" if err < tol else "FAIL"
        if status == "FAIL" or case % 10 == 0:
            print(f"case {case} layout {layout} M {M} N {N} K {K} pads {pad_a},{pad_b} {name}: {err:.2e} {status}", flush=True)
        if status == "FAIL":
            sys.exit(1)
print("worst", worst)
