"""Rates of the input-side kernels (rows f2 / f4): python tools/bench_data.py [cells_in_store] [genes]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vae_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
store = torch.randint(0, 4, (n, G), dtype=torch.uint8, device="cuda")
sub = torch.sort(torch.randperm(G, device="cuda")[: G * 3 // 4]).values.int()


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


for B in (512, 4096):
    rows = torch.randint(0, n, (B,), device="cuda")
    part = torch.randint(0, B, (B,), device="cuda").int()
    alpha = torch.rand(B, device="cuda", dtype=torch.float64)
    x8 = ops.gather_counts(store, rows)
    xf = ops.gather_counts(store, rows, dtype=torch.float32)
    for name, fn, nbytes in (
        ("gather u8->u8", lambda: ops.gather_counts(store, rows), 2 * B * G),
        ("gather u8->f32", lambda: ops.gather_counts(store, rows, dtype=torch.float32), 5 * B * G),
        ("gather subset u8->u8", lambda: ops.gather_counts(store, rows, sub), 2 * B * sub.numel()),
        ("mixup u8->f32", lambda: ops.mixup_counts(x8, part, alpha), 6 * B * G),
        ("mixup f32->f32", lambda: ops.mixup_counts(xf, part, alpha), 12 * B * G),
    ):
        us = t(fn)
        print(f"B={B:5d} {name:22s} {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s", flush=True)
