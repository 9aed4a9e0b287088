"""Achievable HBM stream rates on this box (calibration for the HBM-bound kernels): fill, copy, read-reduce."""
import torch
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (82, 655, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    s = t(lambda: a.fill_(1.0)); print(f"{mb:5d} MB fill   {mb / 1024 / s / 1e3 * 1.048576:6.2f} TB/s", flush=True)
    s = t(lambda: b.copy_(a));   print(f"{mb:5d} MB copy   {2 * mb / 1024 / s / 1e3 * 1.048576:6.2f} TB/s (read + write)", flush=True)
    s = t(lambda: a.sum());      print(f"{mb:5d} MB sum    {mb / 1024 / s / 1e3 * 1.048576:6.2f} TB/s", flush=True)
