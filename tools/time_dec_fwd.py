"""Decoder-head forward product (M, 40000, 256), bf16 and two-f16-term modes."""
import sys, torch
sys.path.insert(0, ".")
from vae_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = torch.randn(M, 256, device="cuda"); W = torch.randn(40000, 256, device="cuda") * 0.05; b = torch.randn(40000, device="cuda")
out = torch.empty(M, 40000, device="cuda")
sa, sb = ops.absmax_scale(A), ops.absmax_scale(W)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("bf16  %.1f us" % t(lambda: ops.gemm(ops.NT, A, W, bias=b, out=out, precision=ops.BF16)))
print("f16x2 %.1f us" % t(lambda: ops.gemm_scaled(ops.NT, A, W, sa, sb, bias=b, out=out)))
