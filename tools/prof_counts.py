"""Stand-alone timing of the count-operand products (vae_gemm_counts) at the cfg2 encoder shapes: one f16 term for integer
counts vs the in-kernel two-term fallback (fractional counts) vs the plain two-term kernel (vae_gemm_scaled)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import ops

def t(fn, iters=30):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

B, G, HE = int(os.environ.get("PB", 512)), 20000, 384
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.poisson(torch.rand(B, G, generator=g, device="cuda") * 2)
x[:, 0] += 1
xf = x + 0.37
w = torch.randn(HE, G, generator=g, device="cuda") * 0.01
dp = torch.randn(B, HE, generator=g, device="cuda") * 1e-3
gw = torch.empty(HE, G, device="cuda")
wmax, dmax = w.abs().amax().expand(4).contiguous(), dp.abs().amax().expand(4).contiguous()
cs = dp.sum(0)
for name, xx in (("integer", x), ("fractional", xf)):
    _, xmax, fl, _ = ops.row_stats(xx, w)
    print(name, "flags", float(fl.max()),
          "NT counts %.1f us" % t(lambda: ops.gemm_counts(ops.NT, xx, w, xmax, wmax, fl)),
          "TN counts %.1f us" % t(lambda: ops.gemm_counts(ops.TN, dp, xx, dmax, xmax, fl, row_sub=cs, out=gw)),
          "TN counts no row_sub %.1f us" % t(lambda: ops.gemm_counts(ops.TN, dp, xx, dmax, xmax, fl, out=gw)))
sx, sw, sd = ops.absmax_scale(x), ops.absmax_scale(w), ops.absmax_scale(dp)
print("plain two-term: NT %.1f us  TN %.1f us" % (t(lambda: ops.gemm_scaled(ops.NT, x, w, sx, sw)), t(lambda: ops.gemm_scaled(ops.TN, dp, x, sd, sx, out=gw))))
print("row_stats %.1f us   normalize_rows %.1f us" % (t(lambda: ops.row_stats(x, w)), t(lambda: ops.normalize_rows(x))))
