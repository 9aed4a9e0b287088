"""Micro-driver: times the five large GEMMs of the cfg2/cfg3 step.
   python tools/prof_gemm.py [B] [precision 0|1] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
G, H, Hd = 20000, 384, 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.poisson(torch.rand(B, G, generator=g, device=dev) * 2, generator=g); x[:, 0] += 1
w_enc = torch.randn(H, G, generator=g, device=dev) * 0.01
w_heads = torch.randn(2 * G, Hd, generator=g, device=dev) * 0.05
px = torch.randn(B, Hd, generator=g, device=dev)
dpre = torch.randn(B, H, generator=g, device=dev)
logits = torch.randn(B, 2 * G, generator=g, device=dev)
mean = ops.row_mean(x)
gw_heads, gw_enc = torch.empty_like(w_heads), torch.empty_like(w_enc)
def t_ms(fn, n=iters):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
cases = {
  "enc_fwd   NT+norm": (lambda: ops.gemm(ops.NT, x, w_enc, x_mean=mean, precision=P), 2.0*B*H*G),
  "dec_fwd   NT     ": (lambda: ops.gemm(ops.NT, px, w_heads, out=logits, precision=P), 2.0*B*2*G*Hd),
  "dec_wgrad TN     ": (lambda: ops.gemm(ops.TN, logits, px, out=gw_heads, precision=P), 2.0*B*2*G*Hd),
  "dec_dgrad NN     ": (lambda: ops.gemm(ops.NN, logits, w_heads, precision=P), 2.0*B*2*G*Hd),
  "enc_wgrad TN+norm": (lambda: ops.gemm(ops.TN, dpre, x, out=gw_enc, x_mean=mean, precision=P), 2.0*B*H*G),
  "small NT 512x31x256": (lambda: ops.gemm(ops.NT, px, w_heads[:31], precision=P), 2.0*B*31*Hd),
  "small TN 31x256x512": (lambda: ops.gemm(ops.TN, logits[:, :31], px, precision=P), 2.0*B*31*Hd),
}
only = os.environ.get("ONLY")
for k, (fn, fl) in cases.items():
    if only and only not in k: continue
    t = t_ms(fn)
    print(f"B={B} P={P} {k}: {t*1e3:8.1f} us  {fl/t/1e9:8.1f} TFLOP/s")
