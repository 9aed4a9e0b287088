"""Times the five large GEMMs of the step on pre-normalised counts (no x_mean), kernel + reduce.
   python tools/prof_gemm2.py [B] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import ops, _lib
if os.environ.get("VAE_LIB"):   # an experimental build of the library
    _lib.LIB_PATH = os.path.abspath(os.environ["VAE_LIB"])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P = int(sys.argv[3]) if len(sys.argv) > 3 else 0
G, H, Hd = 20000, 384, 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
xn = torch.randn(B, G, generator=g, device=dev)
w_enc = torch.randn(H, G, generator=g, device=dev) * 0.01
w_heads = torch.randn(2 * G, Hd, generator=g, device=dev) * 0.05
px = torch.randn(B, Hd, generator=g, device=dev)
dpre = torch.randn(B, H, generator=g, device=dev)
logits = torch.randn(B, 2 * G, generator=g, device=dev)
gw_heads, gw_enc = torch.empty_like(w_heads), torch.empty_like(w_enc)
def t_ms(fn, n=iters):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
cases = {
  "enc_fwd   NT": (lambda: ops.gemm(ops.NT, xn, w_enc, precision=P), 2.0*B*H*G),
  "dec_fwd   NT": (lambda: ops.gemm(ops.NT, px, w_heads, out=logits, precision=P), 2.0*B*2*G*Hd),
  "dec_wgrad TN": (lambda: ops.gemm(ops.TN, logits, px, out=gw_heads, precision=P), 2.0*B*2*G*Hd),
  "dec_dgrad NN": (lambda: ops.gemm(ops.NN, logits, w_heads, precision=P), 2.0*B*2*G*Hd),
  "enc_wgrad TN": (lambda: ops.gemm(ops.TN, dpre, xn, out=gw_enc, precision=P), 2.0*B*H*G),
}
if os.environ.get("SCALED"):   # two scaled f16 terms, three products
    sc = {k: ops.absmax_scale(t) for k, t in dict(xn=xn, w_enc=w_enc, w_heads=w_heads, px=px, dpre=dpre, logits=logits).items()}
    cases = {
      "enc_fwd   NT": (lambda: ops.gemm_scaled(ops.NT, xn, w_enc, sc["xn"], sc["w_enc"]), 2.0*B*H*G),
      "dec_fwd   NT": (lambda: ops.gemm_scaled(ops.NT, px, w_heads, sc["px"], sc["w_heads"], out=logits), 2.0*B*2*G*Hd),
      "dec_wgrad TN": (lambda: ops.gemm_scaled(ops.TN, logits, px, sc["logits"], sc["px"], out=gw_heads), 2.0*B*2*G*Hd),
      "dec_dgrad NN": (lambda: ops.gemm_scaled(ops.NN, logits, w_heads, sc["logits"], sc["w_heads"]), 2.0*B*2*G*Hd),
      "enc_wgrad TN": (lambda: ops.gemm_scaled(ops.TN, dpre, xn, sc["dpre"], sc["xn"], out=gw_enc), 2.0*B*H*G),
    }
only = os.environ.get("ONLY")
tot = 0
for k, (fn, fl) in cases.items():
    if only and only not in k: continue
    t = t_ms(fn); tot += t
    print(f"B={B} P={P} {k}: {t*1e3:8.1f} us  {fl/t/1e9:8.1f} TFLOP/s")
print(f"total {tot*1e3:.1f} us")
