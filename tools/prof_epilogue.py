"""Micro-driver for profiling the fused epilogue kernels alone at a bench shape:
   rocprofv3 --kernel-trace --stats -- python3 tools/prof_epilogue.py [B] [G] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
lam = torch.exp(torch.randn(G, generator=g, device=dev) * 1.5 - 1.5).clamp(0, 50)
x = torch.poisson(lam.unsqueeze(0) * torch.exp(torch.randn(B, 1, generator=g, device=dev) * 0.25), generator=g)
x = x * (torch.rand(B, G, generator=g, device=dev) >= 0.3)
x[:, 0] += 1
logits = torch.randn(B, 2 * G, generator=g, device=dev)
lib = torch.full((B,), 9.0, device=dev)
table = ops.gene_table(torch.randn(G, generator=g, device=dev))
dl = torch.empty_like(logits)
dlib, dpxr, dbs, dbp = torch.empty(B, device=dev), torch.zeros(G, device=dev), torch.zeros(G, device=dev), torch.zeros(G, device=dev)
det = os.environ.get("DET", "1") == "1"
def t_ms(fn, n=iters):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
# the streaming forward of the step (statistics handed in, per-strip partials left to the caller), then the per-cell kernel
fo = torch.stack(ops.nbzinb_fwd(True, logits[:, :G], logits[:, G:], x, lib, table, streaming=True))
ll, rmax, rsum, S = fo[0], fo[1], fo[2], fo[3]
tf = t_ms(lambda: ops.nbzinb_fwd(True, logits[:, :G], logits[:, G:], x, lib, table, streaming=True, out=fo, have_stats=True,
                                 partials_only=True))
if os.environ.get("PER_CELL") and G <= 32768:
    tpc = t_ms(lambda: ops.nbzinb_fwd(True, logits[:, :G], logits[:, G:], x, lib, table))
    print(f"per-cell forward kernel {tpc*1e3:.1f} us")
tb = t_ms(lambda: ops.nbzinb_bwd(True, logits[:, :G], logits[:, G:], x, lib, table, rmax, rsum, S, None, -1e-5, dl[:, :G], dl[:, G:],
                   dlib, dpxr, dbs, dbp, deterministic=det))
print(f"B={B} G={G} det={det} wg={os.environ.get('VAE_BWD_WORKGROUPS')} fwd {tf*1e3:.1f} us ({12*B*G/tf/1e6:.0f} GB/s)  bwd {tb*1e3:.1f} us ({20*B*G/tb/1e6:.0f} GB/s)")
print("ll finite:", bool(torch.isfinite(ll).all()), "nonzero frac:", float((x > 0).float().mean()))
