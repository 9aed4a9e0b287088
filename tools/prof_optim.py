"""Times the optimiser step at the cfg2 parameter count: fused vae_adamw_step vs clip_grad_norm_ + torch.optim.AdamW."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_amd.arena import ParamArena
from vae_amd.networks import VAE
from vae_amd.optim import FusedAdamW, reference_weight_decay
torch.manual_seed(43)
net = VAE(20000, n_hidden=256, n_hidden_decoder=256, n_hidden_library=128, n_latent=30, n_latent_cell_decoder=30, n_layers=1,
          dropout_rate=0.0, input_dropout_rate=0.0, cell_properties={"A": {"discrete": True, "values": [0, 1], "stop_grad": False}},
          grad_reverse_dict={}, reconstruction_loss="zinb")
arena = ParamArena(dict(net.named_parameters()), 1, "zinb", "cuda")
arena.flat_grad.normal_(0, 1e-3)
def t_ms(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
fused = FusedAdamW(arena, lr=5e-4, weight_decay=0.1)
n_act = sum(e - b for b, e, _ in fused.ranges)
t = t_ms(fused.step)
print(f"fused clip+AdamW: {t*1e3:.1f} us for {n_act/1e6:.1f} M parameters = {n_act*32/t/1e6:.0f} GB/s (32 B/param)")
names = [k for k in arena.offsets if k not in arena.no_grad]
tp = []
for k in names:
    p = torch.nn.Parameter(arena.p(k)); p.grad = arena.g(k); tp.append((k, p))
groups = {}
for k, p in tp: groups.setdefault(reference_weight_decay(k, 0.1), []).append(p)
opt = torch.optim.AdamW([{"params": v, "weight_decay": w} for w, v in groups.items()], lr=5e-4, betas=(0.9, 0.998), eps=1e-7)
def torch_step():
    torch.nn.utils.clip_grad_norm_([p for _, p in tp], 0.5)
    opt.step()
t2 = t_ms(torch_step)
print(f"torch clip_grad_norm_ + AdamW (foreach): {t2*1e3:.1f} us  -> fused is {t2/t:.1f}x faster")
