"""End-to-end training rate from the device-resident count store (rows f1 + f2 + the step): every step draws a batch
(torch sampler -> vae_gather_counts [-> vae_mixup_counts]), runs the fused fwd+bwd ELBO and the fused AdamW step.
    python tools/bench_e2e.py [--cells 200000] [--genes 20000] [--batch 512] [--steps 200] [--mixup] [--u8]
Prints one JSON line.  Synthetic study (no labels: gene loss only), cfg2 model."""
import argparse, contextlib, io, json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from vae_amd import ops
from vae_amd.data import CountStore, DataModule
from vae_amd.networks import VAE
from vae_amd.optim import FusedAdamW
from vae_amd.task import LitVAE

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=200_000); ap.add_argument("--genes", type=int, default=20_000)
ap.add_argument("--batch", type=int, default=512); ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--mixup", action="store_true"); ap.add_argument("--u8", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
store = CountStore((torch.rand(a.cells, a.genes, device=dev, generator=g) < 0.08).to(torch.uint8) *
                   torch.randint(1, 6, (a.cells, a.genes), device=dev, generator=g, dtype=torch.uint8))
n = a.cells
meta = {"obs": {"class": np.array(["OPC"] * n), "include_training": np.ones(n, np.int64)},
        "var": {"gene_name": np.array([f"G{i}" for i in range(a.genes)])}}
dm = DataModule(None, None, np.arange(n), np.arange(0), batch_size=a.batch, cell_properties=None, cell_restrictions=None,
                group_balancing=None, mixup=a.mixup, metadata=meta, store=store,
                counts_dtype=torch.uint8 if a.u8 else torch.float32)
with contextlib.redirect_stdout(io.StringIO()):
    dm.train_idx = np.arange(n)
    from vae_amd.data import SingleCellDataset
    ds = SingleCellDataset(None, None, np.arange(n), batch_size=a.batch, cell_restrictions=None, group_balancing=None,
                           mixup=a.mixup, training=True, metadata=meta, store=store,
                           counts_dtype=torch.uint8 if a.u8 else torch.float32)
    dm.train_dataset = ds
    lm, lv = ds.library_size_stats(20_000)
    torch.manual_seed(0)
    net = VAE(n_input=a.genes, n_layers=1, n_hidden=256, n_hidden_decoder=256, n_hidden_library=128, n_latent=30,
              n_latent_cell_decoder=30, dropout_rate=0.0, input_dropout_rate=0.0, grad_reverse_dict={},
              cell_properties={"_": {"discrete": True, "values": [0, 1], "stop_grad": True}}, batch_properties=None)
    task = LitVAE(net, cell_properties=None, library_mean=lm, library_var=lv)
task.train()
arena = task.attach_arena(dev)
opt = FusedAdamW(arena, lr=1e-4, weight_decay=0.0, max_grad_norm=0.5,
                 skip=[k for k in arena.offsets if k.startswith("cell_decoder")])
it = iter(dm.train_dataloader())


def step():
    global it
    try:
        batch = next(it)
    except StopIteration:
        it = iter(dm.train_dataloader()); batch = next(it)
    out = task.training_step_fused(batch)
    opt.step()
    return out


for _ in range(20):
    out = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"what": "end-to-end training step from the HBM count store (sampler + gather" +
                  (" + mixup" if a.mixup else "") + " + fwd/bwd ELBO + AdamW)", "cells_per_s": a.batch / dt,
                  "ms_per_step": dt * 1e3, "batch": a.batch, "store_cells": a.cells, "genes": a.genes,
                  "counts": "uint8" if a.u8 and not a.mixup else "float32", "loss": float(out["loss"])}))
