"""Training entry point with the reference's ``main`` signature (src/train.py:17-65), without
Lightning and without the authors' data files: the count matrix is passed in (``counts=``), or a
``vae_amd.data.DataModule`` over a device-resident uint8 store (``datamodule=``; train.py:23-37 builds the same
object from ``dataset_cfg``).

The loop is the reference's: seed 43, VAE(**model_cfg), library prior from log1p(row sums)
(data.py:233-244), AdamW + warm-up/decay schedule (task.py:532-565), gradient-norm clip 0.5
(train.py:54), ``max_steps``.  Each step is ONE fused fwd+bwd pass on the HIP kernels; with
several processes (torch.distributed initialised) gradients are averaged over ranks."""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .config import model_cfg as _model_cfg, task_cfg as _task_cfg, trainer_cfg as _trainer_cfg
from .networks import VAE
from .optim import FusedAdamW
from .parallel import GradAllReducer, broadcast_parameters
from .task import LitVAE


def rank_seed(base: int = 43) -> int:
    """Seed of a rank's batch sampler: ``base`` (pl.seed_everything(43), train.py:20) on one process, ``base + rank``
    under torch.distributed -- every rank must draw its OWN cells (Lightning's DDP shards the sampler per rank,
    train.py:57); with one seed for all, N GPUs would average N copies of the same gradient."""
    return base + (dist.get_rank() if dist.is_initialized() else 0)


def library_size_stats(counts: torch.Tensor, n: int = 200_000):
    """mean / variance of log1p(row sums) over at most n cells (data.py:233-244)."""
    lib = torch.log1p(counts[:n].sum(dim=1).double())
    return float(lib.mean()), float(lib.var(unbiased=False))


def main(train_idx: List[int], test_idx: List[int], cell_class: str, additional_restriction: Optional[str] = None, *,
         counts: torch.Tensor = None, cell_prop_vals: Optional[torch.Tensor] = None,
         cell_mask: Optional[torch.Tensor] = None, cell_properties: Optional[Dict] = None,
         model_cfg: Optional[Dict] = None, task_cfg: Optional[Dict] = None, trainer_cfg: Optional[Dict] = None,
         batch_size: int = 256, max_steps: int = 50_000, device: str = "cuda", log_every: int = 0,
         fused_optimizer: bool = True, datamodule=None):
    """Train on ``counts[train_idx]`` (cells x genes, fp32, on ``device``); returns the LitVAE task.

    ``test_idx`` / ``cell_class`` / ``additional_restriction`` are accepted for signature parity;
    dataset restriction and validation belong to the data layer, which is out of scope here."""
    if datamodule is not None:
        return _main_datamodule(datamodule, model_cfg, task_cfg, trainer_cfg, max_steps, log_every, fused_optimizer)
    if counts is None:
        raise ValueError("vae_amd.train.main: pass the count matrix (cells x genes) as `counts`, or a `datamodule`")
    torch.manual_seed(43)  # pl.seed_everything(43), train.py:20
    mcfg, tcfg, rcfg = dict(_model_cfg), dict(_task_cfg), dict(_trainer_cfg)
    mcfg.update(model_cfg or {}); tcfg.update(task_cfg or {}); rcfg.update(trainer_cfg or {})
    cell_properties = cell_properties or {"_": {"discrete": True, "values": [0, 1], "stop_grad": True}}
    mcfg.setdefault("grad_reverse_dict", {})
    mcfg["grad_reverse_dict"] = {k: v for k, v in mcfg["grad_reverse_dict"].items() if k in cell_properties}
    x_all = counts[torch.as_tensor(train_idx, device=counts.device)] if train_idx is not None else counts
    x_all = x_all.to(device=device, dtype=torch.float32)
    network = VAE(n_input=x_all.shape[1], cell_properties=cell_properties, batch_properties=None, **mcfg)
    lib_mean, lib_var = library_size_stats(x_all)
    precision = ops.BF16 if str(rcfg["precision"]).startswith("bf16") else ops.F16X2
    use_heads = cell_prop_vals is not None
    task = LitVAE(network, cell_properties=cell_properties if use_heads else None, library_mean=lib_mean,
                  library_var=lib_var, precision=precision,
                  **{k: v for k, v in tcfg.items() if k not in ("batch_properties",)})
    task.train()  # dropout / gene-dropout run on the fused step's own Philox stream (include/vae_amd.h, "dropout")
    if dist.is_initialized():
        task.dropout_seed += 0x9E3779B97F4A7C15 * dist.get_rank() % (2 ** 63)   # every rank its own masks
    arena = task.attach_arena(device)
    broadcast_parameters(arena)
    reducer = (GradAllReducer(arena, payload="bf16" if precision == ops.BF16 else "fp32",
                              enc_slabs=int(rcfg.get("enc_grad_slabs", 0)))
               if dist.is_initialized() and dist.get_world_size() > 1 else None)
    # optimiser: clip 0.5 + AdamW + warm-up / inverse-sqrt schedule, one fused call per step (optim.py);
    # fused_optimizer=False keeps torch.optim.AdamW + clip_grad_norm_ exactly as configure_optimizers builds them
    fused_opt = None
    if fused_optimizer:
        warm = float(tcfg["warmup_steps"])
        fused_opt = FusedAdamW(arena, lr=tcfg["learning_rate"], weight_decay=tcfg["weight_decay"],
                               max_grad_norm=rcfg["grad_clip_value"],
                               lr_lambda=lambda s: (warm ** 0.5) * min((s + 1) ** -0.5, (s + 1) * warm ** -1.5),
                               skip=() if use_heads else [k for k in arena.offsets if k.startswith("cell_decoder")],
                               track_weight_maxima=precision == ops.F16X2 and x_all.shape[1] % 4 == 0,
                               track_bf16_images=precision == ops.BF16 and x_all.shape[1] % 8 == 0)
        task.weight_maxima_source = fused_opt    # the optimiser keeps the GEMM operand scales / bf16 images of the weights current
    else:
        opt_cfg = task.configure_optimizers()
        opt, sched = opt_cfg["optimizer"], opt_cfg["lr_scheduler"]["scheduler"]
    n = x_all.shape[0]
    gen = torch.Generator(device=device).manual_seed(rank_seed(43))
    params = [p for p in network.parameters() if p.grad is not None]
    steps_per_epoch = max(1, -(-n // batch_size))
    for step in range(max_steps):
        task.current_epoch = step // steps_per_epoch      # Lightning advances it per pass over the data (task.py:213-220)
        idx = torch.randint(0, n, (batch_size,), generator=gen, device=device)
        batch = (x_all[idx], None if not use_heads else cell_prop_vals[idx], None if not use_heads else cell_mask[idx],
                 None, None, idx, None, None)
        out = task.training_step_fused(batch, reducer=reducer)
        if reducer is not None:
            reducer.finish()
        if fused_opt is not None:
            fused_opt.step()
        else:
            torch.nn.utils.clip_grad_norm_(params, rcfg["grad_clip_value"])
            opt.step()
            sched.step()
        if log_every and step % log_every == 0:
            print(f"step {step}: loss {float(out['loss']):.5f} gene_loss {float(out['gene_loss']):.3f}")
    return task


def _main_datamodule(dm, model_cfg, task_cfg, trainer_cfg, max_steps, log_every, fused_optimizer):
    """train.py:17-65 over a ``vae_amd.data.DataModule``: seeds (pl.seed_everything(43) seeds python, numpy and
    torch), ``dm.setup``, the library prior from the training set, VAE + LitVAE, then epochs of the train loader
    until ``max_steps``; every batch is a device gather (+ mixup) followed by one fused step."""
    import random
    random.seed(43); np.random.seed(43); torch.manual_seed(43)
    mcfg, tcfg, rcfg = dict(_model_cfg), dict(_task_cfg), dict(_trainer_cfg)
    mcfg.update(model_cfg or {}); tcfg.update(task_cfg or {}); rcfg.update(trainer_cfg or {})
    dm.setup("train")
    ds = dm.train_dataset
    lib_mean, lib_var = ds.library_size_stats()
    use_heads = bool(dm.cell_properties)
    cell_properties = dm.cell_properties if use_heads else {"_": {"discrete": True, "values": [0, 1], "stop_grad": True}}
    mcfg["grad_reverse_dict"] = {k: v for k, v in mcfg.get("grad_reverse_dict", {}).items() if k in cell_properties}
    network = VAE(n_input=ds.n_genes, cell_properties=cell_properties, batch_properties=dm.batch_properties, **mcfg)
    precision = ops.BF16 if str(rcfg["precision"]).startswith("bf16") else ops.F16X2
    task = LitVAE(network, cell_properties=cell_properties if use_heads else None, library_mean=lib_mean,
                  library_var=lib_var, precision=precision, batch_properties=dm.batch_properties,
                  **{k: v for k, v in tcfg.items() if k not in ("batch_properties",)})
    task.train()
    task.fractional_counts = bool(getattr(ds, "mixup", False) or getattr(ds, "cutmix", False))   # a hint for the fused step
    if dist.is_initialized():
        task.dropout_seed += 0x9E3779B97F4A7C15 * dist.get_rank() % (2 ** 63)
        np.random.seed(rank_seed(43))              # every rank draws its own balanced batches / mixup partners
    sampler_gen = torch.Generator().manual_seed(rank_seed(43))   # ... and its own RandomSampler permutations
    arena = task.attach_arena(ds.device)
    broadcast_parameters(arena)
    reducer = (GradAllReducer(arena, payload="bf16" if precision == ops.BF16 else "fp32",
                              enc_slabs=int(rcfg.get("enc_grad_slabs", 0)))
               if dist.is_initialized() and dist.get_world_size() > 1 else None)
    if fused_optimizer:
        warm = float(tcfg["warmup_steps"])
        opt = FusedAdamW(arena, lr=tcfg["learning_rate"], weight_decay=tcfg["weight_decay"],
                         max_grad_norm=rcfg["grad_clip_value"],
                         lr_lambda=lambda s: (warm ** 0.5) * min((s + 1) ** -0.5, (s + 1) * warm ** -1.5),
                         skip=() if use_heads else [k for k in arena.offsets if k.startswith("cell_decoder")],
                         track_weight_maxima=precision == ops.F16X2 and ds.n_genes % 4 == 0,
                         track_bf16_images=precision == ops.BF16 and ds.n_genes % 8 == 0)
        task.weight_maxima_source = opt
        sched = None
    else:
        opt_cfg = task.configure_optimizers()
        opt, sched = opt_cfg["optimizer"], opt_cfg["lr_scheduler"]["scheduler"]
    params = [p for p in network.parameters() if p.grad is not None]
    step = 0
    while step < max_steps:
        n_before = step
        for batch in dm.train_dataloader(generator=sampler_gen):
            out = task.training_step_fused(batch, reducer=reducer)
            if reducer is not None:
                reducer.finish()
            if sched is None:
                opt.step()
            else:
                torch.nn.utils.clip_grad_norm_(params, rcfg["grad_clip_value"])
                opt.step()
                sched.step()
            if log_every and step % log_every == 0:
                print(f"step {step}: loss {float(out['loss']):.5f} gene_loss {float(out['gene_loss']):.3f}")
            step += 1
            if step >= max_steps:
                break
        task.current_epoch += 1                      # one pass over the train loader = one epoch (task.py:213-220)
        if step == n_before:
            raise ValueError("vae_amd.train.main: the training loader yields no batch (fewer cells than one batch)")
    return task
