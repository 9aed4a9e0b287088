"""Training task with the reference's LitVAE interface (src/task.py:29-565), without Lightning.

``training_step(batch, batch_idx)`` follows task.py:453-502 through the module API (autograd
over the HIP ops) and returns the loss; ``training_step_fused(batch)`` is the production fast
path: one fused fwd+bwd pass (vae_amd.step.ElboStep) that leaves gradients in the arena.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Normal, kl_divergence as kl
from torch.optim.lr_scheduler import LambdaLR

from . import ops
from .arena import ParamArena
from .distributions import NegativeBinomial, Poisson, ZeroInflatedNegativeBinomial
from .step import CellHeads, ElboStep


def class_balance_weights(cell_properties: Dict[str, Any]) -> Dict[str, torch.Tensor]:
    """Per discrete property the ``weight`` of its cross-entropy under ``balance_classes`` (task.py:129-132):
    ``clip(1 / freq, 0.1, 10)`` as float32, from the property's ``freq`` entry (data.py fills it)."""
    out = {}
    for k, prop in cell_properties.items():
        if prop["discrete"]:
            if "freq" not in prop:
                raise ValueError(f"balance_classes: cell property {k!r} has no 'freq' entry (class frequencies)")
            out[k] = torch.from_numpy(np.float32(np.clip(1 / np.asarray(prop["freq"]), 0.1, 10.0)))
    return out


def cell_loss(cell_pred: Dict[str, torch.Tensor], cell_properties: Dict[str, Any], cell_prop_vals: torch.Tensor,
              cell_mask: torch.Tensor, weights=1.0, class_weights: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """LitVAE._cell_loss without GroupDRO (task.py:419-451): masked cross-entropy (ignore_index
    -100; per-class ``class_weights`` under balance_classes, task.py:129-132) / MSE per property, each averaged over
    ALL cells of the batch."""
    total = 0.0
    for n, (k, prop) in enumerate(cell_properties.items()):
        if prop["discrete"]:
            cw = (class_weights or {}).get(k)
            if cw is not None:
                cw = cw.to(device=cell_pred[k].device, dtype=cell_pred[k].dtype)
            per = nn.functional.cross_entropy(cell_pred[k], cell_prop_vals[:, n].to(torch.int64), weight=cw, reduction="none",
                                              ignore_index=-100)
        else:
            per = (torch.squeeze(cell_pred[k]) - cell_prop_vals[:, n]) ** 2
        total = total + (weights * per * cell_mask[:, n]).mean()
    return total


class WarmupConstantAndDecaySchedule(LambdaLR):
    """sqrt(w) * min((s+1)^-0.5, (s+1) * w^-1.5) (task.py:662-676)."""

    def __init__(self, optimizer, warmup_steps, last_epoch=-1):
        self.warmup_steps = warmup_steps
        self.alpha = np.sqrt(warmup_steps)
        super().__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        return self.alpha * np.minimum((step + 1) ** (-0.5), (step + 1) * (self.warmup_steps ** -1.5))


class LitVAE(nn.Module):
    def __init__(
        self,
        network,
        cell_properties: Optional[Dict[str, Any]] = None,
        batch_properties: Optional[Dict[str, Any]] = None,
        balance_classes: bool = False,
        dispersion: str = "gene",
        reconstruction_loss: str = "zinb",
        latent_distribution: str = "normal",
        n_epochs_kl_warmup: Optional[int] = 20,
        learning_rate: float = 0.0005,
        warmup_steps: float = 2000.0,
        weight_decay: float = 0.1,
        l1_lambda: float = 0.0,
        gene_loss_coeff: float = 5e-4,
        save_gene_vals: bool = False,
        library_mean: Optional[float] = None,
        library_var: Optional[float] = None,
        subject_batch_size: int = 1,
        use_gdro: bool = False,
        precision: int = ops.F16X2,
    ):
        super().__init__()
        self.network = network
        self.cell_properties = cell_properties
        self.batch_properties = batch_properties
        self.balance_classes = balance_classes
        self.dispersion = dispersion
        self.reconstruction_loss = reconstruction_loss
        self.latent_distribution = latent_distribution
        self.n_epochs_kl_warmup = n_epochs_kl_warmup
        self.learning_rate = learning_rate
        self.warmup_steps = warmup_steps
        self.weight_decay = weight_decay
        self.l1_lambda = l1_lambda
        self.gene_loss_coeff = gene_loss_coeff
        self.save_gene_vals = save_gene_vals
        self.library_mean = library_mean
        self.library_var = library_var
        self.subject_batch_size = subject_batch_size
        self.use_gdro = use_gdro
        self.precision = precision
        self.dropout_seed = int(torch.initial_seed()) & (2 ** 63 - 1)   # key of the fused step's dropout stream
        self.current_epoch = 0
        self.train_step = 0
        self.logged: Dict[str, float] = {}
        self._arena: Optional[ParamArena] = None
        self._fused: Optional[ElboStep] = None
        self._fused_val: Optional[ElboStep] = None
        self._l1 = None
        # a FusedAdamW(track_weight_maxima=True) or (track_bf16_images=True): the fused step then skips its weight scan /
        # its cast of the two gene-sized weight matrices
        self.weight_maxima_source = None
        self._create_results_dict()
        # balance_classes (task.py:129-132): class weights of every discrete head's cross-entropy
        self.class_weights = class_balance_weights(cell_properties) if (balance_classes and cell_properties) else None
        # GroupDRO re-weighting of two heads' per-cell losses by donor group (task.py:84-101; off in the shipped config)
        self.gdro = None
        if use_gdro:
            from .losses import GroupDRO
            paired = [[0, 1, 60, 61], [10, 11, 50, 51], [20, 21, 40, 41], [30, 31]]
            self.gdro = {"Dementia": GroupDRO(balances=[[g for pair in paired for g in pair]]),
                         "BRAAK_AD": GroupDRO(balances=paired)}

    # -- bookkeeping the reference delegates to Lightning ---------------------------------
    def log(self, name, value, **_):
        self.logged[name] = value

    @property
    def kl_weight(self):
        if self.n_epochs_kl_warmup is not None:
            return float(np.clip((1 + self.current_epoch) / self.n_epochs_kl_warmup, 0.0, 1.0))
        return 1.0

    def library_stats(self, gene_vals):
        """Prior of the library-size KL (task.py:222-233)."""
        if self.library_mean is None or self.library_var is None:
            lib = torch.log(gene_vals.sum(dim=1))
            return lib.mean(), torch.clip(lib.var(), 0.1, 1e3)
        return (torch.from_numpy(np.array(self.library_mean, dtype=np.float32)),
                torch.from_numpy(np.array(self.library_var, dtype=np.float32)))

    # -- module-API path (autograd over HIP ops) --------------------------------------------
    def get_reconstruction_loss(self, x, px_rate, px_r, px_dropout, **kwargs) -> torch.Tensor:
        if self.reconstruction_loss == "zinb":
            return -ZeroInflatedNegativeBinomial(mu=px_rate, theta=px_r, zi_logits=px_dropout).log_prob(x).sum(dim=-1)
        if self.reconstruction_loss == "nb":
            return -NegativeBinomial(mu=px_rate, theta=px_r).log_prob(x).sum(dim=-1)
        if self.reconstruction_loss == "poisson":
            return -Poisson(px_rate).log_prob(x).sum(dim=-1)
        raise ValueError(f"unknown reconstruction_loss {self.reconstruction_loss!r}")

    def gene_loss(self, x, qz_m, qz_v, ql_m, ql_v, local_l_mean, local_l_var, px_rate, px_r, px_dropout,
                  training: bool = True):
        kl_z = kl(Normal(qz_m, torch.sqrt(qz_v)), Normal(torch.zeros_like(qz_m), torch.ones_like(qz_v))).sum(dim=1)
        local_l_mean = torch.as_tensor(local_l_mean, dtype=ql_m.dtype, device=ql_m.device)
        local_l_var = torch.as_tensor(local_l_var, dtype=ql_m.dtype, device=ql_m.device)
        kl_l = kl(Normal(ql_m, torch.sqrt(ql_v)), Normal(local_l_mean, torch.sqrt(local_l_var))).sum(dim=1)
        recon_loss = self.get_reconstruction_loss(x, px_rate, px_r, px_dropout)
        loss = recon_loss.mean() + self.kl_weight * (kl_l + kl_z).mean()
        suffix = "" if training else "_val"
        self.log("recon_loss" + suffix, recon_loss.mean())
        self.log("kl_z" + suffix, kl_z.mean())
        return loss, kl_z

    def _cell_loss(self, cell_pred, cell_prop_vals, cell_mask, group_idx):
        if not self.gdro:
            return cell_loss(cell_pred, self.cell_properties, cell_prop_vals, cell_mask, class_weights=self.class_weights)
        # task.py:419-451 with GroupDRO: the heads it tracks weight every cell's loss by its donor group's running loss
        # (from training step 500 on) and feed the UNWEIGHTED per-cell losses back into the running statistics -- a host
        # round trip per step, as in the reference (`loss.tolist()`)
        total = 0.0
        for n, (k, prop) in enumerate(self.cell_properties.items()):
            tracked = k in self.gdro
            w = self.gdro[k].get_weights(group_idx).to(cell_pred[k].device) if tracked and self.train_step >= 500 else 1.0
            if prop["discrete"]:
                cw = (self.class_weights or {}).get(k)
                if cw is not None:
                    cw = cw.to(device=cell_pred[k].device, dtype=cell_pred[k].dtype)
                per = nn.functional.cross_entropy(cell_pred[k], cell_prop_vals[:, n].to(torch.int64), weight=cw,
                                                  reduction="none", ignore_index=-100)
            else:
                per = (torch.squeeze(cell_pred[k]) - cell_prop_vals[:, n]) ** 2
            total = total + (w * per * cell_mask[:, n]).mean()
            if tracked:
                self.gdro[k].update_batch(per, cell_mask[:, n], group_idx)
                self.gdro[k].update_weights()
        return total

    def training_step(self, batch, batch_idx):
        self.train_step += 1
        gene_vals, cell_prop_vals, cell_mask, batch_labels, batch_mask, _, group_idx, _ = batch
        net = self.network
        qz_m, qz_v, z = net.z_encoder(gene_vals, batch_labels, batch_mask)
        ql_m, ql_v, library = net.l_encoder(gene_vals, batch_labels, batch_mask)
        px_scale, px_r, px_rate, px_dropout = net.decoder(self.dispersion, z, library, batch_labels, batch_mask)
        if self.cell_properties is not None:
            c_loss = self._cell_loss(net.cell_decoder(z), cell_prop_vals, cell_mask, group_idx)
        else:
            c_loss = 0.0
        if self.dispersion == "gene":
            px_r = net.px_r
        elif self.dispersion != "gene-cell":
            raise NotImplementedError(f"dispersion {self.dispersion!r}: dead code in the reference (task.py:475-480)")
        px_r = torch.exp(px_r)
        l_mean, l_var = self.library_stats(gene_vals)
        g_loss, kl_z = self.gene_loss(gene_vals, qz_m, qz_v, ql_m, ql_v, l_mean, l_var, px_rate, px_r, px_dropout)
        loss = self.gene_loss_coeff * g_loss + c_loss
        if self.l1_lambda > 0:
            l1 = sum(torch.norm(p, p=1) for n, p in net.named_parameters() if "z_encoder" in n and "bias" not in n)
            loss = loss + self.l1_lambda * l1
        self.log("gene_loss", g_loss)
        self.log("cell_loss", c_loss)
        self.log("kl_weight", self.kl_weight)
        return loss

    # -- validation / inference (task.py:235-282, 301-362, 364-416) ---------------------------------
    def _create_results_dict(self):
        """Per-epoch buffers the validation step appends to (task.py:103-117)."""
        res: Dict[str, Any] = {"epoch": self.current_epoch}
        for k in (self.cell_properties or {}):
            res[k], res["pred_" + k] = [], []
        for k in (self.batch_properties or {}):
            res[k] = []
        for k in ("latent", "cell_mask", "px_r", "cell_idx", "SubID"):
            res[k] = []
        self.results = res
        # running sums behind the logged accuracy / explained variance of each property
        self._hits = {k: [0, 0] for k, v in (self.cell_properties or {}).items() if v["discrete"]}
        self._moments = {k: np.zeros(6) for k, v in (self.cell_properties or {}).items() if not v["discrete"]}

    def cell_scores(self, cell_pred, cell_targets, cell_mask, latent_mean, subject):
        """Accuracy (discrete) / explained variance (continuous) over the masked cells and the per-cell
        records of the epoch (task.py:364-416; subject_batch_size 1)."""
        if self.subject_batch_size > 1:
            raise NotImplementedError("vae_amd: subject-level batches are not wired into validation")
        for n, (k, prop) in enumerate(self.cell_properties.items()):
            keep = cell_mask[:, n] != 0
            if prop["discrete"]:
                tgt = cell_targets[:, n].to(torch.int64)
                hit = (cell_pred[k].argmax(dim=-1) == tgt) & keep
                self._hits[k][0] += int(hit.sum())
                self._hits[k][1] += int(keep.sum())
                if k != "SubID":
                    self.results[k].append(tgt.cpu().numpy())
                    self.results["pred_" + k].append(torch.softmax(cell_pred[k].float(), dim=-1).detach().cpu().numpy())
                self.log(k, self._hits[k][0] / max(self._hits[k][1], 1))
            else:
                pred = cell_pred[k][:, 0].float()
                p64, t64 = pred[keep].double(), cell_targets[keep, n].double()
                d = t64 - p64
                self._moments[k] += np.array([p64.numel(), float(t64.sum()), float((t64 * t64).sum()), float(d.sum()),
                                              float((d * d).sum()), 0.0])
                cnt, st, stt, sd, sdd, _ = self._moments[k]
                if cnt > 1:   # 1 - Var(target - pred) / Var(target), the ExplainedVariance metric
                    var_t = stt / cnt - (st / cnt) ** 2
                    var_d = sdd / cnt - (sd / cnt) ** 2
                    self.log(k, 1.0 - var_d / var_t if var_t > 0 else float("nan"))
                self.results[k].append(cell_targets[:, n].detach().cpu().numpy())
                self.results["pred_" + k].append(pred.detach().cpu().numpy())
        if latent_mean is not None:
            self.results["latent"].append(latent_mean.detach().float().cpu().numpy())
        self.results["cell_mask"].append(cell_mask.detach().float().cpu().numpy())
        self.results["SubID"].append(subject)

    def _record_validation(self, batch, qz_m, px_rate):
        _, cell_targets, cell_mask, batch_labels, _, cell_idx, _, subject = batch
        self.results["cell_idx"].append(cell_idx.detach().cpu().numpy() if torch.is_tensor(cell_idx) else np.asarray(cell_idx))
        if self.save_gene_vals and px_rate is not None:
            self.results["px_r"].append(px_rate.detach().float().cpu().numpy())
        if self.cell_properties is not None:
            with torch.no_grad():
                self.cell_scores(self.network.cell_decoder(qz_m), cell_targets, cell_mask, qz_m, subject)
        if self.batch_properties is not None:
            for n, k in enumerate(self.batch_properties.keys()):
                self.results[k].append(batch_labels[:, n].detach().cpu().numpy())

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """Module-API validation step (task.py:235-282): forward, validation losses, cell heads on qz_m."""
        gene_vals, _, _, batch_labels, batch_mask = batch[:5]
        net = self.network
        qz_m, qz_v, z = net.z_encoder(gene_vals, batch_labels, batch_mask)
        ql_m, ql_v, library = net.l_encoder(gene_vals, batch_labels, batch_mask)
        _, px_r, px_rate, px_dropout = net.decoder(self.dispersion, z, library, batch_labels, batch_mask)
        px_r = torch.exp(net.px_r if self.dispersion == "gene" else px_r)
        l_mean, l_var = self.library_stats(gene_vals)
        g_loss, kl_z = self.gene_loss(gene_vals, qz_m, qz_v, ql_m, ql_v, l_mean, l_var, px_rate, px_r, px_dropout,
                                      training=False)
        self._record_validation(batch, qz_m, px_rate)
        return g_loss

    @torch.no_grad()
    def validation_step_fused(self, batch, eps_z=None, eps_l=None):
        """The validation step on the fused forward (ElboStep.infer): one native call, px_rate only when
        ``save_gene_vals`` asks for it."""
        if self._arena is None:
            self.attach_arena()
        if self.library_mean is None or self.library_var is None:
            raise ValueError("fused step needs the library prior constants (library_mean / library_var)")
        # validation has its OWN step object (no dropout; own workspace): rebuilding the training one here would make
        # the next training step rebuild it again and reallocate its workspace
        if self._fused_val is None:
            self._fused_val = ElboStep(self._arena, l_mean=float(self.library_mean), l_var=float(self.library_var),
                                       kl_weight=self.kl_weight, gene_loss_coeff=self.gene_loss_coeff,
                                       precision=self.precision)
        gene_vals = batch[0]
        B, L = gene_vals.shape[0], self.network.n_latent
        if eps_z is None:
            eps_z = torch.empty(B, L, device=gene_vals.device).normal_()
            eps_l = torch.empty(B, 1, device=gene_vals.device).normal_()
        self._use_weight_side_products(self._fused_val)
        self._fused_val.kl_weight = self.kl_weight      # KL warm-up: a field of the step's config, not a new step
        cov = None
        if self.batch_properties:
            cov = ElboStep.covariates(batch[3], batch[4], self.network.z_encoder.encoder.batch_dims)
        out = self._fused_val.infer(gene_vals, eps_z, eps_l, want_px_rate=self.save_gene_vals, batch_cov=cov)
        self.log("recon_loss_val", out["recon_mean"])
        self.log("kl_z_val", out["scalars"][1])
        self._record_validation(batch, out["qz_m"], out.get("px_rate"))
        return out

    def on_validation_epoch_end(self):
        """Concatenate the epoch's records (task.py:301-362, without the pickle dump) and start a new epoch."""
        done = {}
        for k, v in self.results.items():
            if isinstance(v, list) and k != "SubID":
                done[k] = np.concatenate(v, axis=0) if len(v) > 0 else v
            else:
                done[k] = v
        self._create_results_dict()
        self.results["epoch"] = self.current_epoch + 1
        return done

    # -- fused fast path ------------------------------------------------------------------------
    def _use_weight_side_products(self, step: ElboStep) -> None:
        src = self.weight_maxima_source
        step.use_weight_maxima(src if getattr(src, "weight_maxima", None) is not None else None)
        step.use_weight_images(src if getattr(src, "weight_images", None) is not None else None)

    def attach_arena(self, device="cuda") -> ParamArena:
        """Re-home every parameter of the network into the flat arena (data and .grad become
        views), so the fused step, the module-API path, the optimiser and the gradient
        all-reduce share one storage."""
        net = self.network
        if net.dispersion != "gene" or net.grouped_encoder:
            raise NotImplementedError("fused step: dispersion 'gene', single encoder")
        sd = {k: v for k, v in net.named_parameters()}
        arena = ParamArena(sd, net.n_layers, self.reconstruction_loss, device)
        net.to(device)
        for k, p in net.named_parameters():
            p.data = arena.p(k)
            p.grad = None if k in arena.no_grad else arena.g(k)
        self._arena = arena
        # a checkpoint loaded into the network writes the arena views behind the optimiser's back
        if getattr(self, "_load_hook", None) is not None:
            self._load_hook.remove()
        self._load_hook = net.register_load_state_dict_post_hook(lambda *_: arena.touch())
        self._fused = self._fused_val = None
        self._l1 = None
        self._native_heads = None
        return arena

    def _hidden_dropout(self) -> float:
        """p of the nn.Dropout layers the fused step applies (z-encoder and decoder FCLayers, modules.py:115)."""
        net = self.network
        ps = {float(m.p) for part in (net.z_encoder, net.decoder) for m in part.modules() if isinstance(m, nn.Dropout)}
        if any(float(m.p) > 0 for m in net.l_encoder.modules() if isinstance(m, nn.Dropout)):
            raise NotImplementedError("fused step: the library encoder is built without dropout (networks.py:215-222)")
        if len(ps) > 1:
            raise NotImplementedError(f"fused step: one dropout rate for the z-encoder and the decoder, got {sorted(ps)}")
        return ps.pop() if ps else 0.0

    def training_step_fused(self, batch, eps_z=None, eps_l=None, loss_scale: float = 1.0, reducer=None):
        """One fused fwd+bwd pass; gradients land in ``arena.flat_grad`` (= every p.grad)."""
        self.train_step += 1
        if self._arena is None:
            self.attach_arena()
        if self.library_mean is None or self.library_var is None:
            raise ValueError("fused step needs the library prior constants (library_mean / library_var)")
        p_drop = self._hidden_dropout() if self.training else 0.0
        gdrop = self.network.z_encoder.drop
        p_in = float(gdrop.p) if self.training and gdrop.p >= 1e-6 else 0.0
        if getattr(self.network.l_encoder.drop, "p", 0.0) >= 1e-6:
            raise NotImplementedError("fused step: the library encoder is built without input dropout (networks.py:215-222)")
        if self._fused is None or self._fused.dropout_rate != p_drop or self._fused.input_dropout_rate != p_in:
            running = None
            if p_in > 0:   # share GeneDropout.running_non_zero_means with the module: one state for both paths
                G = self.network.px_r.shape[0]
                cur = gdrop.running_non_zero_means
                running = torch.zeros(1, G, dtype=torch.float32, device=self._arena.device)
                if torch.is_tensor(cur):
                    running.copy_(cur.to(running.device).reshape(1, G))
                gdrop.running_non_zero_means = running
                running = running.view(G)
            self._fused = ElboStep(self._arena, l_mean=float(self.library_mean), l_var=float(self.library_var),
                                   kl_weight=self.kl_weight, gene_loss_coeff=self.gene_loss_coeff,
                                   precision=self.precision, dropout_rate=p_drop, seed=self.dropout_seed,
                                   input_dropout_rate=p_in, input_dropout_alpha=float(gdrop.alpha), gene_running=running)
            self._fused.rng_step = self.train_step
        self._use_weight_side_products(self._fused)
        gene_vals, cell_prop_vals, cell_mask = batch[0], batch[1], batch[2]
        B, L = gene_vals.shape[0], self.network.n_latent
        if eps_z is None:  # same draw order as the reference: z-encoder noise, then library noise
            eps_z = torch.empty(B, L, device=gene_vals.device).normal_()
            eps_l = torch.empty(B, 1, device=gene_vals.device).normal_()
        fn = heads = None
        if self.cell_properties is not None and cell_prop_vals is not None:
            if getattr(self, "_native_heads", None) is None:   # linear heads (the shipped setting): native kernels
                dec = self.network.cell_decoder
                if dec.grad_reverse_dict is None:   # the module path (and the reference, modules.py:317) raise here too
                    raise AttributeError("CellDecoder: grad_reverse_dict is None (pass a dict, may be empty)")
                hidden = any(not isinstance(m, nn.Linear) for m in dec.cell_mlp.values())
                widths = sum(len(v["values"]) if v["discrete"] else 1 for v in self.cell_properties.values())
                from . import _lib
                if hidden or self.gdro or len(self.cell_properties) > _lib.VAE_MAX_HEADS or widths > 64 or self.network.n_latent > 64:
                    self._native_heads = False      # hidden-layer heads / GroupDRO (per-cell losses go back to the host) /
                                                    # beyond the kernel's limits: torch autograd on (B, L)
                else:                               # any other configuration error surfaces
                    self._native_heads = CellHeads(self._arena, self.cell_properties, dec.grad_reverse_dict,
                                                   class_weights=self.class_weights)
            if self._native_heads:
                heads = (self._native_heads, cell_prop_vals.to(torch.float32), cell_mask.to(torch.float32))
            else:
                group_idx = batch[6] if len(batch) > 6 else None
                fn = lambda z: self._cell_loss(self.network.cell_decoder(z), cell_prop_vals, cell_mask, group_idx)  # noqa: E731
        l1 = None
        if self.l1_lambda > 0:   # task.py:491-496; its gradient must be in place before the buckets are reduced
            from .optim import L1Penalty
            if self._l1 is None:
                self._l1 = L1Penalty(self._arena)
            if reducer is not None and reducer.world > 1:
                raise NotImplementedError("fused step: l1_lambda with a gradient reducer (add the penalty after finish())")
        self._fused.kl_weight = self.kl_weight          # KL warm-up (task.py:213-220): a field of the step's config
        cov = None
        if self.batch_properties:   # modules.py:121-130: the masked one-hot block every FCLayers appends to its input
            cov = ElboStep.covariates(batch[3], batch[4], self.network.z_encoder.encoder.batch_dims)
        out = self._fused.run(gene_vals, eps_z, eps_l, cell_head_fn=fn, loss_scale=loss_scale, reducer=reducer, batch_cov=cov,
                              cell_heads=heads, fractional_counts=getattr(self, "fractional_counts", False))
        if self.l1_lambda > 0:
            l1 = self._l1.apply(self.l1_lambda * loss_scale)
            out["l1"] = l1
            out["loss"] = out["loss"] + self.l1_lambda * l1
        self.log("gene_loss", out["gene_loss"])
        self.log("cell_loss", out["cell_loss"])
        return out

    def configure_optimizers(self):
        """AdamW(beta 0.9/0.998, eps 1e-7), weight decay on z_encoder / cell_decoder only, warm-up
        then inverse-sqrt decay (task.py:532-565)."""
        enc = [p for n, p in self.network.named_parameters() if "z_encoder" in n or "cell_decoder" in n]
        other = [p for n, p in self.network.named_parameters() if not ("z_encoder" in n or "cell_decoder" in n)]
        opt = torch.optim.AdamW([{"params": enc, "weight_decay": self.weight_decay},
                                 {"params": other, "weight_decay": 0.0}], lr=self.learning_rate,
                                weight_decay=self.weight_decay, betas=(0.9, 0.998), eps=1e-7)
        sched = {"scheduler": WarmupConstantAndDecaySchedule(opt, warmup_steps=self.warmup_steps), "interval": "step"}
        return {"optimizer": opt, "lr_scheduler": sched, "interval": "step"}
