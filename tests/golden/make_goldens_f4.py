#!/usr/bin/env python3
"""Golden fixtures for SURVEY §8 row f4's two remaining model variants, produced by running the REFERENCE modules:

  f4_batchcov.npz   VAE(batch_properties=...): masked one-hot batch covariates appended to the input of every FCLayers
                    (both encoders and the decoder; modules.py:121-130), labels with -1 (clamped in place) and masked rows
  f4_grouped.npz    VAE(grouped_encoder=True): one Encoder per cell property, latents concatenated (modules.py:134-188)

Run ONCE in the build container (where /root/reference is mounted read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens_f4.py

The live sequence of task.py:460-489 is executed by hand (task.py needs pytorch_lightning, absent here).  The files hold
DATA only: the reference's initial parameters under a fixed seed, the inputs, every output the step tests pin and every
parameter gradient, in fp64.  Nothing here runs on the GPU box."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True
with contextlib.redirect_stdout(io.StringIO()):
    import networks as ref_networks  # noqa: E402
    import distributions as ref_dist  # noqa: E402
from torch.distributions import Normal, kl_divergence as kl  # noqa: E402

sys.path.insert(0, HERE)
from golden_inputs import F4_CASES, build_f4_inputs  # noqa: E402


def run(case):
    torch.manual_seed(case["seed"])
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_networks.VAE(case["G"], n_hidden=case["H"], n_hidden_decoder=case["Hd"], n_hidden_library=case["Hl"],
                               n_latent=case["L"], n_latent_cell_decoder=case["L_cell"], n_layers=case["n_layers"],
                               dropout_rate=0.0, input_dropout_rate=0.0, cell_properties=case["cell_properties"],
                               batch_properties=case["batch_properties"], grouped_encoder=case["grouped"],
                               grad_reverse_dict=case["grad_reverse"], reconstruction_loss=case["likelihood"],
                               cell_decoder_hidden_layer=False)
    net = net.double().train()
    x, eps_z, eps_l, batch_labels, batch_mask, targets, mask = build_f4_inputs(case)
    blob = {f"param/{k}": v.detach().numpy().copy() for k, v in net.state_dict().items()}
    queue = [e for e in eps_z] + [eps_l]        # one (B, L) draw per z-encoder (grouped: one per property), then the library
    orig = torch.distributions.normal._standard_normal

    def fake(shape, dtype, device):
        e = queue.pop(0)
        assert tuple(e.shape) == tuple(shape), (e.shape, shape)
        return e

    bl = None if batch_labels is None else batch_labels.clone()
    torch.distributions.normal._standard_normal = fake
    try:
        qz_m, qz_v, z = net.z_encoder(x, bl, batch_mask)                      # task.py:460
        ql_m, ql_v, library = net.l_encoder(x, bl, batch_mask)                # task.py:463
    finally:
        torch.distributions.normal._standard_normal = orig
    px_scale, _, px_rate, px_dropout = net.decoder("gene", z, library, bl, batch_mask)   # task.py:465
    pred = net.cell_decoder(z)                                                # task.py:470
    if case["grouped"]:
        # The reference cannot take a training step with GroupedEncoder: gene_loss (task.py:188-193) builds
        # Normal(qz_m (B, 3L), sqrt(qz_v (B, 3))) and the broadcast raises.  What it CAN run -- the forward of the
        # three modules and the heads -- is what this fixture pins.
        try:
            kl(Normal(qz_m, torch.sqrt(qz_v)), Normal(torch.zeros_like(qz_m), torch.ones_like(qz_v)))
            raise SystemExit("the reference's KL accepted the grouped shapes: extend this fixture to the full step")
        except RuntimeError:
            pass
        for k, v in dict(qz_m=qz_m, qz_v=qz_v, z=z, ql_m=ql_m, ql_v=ql_v, library=library, px_rate=px_rate,
                         px_dropout=px_dropout, **{"pred_" + k: v for k, v in pred.items()}).items():
            blob[f"out/{k}"] = v.detach().numpy()
        path = os.path.join(HERE, f"f4_{case['name']}.npz")
        np.savez_compressed(path, **blob)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB (forward only)")
        return
    cell_loss = torch.zeros((), dtype=torch.float64)
    for n, (k, prop) in enumerate(case["cell_properties"].items()):           # task.py:429-442
        if prop["discrete"]:
            per = torch.nn.CrossEntropyLoss(reduction="none", ignore_index=-100)(pred[k], targets[:, n].to(torch.int64))
        else:
            per = torch.nn.MSELoss(reduction="none")(torch.squeeze(pred[k]), targets[:, n].double())
        cell_loss = cell_loss + (1.0 * per * mask[:, n].double()).mean()
    px_r = torch.exp(net.px_r)
    l_mean = torch.tensor(case["l_mean"], dtype=torch.float64)
    l_var = torch.tensor(case["l_var"], dtype=torch.float64)
    kl_z = kl(Normal(qz_m, torch.sqrt(qz_v)), Normal(torch.zeros_like(qz_m), torch.ones_like(qz_v))).sum(dim=1)
    kl_l = kl(Normal(ql_m, torch.sqrt(ql_v)), Normal(l_mean, torch.sqrt(l_var))).sum(dim=1)
    recon = -ref_dist.ZeroInflatedNegativeBinomial(mu=px_rate, theta=px_r, zi_logits=px_dropout).log_prob(x).sum(dim=-1)
    gene_loss = recon.mean() + (kl_l + kl_z).mean()
    loss = case["gene_loss_coeff"] * gene_loss + cell_loss
    loss.backward()
    for k, v in dict(qz_m=qz_m, qz_v=qz_v, z=z, ql_m=ql_m, ql_v=ql_v, library=library, px_rate=px_rate, px_dropout=px_dropout,
                     recon=recon, kl_z=kl_z, kl_l=kl_l, gene_loss=gene_loss, cell_loss=cell_loss, loss=loss).items():
        blob[f"out/{k}"] = v.detach().numpy()
    for k, p in net.named_parameters():
        blob[("gradnone/" if p.grad is None else "grad/") + k] = np.zeros(0) if p.grad is None else p.grad.numpy()
    path = os.path.join(HERE, f"f4_{case['name']}.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    torch.set_num_threads(4)
    for c in F4_CASES:
        run(c)
