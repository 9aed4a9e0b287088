#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE modules.

Run ONCE in the build container (where /root/reference is mounted read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

It imports ``networks``, ``distributions`` from /root/reference/src and executes the
live training sequence of ``task.py:460-489`` by hand (``task.py`` itself needs
pytorch_lightning, which is absent here -- SURVEY.md §8c).  Only the resulting
``.npz`` data files are committed; no reference source is copied and nothing here runs
on the GPU box.  Inputs that would be large (the 20k-gene case) are produced by
``formula_tensor`` below, so the tests can regenerate them bit-exactly without
storing them.
"""
import io
import contextlib
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
from golden_inputs import (  # noqa: E402
    formula_tensor, formula_counts, E2E_CASES, WIDE_CASE, build_case_inputs, LIKELIHOOD_GRID,
)


def run_reference(case, params64, x64, eps_z64, eps_l64, targets, mask, dtype):
    """One fwd+bwd of the reference at ``dtype``; returns (outputs, grads)."""
    G, H, Hd, Hl, L = case["G"], case["H"], case["Hd"], case["Hl"], case["L"]
    cp = case["cell_properties"]
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_networks.VAE(
            G, n_hidden=H, n_hidden_decoder=Hd, n_hidden_library=Hl, n_latent=L,
            n_latent_cell_decoder=L, n_layers=case["n_layers"], dropout_rate=0.0,
            input_dropout_rate=0.0, cell_properties=cp, grad_reverse_dict=case["grad_reverse"],
            reconstruction_loss=case["likelihood"], cell_decoder_hidden_layer=False,
        )
    net = net.to(dtype)
    sd = net.state_dict()
    assert set(sd.keys()) == set(params64.keys()), set(sd.keys()) ^ set(params64.keys())
    net.load_state_dict({k: v.to(dtype) for k, v in params64.items()})
    net.train()
    x = x64.to(dtype)
    B = x.shape[0]

    # Normal.rsample draws empty(shape).normal_(): make the reference draw OUR noise by
    # patching the sampler for the duration of the call (order: eps_z then eps_l).
    queue = [eps_z64.to(dtype), eps_l64.to(dtype)]
    orig = torch.distributions.normal._standard_normal

    def fake(shape, dtype, device):
        e = queue.pop(0)
        assert tuple(e.shape) == tuple(shape), (e.shape, shape)
        return e

    torch.distributions.normal._standard_normal = fake
    try:
        qz_m, qz_v, z = net.z_encoder(x, None, None)          # task.py:460
        ql_m, ql_v, library = net.l_encoder(x, None, None)    # task.py:463
    finally:
        torch.distributions.normal._standard_normal = orig
    px_scale, px_r_none, px_rate, px_dropout = net.decoder("gene", z, library, None, None)  # :465
    assert px_r_none is None

    cell_loss = torch.zeros((), dtype=dtype)
    if case["use_cell_loss"]:
        pred = net.cell_decoder(z)                            # task.py:470
        for n, (k, prop) in enumerate(cp.items()):            # task.py:429-442
            if prop["discrete"]:
                per = torch.nn.CrossEntropyLoss(reduction="none", ignore_index=-100)(
                    pred[k], targets[:, n].to(torch.int64))
            else:
                per = torch.nn.MSELoss(reduction="none")(torch.squeeze(pred[k]), targets[:, n].to(dtype))
            cell_loss = cell_loss + (1.0 * per * mask[:, n].to(dtype)).mean()

    px_r = torch.exp(net.px_r)                                # task.py:482-483
    l_mean = torch.from_numpy(np.array(case["l_mean"], dtype=np.float32)).to(dtype)
    l_var = torch.from_numpy(np.array(case["l_var"], dtype=np.float32)).to(dtype)

    kl_z = kl(Normal(qz_m, torch.sqrt(qz_v)), Normal(torch.zeros_like(qz_m), torch.ones_like(qz_v))).sum(dim=1)
    kl_l = kl(Normal(ql_m, torch.sqrt(ql_v)), Normal(l_mean, torch.sqrt(l_var))).sum(dim=1)
    if case["likelihood"] == "zinb":
        recon = -ref_dist.ZeroInflatedNegativeBinomial(mu=px_rate, theta=px_r, zi_logits=px_dropout).log_prob(x).sum(dim=-1)
    else:
        recon = -ref_dist.NegativeBinomial(mu=px_rate, theta=px_r).log_prob(x).sum(dim=-1)
    gene_loss = recon.mean() + case["kl_weight"] * (kl_l + kl_z).mean()   # task.py:202
    loss = case["gene_loss_coeff"] * gene_loss + cell_loss               # task.py:489
    loss.backward()

    out = dict(qz_m=qz_m, qz_v=qz_v, z=z, ql_m=ql_m, ql_v=ql_v, library=library, px_scale=px_scale,
               px_rate=px_rate, px_dropout=px_dropout, recon=recon, kl_z=kl_z, kl_l=kl_l,
               gene_loss=gene_loss, cell_loss=cell_loss, loss=loss)
    out = {k: v.detach().numpy() for k, v in out.items()}
    grads = {n: (None if p_.grad is None else p_.grad.detach().numpy()) for n, p_ in net.named_parameters()}
    return out, grads


def save_case(case, keep_big: bool):
    params, x, eps_z, eps_l, targets, mask = build_case_inputs(case)
    blob = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        out, grads = run_reference(case, params, x, eps_z, eps_l, targets, mask, dtype)
        for k, v in out.items():
            if not keep_big and v.ndim == 2 and v.shape[1] == case["G"]:
                v = v[:, :: case["sample_stride"]]          # strided sample of (B,G) outputs
            blob[f"{tag}/out/{k}"] = v
        for k, v in grads.items():
            if v is None:
                blob[f"{tag}/gradnone/{k}"] = np.zeros(0)
                continue
            if not keep_big and v.ndim == 2 and case["G"] in v.shape:
                ax = v.shape.index(case["G"])
                v = np.take(v, np.arange(0, case["G"], case["sample_stride"]), axis=ax)
            blob[f"{tag}/grad/{k}"] = v
    path = os.path.join(HERE, f"e2e_{case['name']}.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def save_likelihood_grid():
    """Kernel-level grid for log_nb_positive / log_zinb_positive + autograd gradients."""
    xs, mus, thetas, pis = LIKELIHOOD_GRID
    X, MU, TH, PI = np.meshgrid(xs, mus, thetas, pis, indexing="ij")
    blob = dict(x=X.ravel().astype(np.float64), mu=MU.ravel(), theta=TH.ravel(), pi=PI.ravel())
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        x = torch.tensor(blob["x"], dtype=dtype)
        mu = torch.tensor(blob["mu"], dtype=dtype, requires_grad=True)
        th = torch.tensor(blob["theta"], dtype=dtype, requires_grad=True)
        pi = torch.tensor(blob["pi"], dtype=dtype, requires_grad=True)
        z = ref_dist.log_zinb_positive(x, mu, th, pi)
        z.sum().backward()
        blob[f"{tag}/zinb"] = z.detach().numpy()
        blob[f"{tag}/zinb_dmu"], blob[f"{tag}/zinb_dtheta"], blob[f"{tag}/zinb_dpi"] = (
            mu.grad.numpy().copy(), th.grad.numpy().copy(), pi.grad.numpy().copy())
        mu.grad = None; th.grad = None
        n = ref_dist.log_nb_positive(x, mu, th)
        n.sum().backward()
        blob[f"{tag}/nb"] = n.detach().numpy()
        blob[f"{tag}/nb_dmu"], blob[f"{tag}/nb_dtheta"] = mu.grad.numpy().copy(), th.grad.numpy().copy()
    path = os.path.join(HERE, "likelihood_grid.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def save_init_fixture():
    """Initial weights of the reference VAE under torch.manual_seed(43) (train.py:20) for a small
    configuration: the product's VAE must consume the RNG identically and reproduce them."""
    from golden_inputs import INIT_CASE
    c = INIT_CASE
    torch.manual_seed(43)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref_networks.VAE(c["G"], n_hidden=c["H"], n_hidden_decoder=c["Hd"], n_hidden_library=c["Hl"],
                               n_latent=c["L"], n_latent_cell_decoder=c["L"], n_layers=c["n_layers"], dropout_rate=0.1,
                               input_dropout_rate=0.0, cell_properties=c["cell_properties"],
                               grad_reverse_dict=c["grad_reverse"], cell_decoder_hidden_layer=False)
    blob = {k: v.detach().numpy() for k, v in net.state_dict().items()}
    path = os.path.join(HERE, "init_seed43.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    torch.set_num_threads(4)
    save_init_fixture()
    save_likelihood_grid()
    for c in E2E_CASES:
        save_case(c, keep_big=True)
    save_case(WIDE_CASE, keep_big=False)
