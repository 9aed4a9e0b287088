"""GPU: the reference's module API (VAE / Encoder / DecoderSCVI / distributions / LitVAE) running on
the HIP kernels, against the reference-made goldens.  Two routes are checked:
  * LitVAE.training_step  -- module by module through autograd Functions over the C-ABI ops;
  * LitVAE.training_step_fused -- the native fused step sharing the same parameters (arena views).
"""
import copy
import io
import contextlib

import numpy as np
import pytest
import torch

from golden_inputs import E2E_CASES, build_case_inputs
from helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL_CELL, TOL_GRAD = 1e-5, 1e-4


def build_task(case):
    from vae_amd.networks import VAE
    from vae_amd.task import LitVAE
    case = copy.deepcopy(case)
    params, x, eps_z, eps_l, targets, mask = build_case_inputs(case)
    with contextlib.redirect_stdout(io.StringIO()):
        net = VAE(case["G"], n_hidden=case["H"], n_hidden_decoder=case["Hd"], n_hidden_library=case["Hl"],
                  n_latent=case["L"], n_latent_cell_decoder=case["L"], n_layers=case["n_layers"], dropout_rate=0.0,
                  input_dropout_rate=0.0, cell_properties=case["cell_properties"], grad_reverse_dict=case["grad_reverse"],
                  reconstruction_loss=case["likelihood"], cell_decoder_hidden_layer=False)
    net.load_state_dict({k: v.float() for k, v in params.items()})
    task = LitVAE(net, cell_properties=case["cell_properties"] if case["use_cell_loss"] else None,
                  reconstruction_loss=case["likelihood"], n_epochs_kl_warmup=None, gene_loss_coeff=case["gene_loss_coeff"],
                  library_mean=np.float32(case["l_mean"]), library_var=np.float32(case["l_var"]))
    task.train()
    batch = (x.float().cuda(), targets.cuda(), mask.cuda(), None, None, None, None, None)
    return case, task, batch, eps_z.float().cuda(), eps_l.float().cuda()


def check_grads(net, gold, tag=""):
    n = 0
    for k, p in net.named_parameters():
        if f"f64/gradnone/{k}" in gold.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        assert rel_err(p.grad.cpu().numpy(), gold[f"f64/grad/{k}"]) < TOL_GRAD, f"{tag}{k}"
        n += 1
    assert n >= 15


@pytest.mark.parametrize("case", [E2E_CASES[0], E2E_CASES[2], E2E_CASES[4], E2E_CASES[5]], ids=lambda c: c["name"])
def test_module_api_training_step(case, monkeypatch):
    import vae_amd.modules as M
    gold = load_golden(f"e2e_{case['name']}.npz")
    case, task, batch, eps_z, eps_l = build_task(case)
    task.network.cuda()
    queue = [eps_z, eps_l]
    monkeypatch.setattr(M, "reparameterize_gaussian", lambda mu, var: mu + queue.pop(0) * var.sqrt())
    loss = task.training_step(batch, 0)
    loss.backward()
    g = float(gold["f64/out/loss"])
    assert abs(float(loss.detach()) - g) < TOL_CELL * abs(g)
    check_grads(task.network, gold)


@pytest.mark.parametrize("case", [E2E_CASES[1], E2E_CASES[4]], ids=lambda c: c["name"])
def test_fused_step_through_litvae(case):
    gold = load_golden(f"e2e_{case['name']}.npz")
    case, task, batch, eps_z, eps_l = build_task(case)
    arena = task.attach_arena("cuda")
    out = task.training_step_fused(batch, eps_z, eps_l)
    g = float(gold["f64/out/loss"])
    assert abs(float(out["loss"]) - g) < TOL_CELL * abs(g)
    check_grads(task.network, gold, "fused:")
    # parameters are arena views: an optimiser step on the module moves the arena
    before = arena.flat.clone()
    opt = task.configure_optimizers()["optimizer"]
    opt.step()
    assert not torch.equal(before, arena.flat)


@pytest.mark.parametrize("case", [E2E_CASES[1], E2E_CASES[4]], ids=lambda c: c["name"])
def test_validation_step_module_and_fused_agree(case, monkeypatch):
    """Row f3: LitVAE.validation_step (module API) and validation_step_fused (one native forward) log the
    same validation losses, record the same per-cell results and match the reference's px_rate."""
    import vae_amd.modules as M
    gold = load_golden(f"e2e_{case['name']}.npz")
    case, task, batch, eps_z, eps_l = build_task(case)
    task.save_gene_vals = True
    task.eval()
    B = batch[0].shape[0]
    batch = batch[:5] + (torch.arange(B), None, np.array(["s"] * B))
    task.attach_arena("cuda")
    queue = [eps_z, eps_l]
    monkeypatch.setattr(M, "reparameterize_gaussian", lambda mu, var: mu + queue.pop(0) * var.sqrt())
    task.validation_step(batch, 0)
    logged_mod = dict(task.logged)
    res_mod = task.on_validation_epoch_end()
    task.validation_step_fused(batch, eps_z, eps_l)
    res_fused = task.on_validation_epoch_end()
    for k in ("recon_loss_val", "kl_z_val"):
        assert abs(float(logged_mod[k]) - float(task.logged[k])) < TOL_CELL * abs(float(logged_mod[k])), k
    assert abs(float(task.logged["recon_loss_val"]) - float(gold["f64/out/recon"].mean())) < TOL_CELL * abs(float(gold["f64/out/recon"].mean()))
    for res in (res_mod, res_fused):
        assert res["px_r"].shape == (B, case["G"]) and np.array_equal(res["cell_idx"], np.arange(B))
        assert rel_err(res["px_r"], gold["f64/out/px_rate"]) < 2e-5
    if case["use_cell_loss"]:
        assert rel_err(res_fused["latent"], gold["f64/out/qz_m"]) < TOL_CELL
        for k in case["cell_properties"]:
            assert rel_err(res_fused["pred_" + k], res_mod["pred_" + k]) < 1e-5, k
            assert np.array_equal(res_fused[k], res_mod[k])
    assert task.results["px_r"] == [] and task.results["epoch"] == 1


def test_l1_penalty_fused_matches_module_path(monkeypatch):
    """Row f4, L1 term (task.py:491-496): loss += l1_lambda * sum |p| over z_encoder non-bias parameters."""
    import vae_amd.modules as M
    case0 = E2E_CASES[2]
    lam = 3e-4
    _, task_m, batch, eps_z, eps_l = build_task(case0)
    task_m.l1_lambda = lam
    task_m.network.cuda()
    queue = [eps_z, eps_l]
    monkeypatch.setattr(M, "reparameterize_gaussian", lambda mu, var: mu + queue.pop(0) * var.sqrt())
    loss_m = task_m.training_step(batch, 0)
    loss_m.backward()
    _, task_f, batch, eps_z, eps_l = build_task(case0)
    task_f.l1_lambda = lam
    arena = task_f.attach_arena("cuda")
    out = task_f.training_step_fused(batch, eps_z, eps_l)
    ref_l1 = sum(float(p.detach().abs().double().sum()) for n, p in task_m.network.named_parameters()
                 if "z_encoder" in n and "bias" not in n)
    assert abs(float(out["l1"]) - ref_l1) < 1e-6 * ref_l1
    assert abs(float(out["loss"]) - float(loss_m.detach())) < TOL_CELL * abs(float(loss_m.detach()))
    gm = dict(task_m.network.named_parameters())
    n = 0
    for k, p in task_f.network.named_parameters():
        if gm[k].grad is None:
            continue
        assert rel_err(p.grad.cpu().numpy(), gm[k].grad.cpu().numpy()) < TOL_GRAD, k
        n += 1
    assert n >= 15
    # the penalty really is in the gradient: without it the first-layer weight gradient differs by lam * sign(W)
    w = "z_encoder.encoder.fc_layers.0.weight"
    g_with = arena.g(w).clone()
    task_f.l1_lambda = 0.0
    task_f.training_step_fused(batch, eps_z, eps_l)
    assert torch.allclose(g_with - arena.g(w), lam * torch.sign(arena.p(w)), atol=1e-7)


def test_fused_step_applies_the_modules_dropout():
    """training_step_fused reads p from the FCLayers nn.Dropout modules: a new mask every training step, none in
    eval mode (where the step equals the reference golden again)."""
    from torch import nn
    case0 = E2E_CASES[1]
    gold = load_golden(f"e2e_{case0['name']}.npz")
    _, task, batch, eps_z, eps_l = build_task(case0)
    for part in (task.network.z_encoder, task.network.decoder):
        for m in part.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.3
    task.attach_arena("cuda")
    l1 = float(task.training_step_fused(batch, eps_z, eps_l)["loss"])
    l2 = float(task.training_step_fused(batch, eps_z, eps_l)["loss"])
    g = float(gold["f64/out/loss"])
    assert np.isfinite(l1) and np.isfinite(l2) and l1 != l2 and abs(l1 - g) > 1e-4 * abs(g)
    assert task._fused.dropout_rate == pytest.approx(0.3)
    task.eval()
    l3 = float(task.training_step_fused(batch, eps_z, eps_l)["loss"])
    assert abs(l3 - g) < TOL_CELL * abs(g)
    task.train()
    task.network.z_encoder.drop.p = 0.5          # gene-dropout: shares its running statistic with the module
    l4 = float(task.training_step_fused(batch, eps_z, eps_l)["loss"])
    assert np.isfinite(l4) and task._fused.input_dropout_rate == 0.5
    run = task.network.z_encoder.drop.running_non_zero_means
    frac = (batch[0] > 0).float().mean(dim=0, keepdim=True)
    assert run.shape == frac.shape and torch.allclose(run, 0.05 * frac, atol=1e-7)


@pytest.mark.parametrize("variant", ["gene-cell", "poisson"])
def test_module_path_likelihood_variants(variant, monkeypatch):
    """Row f3 variants on the module API: dispersion "gene-cell" (third decoder head, theta per cell and gene,
    modules.py:366,405) and reconstruction_loss "poisson" (task.py:167-168) vs the fp64 oracle."""
    import vae_amd.modules as M
    from oracle import vae_oracle as O
    from vae_amd.networks import VAE
    from vae_amd.task import LitVAE
    case = copy.deepcopy(E2E_CASES[2])
    params, x, eps_z, eps_l, _, _ = build_case_inputs(case)
    params = dict(params)
    g = torch.Generator().manual_seed(3)
    params["decoder.px_r_decoder.weight"] = torch.randn(case["G"], case["Hd"], generator=g, dtype=torch.float64) * 0.05
    params["decoder.px_r_decoder.bias"] = torch.randn(case["G"], generator=g, dtype=torch.float64) * 0.1
    disp = "gene-cell" if variant == "gene-cell" else "gene"
    lik = "poisson" if variant == "poisson" else case["likelihood"]
    o_out, o_grads = O.training_step_with_grads(params, x, eps_z, eps_l, n_layers=case["n_layers"], likelihood=lik,
                                                l_mean=case["l_mean"], l_var=case["l_var"], kl_weight=case["kl_weight"],
                                                gene_loss_coeff=case["gene_loss_coeff"], dispersion=disp)
    with contextlib.redirect_stdout(io.StringIO()):
        net = VAE(case["G"], n_hidden=case["H"], n_hidden_decoder=case["Hd"], n_hidden_library=case["Hl"],
                  n_latent=case["L"], n_latent_cell_decoder=case["L"], n_layers=case["n_layers"], dropout_rate=0.0,
                  input_dropout_rate=0.0, cell_properties=case["cell_properties"], grad_reverse_dict=case["grad_reverse"],
                  reconstruction_loss=lik, dispersion=disp, cell_decoder_hidden_layer=False)
    net.load_state_dict({k: v.float() for k, v in params.items() if not (k == "px_r" and disp == "gene-cell")})
    task = LitVAE(net.cuda(), cell_properties=None, dispersion=disp, reconstruction_loss=lik, n_epochs_kl_warmup=None,
                  gene_loss_coeff=case["gene_loss_coeff"], library_mean=np.float32(case["l_mean"]),
                  library_var=np.float32(case["l_var"]))
    task.train()
    queue = [eps_z.float().cuda(), eps_l.float().cuda()]
    monkeypatch.setattr(M, "reparameterize_gaussian", lambda mu, var: mu + queue.pop(0) * var.sqrt())
    loss = task.training_step((x.float().cuda(), None, None, None, None, None, None, None), 0)
    loss.backward()
    assert abs(float(loss.detach()) - float(o_out["loss"])) < TOL_CELL * abs(float(o_out["loss"]))
    n = 0
    for k, p in net.named_parameters():
        if o_grads[k] is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert rel_err(p.grad.cpu().numpy(), o_grads[k].numpy()) < TOL_GRAD, k
        n += 1
    assert n >= (17 if variant == "gene-cell" else 12)


def test_module_forward_outputs():
    """DecoderSCVI / Encoder forward tensors (px_scale, px_rate, px_dropout, q_m, q_v) vs goldens."""
    import vae_amd.modules as M
    case = E2E_CASES[2]
    gold = load_golden(f"e2e_{case['name']}.npz")
    case, task, batch, eps_z, eps_l = build_task(case)
    net = task.network.cuda()
    queue = [eps_z, eps_l]
    orig = M.reparameterize_gaussian
    M.reparameterize_gaussian = lambda mu, var: mu + queue.pop(0) * var.sqrt()
    try:
        with torch.no_grad():
            out = net(batch[0])
    finally:
        M.reparameterize_gaussian = orig
    for k in ("qz_m", "qz_v", "z", "ql_m", "ql_v", "library", "px_scale", "px_rate", "px_dropout"):
        assert rel_err(out[k].cpu().numpy(), gold[f"f64/out/{k}"]) < 2e-5, k
    assert out["px_r"] is None
    assert abs(float(out["px_scale"].sum(-1).mean()) - 1.0) < 1e-5


def test_distribution_api():
    from oracle import vae_oracle as O
    from vae_amd.distributions import NegativeBinomial, ZeroInflatedNegativeBinomial, log_zinb_positive
    g = torch.Generator().manual_seed(3)
    x = torch.poisson(torch.rand(6, 40, generator=g) * 4, generator=g)
    mu = (torch.rand(6, 40, generator=g) * 5 + 0.01).requires_grad_(True)
    theta = (torch.rand(40, generator=g) * 3 + 0.1).requires_grad_(True)
    pi = torch.randn(6, 40, generator=g).requires_grad_(True)
    ref = O.log_zinb_positive(x.double(), mu.double(), theta.double(), pi.double())
    ref.sum().backward()
    gm, gt, gp = mu.grad.clone(), theta.grad.clone(), pi.grad.clone()
    xc = x.cuda()
    muc, thc, pic = (t.detach().cuda().requires_grad_(True) for t in (mu, theta, pi))
    d = ZeroInflatedNegativeBinomial(mu=muc, theta=thc, zi_logits=pic)
    lp = d.log_prob(xc)
    lp.sum().backward()
    assert rel_err(lp.detach().cpu().numpy(), ref.detach().numpy()) < 2e-6
    assert rel_err(muc.grad.cpu().numpy(), gm.numpy()) < 1e-5
    assert rel_err(thc.grad.cpu().numpy(), gt.numpy()) < 1e-5 and thc.grad.shape == (40,)
    assert rel_err(pic.grad.cpu().numpy(), gp.numpy()) < 1e-5
    nb = NegativeBinomial(mu=muc.detach(), theta=thc.detach()).log_prob(xc)
    assert rel_err(nb.cpu().numpy(), O.log_nb_positive(x.double(), mu.detach().double(), theta.detach().double()).numpy()) < 2e-6
    with pytest.warns(UserWarning):                       # fractional counts: warn like the reference, still evaluate
        ZeroInflatedNegativeBinomial(mu=muc.detach(), theta=thc.detach(), zi_logits=pic.detach()).log_prob(xc + 0.5)
    with pytest.raises(ValueError):
        log_zinb_positive(xc, muc, thc, pic, eps=1e-6)
    with pytest.raises(ValueError):
        NegativeBinomial(mu=muc.detach(), theta=thc.detach(), total_count=thc.detach())


def test_train_main_reduces_the_loss():
    """vae_amd.train.main (reference signature, train.py:17): a few hundred fused steps with AdamW,
    clip and the LR schedule bring the ELBO down on synthetic counts."""
    from oracle import vae_oracle as O
    from vae_amd import train
    x = O.synthetic_counts(512, 256, seed=3, zero_inflate=0.3).cuda()
    cfg = dict(n_layers=1, n_hidden=64, n_hidden_decoder=64, n_hidden_library=32, n_latent=8, n_latent_cell_decoder=8,
               dropout_rate=0.0, input_dropout_rate=0.0, grad_reverse_dict={})
    with contextlib.redirect_stdout(io.StringIO()):
        task = train.main(list(range(512)), [], "synthetic", None, counts=x, model_cfg=cfg,
                          task_cfg={"warmup_steps": 20.0, "learning_rate": 2e-3}, trainer_cfg={"precision": "32"},
                          batch_size=128, max_steps=1)
        first = float(task.logged["gene_loss"])
        task = train.main(list(range(512)), [], "synthetic", None, counts=x, model_cfg=cfg,
                          task_cfg={"warmup_steps": 20.0, "learning_rate": 2e-3}, trainer_cfg={"precision": "32"},
                          batch_size=128, max_steps=300)
    last = float(task.logged["gene_loss"])
    assert np.isfinite(last) and last < 0.8 * first, (first, last)


def test_train_main_with_the_shipped_regularisers():
    """The reference's shipped model settings (config.py:55-63: 2 layers, dropout 0.5, input dropout 0.5) train on
    the fused step: hidden dropout + GeneDropout + fused AdamW, loss goes down, the running statistic is live."""
    from oracle import vae_oracle as O
    from vae_amd import train
    x = O.synthetic_counts(512, 256, seed=3, zero_inflate=0.3).cuda()
    cfg = dict(n_layers=2, n_hidden=64, n_hidden_decoder=64, n_hidden_library=32, n_latent=8, n_latent_cell_decoder=8,
               dropout_rate=0.5, input_dropout_rate=0.5, grad_reverse_dict={})
    kw = dict(counts=x, model_cfg=cfg, task_cfg={"warmup_steps": 20.0, "learning_rate": 2e-3}, trainer_cfg={"precision": "32"},
              batch_size=128)
    with contextlib.redirect_stdout(io.StringIO()):
        first = float(train.main(list(range(512)), [], "synthetic", None, max_steps=1, **kw).logged["gene_loss"])
        task = train.main(list(range(512)), [], "synthetic", None, max_steps=300, **kw)
    last = float(task.logged["gene_loss"])
    assert np.isfinite(last) and last < 0.85 * first, (first, last)
    run = task.network.z_encoder.drop.running_non_zero_means
    assert torch.is_tensor(run) and run.is_cuda and 0.0 < float(run.mean()) < 1.0
    assert task._fused.dropout_rate == 0.5 and task._fused.input_dropout_rate == 0.5


# ---- row f4: batch covariates (modules.py:121-130) and GroupedEncoder (modules.py:134-188) vs reference-made goldens ----
def _f4_network(case, gold):
    from vae_amd.networks import VAE
    with contextlib.redirect_stdout(io.StringIO()):
        net = VAE(case["G"], n_hidden=case["H"], n_hidden_decoder=case["Hd"], n_hidden_library=case["Hl"], n_latent=case["L"],
                  n_latent_cell_decoder=case["L_cell"], n_layers=case["n_layers"], dropout_rate=0.0, input_dropout_rate=0.0,
                  cell_properties=case["cell_properties"], batch_properties=case["batch_properties"],
                  grouped_encoder=case["grouped"], grad_reverse_dict=case["grad_reverse"],
                  reconstruction_loss=case["likelihood"], cell_decoder_hidden_layer=False)
    params = {k[len("param/"):]: torch.from_numpy(gold[k]).float() for k in gold.files if k.startswith("param/")}
    assert set(params) == set(net.state_dict())                       # the reference's state-dict keys, shapes below
    net.load_state_dict(params)
    return net.cuda()


def test_batch_covariates_training_step_matches_the_reference(monkeypatch):
    """VAE(batch_properties=...) through LitVAE.training_step on the HIP ops: the masked one-hot covariates enter both
    encoders (after the count normalisation, as a small second product next to the gene-wide count GEMM) and the decoder;
    -1 labels are clamped in place and masked, one cell has no covariate at all.  Values and every parameter gradient
    against the fp64 run of the reference (tests/golden/make_goldens_f4.py)."""
    import vae_amd.modules as M
    from golden_inputs import F4_CASES, build_f4_inputs
    from vae_amd.task import LitVAE
    case = copy.deepcopy(F4_CASES[0])
    gold = load_golden("f4_batchcov.npz")
    x, eps_z, eps_l, bl, bm, targets, mask = build_f4_inputs(case)
    net = _f4_network(case, gold)
    G = case["G"]
    assert net.z_encoder.encoder.fc_layers[0].weight.shape == (case["H"], G + 5)      # 3 + 2 covariate columns
    task = LitVAE(net, cell_properties=case["cell_properties"], batch_properties=case["batch_properties"],
                  reconstruction_loss="zinb", n_epochs_kl_warmup=None, gene_loss_coeff=case["gene_loss_coeff"],
                  library_mean=np.float32(case["l_mean"]), library_var=np.float32(case["l_var"])).train()
    labels = bl.clone().cuda()
    batch = (x.float().cuda(), targets.cuda(), mask.cuda(), labels, bm.cuda(), None, None, None)
    queue = [eps_z[0].float().cuda(), eps_l.float().cuda()]
    seen = {}

    def rsample(mu, var):
        out = mu + queue.pop(0) * var.sqrt()
        seen.setdefault("z", out)
        return out

    monkeypatch.setattr(M, "reparameterize_gaussian", rsample)
    loss = task.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
    assert int(labels.min()) == 0                                        # clamped in place, as the reference does
    assert abs(float(loss.detach()) - float(gold["out/loss"])) < TOL_CELL * abs(float(gold["out/loss"]))
    assert rel_err(seen["z"].detach().cpu().numpy(), gold["out/z"]) < TOL_CELL
    assert abs(float(task.logged["gene_loss"]) - float(gold["out/gene_loss"])) < TOL_CELL * abs(float(gold["out/gene_loss"]))
    assert abs(float(task.logged["cell_loss"]) - float(gold["out/cell_loss"])) < TOL_CELL * abs(float(gold["out/cell_loss"]))
    n = 0
    for k, p in net.named_parameters():
        if f"gradnone/{k}" in gold.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        g = gold[f"grad/{k}"]
        assert rel_err(p.grad.cpu().numpy(), g) < TOL_GRAD, k
        if k.endswith("encoder.fc_layers.0.weight"):                     # the covariate columns on their own scale
            assert rel_err(p.grad[:, G:].cpu().numpy(), g[:, G:]) < TOL_GRAD, k + " (covariate columns)"
        n += 1
    assert n >= 30


@pytest.mark.parametrize("precision", ["fp32x3", "f16x2"])
def test_batch_covariates_fused_step_matches_the_reference(precision):
    """The same reference-made fixture through LitVAE.training_step_fused: the arena re-homes the (H, G + 5) first-layer
    weights with padded rows, the native step adds the covariate products (include/vae_amd.h, vae_step_config.n_cov)."""
    from golden_inputs import F4_CASES, build_f4_inputs
    from vae_amd import ops
    from vae_amd.task import LitVAE
    case = copy.deepcopy(F4_CASES[0])
    gold = load_golden("f4_batchcov.npz")
    x, eps_z, eps_l, bl, bm, targets, mask = build_f4_inputs(case)
    net = _f4_network(case, gold)
    G = case["G"]
    task = LitVAE(net, cell_properties=case["cell_properties"], batch_properties=case["batch_properties"],
                  reconstruction_loss="zinb", n_epochs_kl_warmup=None, gene_loss_coeff=case["gene_loss_coeff"],
                  library_mean=np.float32(case["l_mean"]), library_var=np.float32(case["l_var"]),
                  precision=ops.FP32 if precision == "fp32x3" else ops.F16X2).train()
    labels = bl.clone().cuda()
    batch = (x.float().cuda(), targets.cuda(), mask.cuda(), labels, bm.cuda(), None, None, None)
    out = task.training_step_fused(batch, eps_z=eps_z[0].float().cuda(), eps_l=eps_l.float().cuda())
    torch.cuda.synchronize()
    assert int(labels.min()) == 0
    assert net.z_encoder.encoder.fc_layers[0].weight.shape == (case["H"], G + 5)
    assert net.z_encoder.encoder.fc_layers[0].weight.stride() == (72, 1)            # 64 + 5 columns, rows padded to 72
    for k in ("qz_m", "qz_v", "z", "ql_m", "ql_v", "library", "recon", "kl_z", "kl_l"):
        assert rel_err(out[k].cpu().numpy().reshape(gold[f"out/{k}"].shape), gold[f"out/{k}"]) < TOL_CELL, k
    for k in ("loss", "gene_loss", "cell_loss"):
        assert abs(float(out[k]) - float(gold[f"out/{k}"])) < TOL_CELL * abs(float(gold[f"out/{k}"])), k
    n = 0
    for k, p in net.named_parameters():
        if f"gradnone/{k}" in gold.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        g = gold[f"grad/{k}"]
        assert rel_err(p.grad.cpu().numpy(), g) < TOL_GRAD, k
        if k.endswith("encoder.fc_layers.0.weight"):
            assert rel_err(p.grad[:, G:].cpu().numpy(), g[:, G:]) < TOL_GRAD, k + " (covariate columns)"
        if k == "decoder.px_decoder.fc_layers.0.weight":
            assert rel_err(p.grad[:, case["L"]:].cpu().numpy(), g[:, case["L"]:]) < TOL_GRAD, k + " (covariate columns)"
        n += 1
    assert n >= 30
    # the state dict keeps the reference's shapes (strided views of the arena are saved as plain tensors)
    sd = net.state_dict()
    assert sd["z_encoder.encoder.fc_layers.0.weight"].shape == (case["H"], G + 5)


def test_grouped_encoder_forward_matches_the_reference(monkeypatch):
    """VAE(grouped_encoder=True): one Encoder per cell property, latents concatenated (modules.py:134-188).  The
    reference itself cannot take a training step with it (its gene_loss, task.py:188-193, pairs a (B, 3L) mean with a
    (B, 3) variance and the broadcast raises), so what it can run is pinned: the forward of the encoders, the decoder on
    the concatenated latent and the heads."""
    import vae_amd.modules as M
    from golden_inputs import F4_CASES, build_f4_inputs
    case = copy.deepcopy(F4_CASES[1])
    gold = load_golden("f4_grouped.npz")
    x, eps_z, eps_l, _, _, _, _ = build_f4_inputs(case)
    net = _f4_network(case, gold).train()
    queue = [e.float().cuda() for e in eps_z] + [eps_l.float().cuda()]
    monkeypatch.setattr(M, "reparameterize_gaussian", lambda mu, var: mu + queue.pop(0) * var.sqrt())
    xd = x.float().cuda()
    qz_m, qz_v, z = net.z_encoder(xd, None, None)
    ql_m, ql_v, library = net.l_encoder(xd, None, None)
    _, _, px_rate, px_dropout = net.decoder("gene", z, library, None, None)
    pred = net.cell_decoder(z)
    torch.cuda.synchronize()
    L, P = case["L"], len(case["cell_properties"])
    assert qz_m.shape == (case["B"], L * P) and qz_v.shape == (case["B"], P) and z.shape == (case["B"], L * P)
    got = dict(qz_m=qz_m, qz_v=qz_v, z=z, ql_m=ql_m, ql_v=ql_v, library=library, px_dropout=px_dropout,
               **{"pred_" + k: v for k, v in pred.items()})
    for k, v in got.items():
        g = gold[f"out/{k}"]
        assert rel_err(v.detach().cpu().numpy().reshape(g.shape), g) < TOL_CELL, k
    g = gold["out/px_rate"]
    assert np.max(np.abs(px_rate.detach().cpu().numpy() - g) / np.abs(g).max(axis=1, keepdims=True)) < TOL_CELL
    # gradients flow through every per-property encoder (autograd over the HIP ops)
    (px_rate.sum() * 1e-3 + sum(v.sum() for v in pred.values())).backward()
    for name in case["cell_properties"]:
        w = net.z_encoder.z_enc[name].encoder.fc_layers[0].weight
        assert w.grad is not None and float(w.grad.abs().max()) > 0


def test_group_dro_on_the_fused_step_matches_the_module_path(monkeypatch):
    """`use_gdro=True` on the fused step: the heads run through torch autograd on (B, L) with the donor groups of the batch
    (per-cell losses go back to the host, as in the reference, task.py:431-448); loss, tracker state and every gradient equal
    the module path's.  (The tracker itself is pinned against the reference in test_model_cpu.py.)"""
    import vae_amd.modules as M
    from vae_amd.losses import GroupDRO
    case0 = E2E_CASES[4]                     # a heads case
    names = list(case0["cell_properties"])
    groups = GroupDRO.DEFAULT_GROUPS[:14]

    def arm(task, B):
        task.use_gdro = True
        task.gdro = {names[0]: GroupDRO(balances=[list(groups)]), names[1]: GroupDRO()}
        rng = np.random.default_rng(5)
        gi = [groups[i] for i in rng.integers(0, 14, size=4 * B)]
        for g in task.gdro.values():         # running losses from two synthetic batches: unequal weights from here on
            for _ in range(2):
                g.update_batch(torch.from_numpy(rng.random(4 * B) * 3), torch.ones(4 * B), gi)
                g.update_weights()
        task._native_heads = None
        task.train_step = 600
        return [groups[i] for i in np.random.default_rng(6).integers(0, 14, size=B)]

    _, task_m, batch, eps_z, eps_l = build_task(case0)
    gi = arm(task_m, batch[0].shape[0])
    batch = batch[:6] + (gi, None)
    task_m.network.cuda()
    queue = [eps_z, eps_l]
    monkeypatch.setattr(M, "reparameterize_gaussian", lambda mu, var: mu + queue.pop(0) * var.sqrt())
    loss_m = task_m.training_step(batch, 0)
    loss_m.backward()
    _, task_f, batch_f, eps_z, eps_l = build_task(case0)
    assert arm(task_f, batch_f[0].shape[0]) == gi
    task_f.attach_arena("cuda")
    out = task_f.training_step_fused(batch_f[:6] + (gi, None), eps_z, eps_l)
    assert task_f._native_heads is False
    w = task_f.gdro[names[0]].get_weights(gi)
    assert float(w.max() - w.min()) > 0.05                           # the weights are not trivial
    assert abs(float(out["loss"]) - float(loss_m.detach())) < TOL_CELL * abs(float(loss_m.detach()))
    for k in task_f.gdro:
        assert task_f.gdro[k].running_loss == pytest.approx(task_m.gdro[k].running_loss, rel=1e-5)
    gm = dict(task_m.network.named_parameters())
    n = 0
    for k, p in task_f.network.named_parameters():
        if gm[k].grad is None:
            continue
        assert rel_err(p.grad.cpu().numpy(), gm[k].grad.cpu().numpy()) < TOL_GRAD, k
        n += 1
    assert n >= 15
