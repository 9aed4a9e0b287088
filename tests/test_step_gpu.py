"""GPU parity of the fused fwd+bwd ELBO step (vae_amd.step.ElboStep, all HIP kernels through
the C ABI) against the golden vectors the REFERENCE produced (fp64 run = ground truth) and
against the oracle at a size where no golden exists.

Tolerances (BASELINE.json north_star, fp32 mode): per-cell ELBO terms 1e-5 relative,
parameter gradients 1e-4 relative (to the largest entry of each gradient tensor)."""
import copy

import os

import numpy as np
import pytest
import torch

from golden_inputs import E2E_CASES, WIDE_CASE, build_case_inputs
from helpers import load_golden, rel_err, sample_like_golden

pytestmark = pytest.mark.gpu

ALL_CASES = E2E_CASES + [WIDE_CASE]
TOL_CELL, TOL_GRAD = 1e-5, 1e-4


def _cell_head_fn(arena, case, targets, mask):
    """Cell heads through the PRODUCT modules (vae_amd.modules.CellDecoder + vae_amd.task.cell_loss)."""
    from vae_amd.modules import CellDecoder
    from vae_amd.task import cell_loss
    dec = CellDecoder(case["L"], case["cell_properties"], grad_reverse_dict=case["grad_reverse"], use_hidden_layer=False)
    dec = dec.cuda()
    for k in case["cell_properties"]:
        for leaf in ("weight", "bias"):
            p = getattr(dec.cell_mlp[k], leaf)
            p.data = arena.p(f"cell_decoder.cell_mlp.{k}.{leaf}")
            p.grad = arena.g(f"cell_decoder.cell_mlp.{k}.{leaf}")
    t, m = targets.cuda(), mask.cuda()
    return lambda z: cell_loss(dec(z), case["cell_properties"], t, m)


def run_step(case, precision=0, backward=True, native_heads=False):
    from vae_amd.arena import ParamArena
    from vae_amd.step import CellHeads, ElboStep
    case = copy.deepcopy(case)
    params, x, eps_z, eps_l, targets, mask = build_case_inputs(case)
    arena = ParamArena({k: v.float() for k, v in params.items()}, case["n_layers"], case["likelihood"], "cuda")
    step = ElboStep(arena, l_mean=float(np.float32(case["l_mean"])), l_var=float(np.float32(case["l_var"])),
                    kl_weight=case["kl_weight"], gene_loss_coeff=case["gene_loss_coeff"], precision=precision)
    fn = heads = None
    if case["use_cell_loss"] and native_heads:   # vae_cell_heads_step instead of torch autograd on the latent
        heads = (CellHeads(arena, case["cell_properties"], case["grad_reverse"]), targets.float().cuda(), mask.float().cuda())
    elif case["use_cell_loss"]:
        fn = _cell_head_fn(arena, case, targets, mask)
    out = step.run(x.float().cuda(), eps_z.float().cuda(), eps_l.float().cuda(), cell_head_fn=fn, cell_heads=heads, backward=backward)
    torch.cuda.synchronize()
    return case, arena, out


FP32_MODES = [0, 3]   # ops.FP32 (three bf16 terms, six products) and ops.F16X2 (two scaled f16 terms, three products)


@pytest.mark.parametrize("native_heads", [False, True], ids=["torch_heads", "native_heads"])
@pytest.mark.parametrize("precision", FP32_MODES, ids=["bf16x3", "f16x2"])
@pytest.mark.parametrize("case", ALL_CASES, ids=[c["name"] for c in ALL_CASES])
def test_fused_step_matches_reference_goldens(case, precision, native_heads):
    if native_heads and not case["use_cell_loss"]:
        pytest.skip("no cell heads in this case")
    gold = load_golden(f"e2e_{case['name']}.npz")
    case, arena, out = run_step(case, precision=precision, native_heads=native_heads)
    G, stride = case["G"], case["sample_stride"]
    for k in ("qz_m", "qz_v", "z", "ql_m", "ql_v", "library", "recon", "kl_z", "kl_l"):
        g = gold[f"f64/out/{k}"]
        got = out[k].cpu().numpy().reshape(g.shape)
        assert rel_err(got, g) < TOL_CELL, k
    # per-cell relative check (not just relative to the largest cell)
    rec = out["recon"].cpu().numpy().astype(np.float64)
    assert np.max(np.abs(rec - gold["f64/out/recon"]) / np.abs(gold["f64/out/recon"])) < TOL_CELL
    for k in ("gene_loss", "cell_loss", "loss"):
        g = float(gold[f"f64/out/{k}"])
        assert abs(float(out[k]) - g) <= TOL_CELL * max(abs(g), 1e-3), k
    grads = arena.grads()
    checked = 0
    for k, v in grads.items():
        if f"f64/gradnone/{k}" in gold.files:
            if v is not None and case["use_cell_loss"] is False and k.startswith("cell_decoder"):
                assert float(v.abs().max()) == 0.0  # head unused -> reference grad None, ours stays zero
            else:
                assert v is None, f"{k}: reference has no gradient here"
            continue
        g = gold[f"f64/grad/{k}"]
        got = sample_like_golden(v, G, stride)
        assert rel_err(got, g) < TOL_GRAD, k
        checked += 1
    assert checked >= 15


def test_cell_head_gradients_do_not_accumulate_across_steps():
    """The cell heads' gradients come from torch autograd (which accumulates): the step clears them, so a second
    identical step leaves the same gradients as the first."""
    case, arena, out = run_step(E2E_CASES[3])
    first = {k: v.clone() for k, v in arena.grads().items() if v is not None and k.startswith("cell_decoder")}
    assert first and all(float(v.abs().max()) > 0 for v in first.values())
    from vae_amd.step import ElboStep
    params, x, eps_z, eps_l, targets, mask = build_case_inputs(copy.deepcopy(E2E_CASES[3]))
    step = ElboStep(arena, l_mean=float(np.float32(case["l_mean"])), l_var=float(np.float32(case["l_var"])),
                    kl_weight=case["kl_weight"], gene_loss_coeff=case["gene_loss_coeff"])
    fn = _cell_head_fn(arena, case, targets, mask)
    for _ in range(2):
        step.run(x.float().cuda(), eps_z.float().cuda(), eps_l.float().cuda(), cell_head_fn=fn)
    for k, v in first.items():
        assert rel_err(arena.g(k).cpu().numpy(), v.cpu().numpy()) < 1e-6, k


def test_forward_only_leaves_grads_untouched():
    case, arena, out = run_step(E2E_CASES[1], backward=False)
    assert float(arena.flat_grad.abs().max()) == 0.0
    assert torch.isfinite(out["loss"])


@pytest.mark.parametrize("case", ALL_CASES, ids=[c["name"] for c in ALL_CASES])
def test_inference_step_matches_reference_goldens(case):
    """Row f3: validation_step = the forward alone + px_rate export (task.py:235-282) vs the reference's
    forward tensors; no gradient is touched."""
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    gold = load_golden(f"e2e_{case['name']}.npz")
    case = copy.deepcopy(case)
    params, x, eps_z, eps_l, _, _ = build_case_inputs(case)
    arena = ParamArena({k: v.float() for k, v in params.items()}, case["n_layers"], case["likelihood"], "cuda")
    step = ElboStep(arena, l_mean=float(np.float32(case["l_mean"])), l_var=float(np.float32(case["l_var"])),
                    kl_weight=case["kl_weight"], gene_loss_coeff=case["gene_loss_coeff"])
    with pytest.raises(RuntimeError):
        step.px_rate()                                   # nothing to export before a forward
    out = step.infer(x.float().cuda(), eps_z.float().cuda(), eps_l.float().cuda(), want_px_rate=True)
    torch.cuda.synchronize()
    assert float(arena.flat_grad.abs().max()) == 0.0
    G, stride = case["G"], case["sample_stride"]
    g = gold["f64/out/px_rate"]
    got = sample_like_golden(out["px_rate"], G, stride)
    assert got.shape == g.shape
    assert np.max(np.abs(got - g) / np.abs(g).max(axis=1, keepdims=True)) < TOL_CELL   # per cell, to its largest rate
    assert rel_err(out["px_rate"].sum(-1).cpu().numpy(), np.exp(gold["f64/out/library"]).reshape(-1)) < TOL_CELL
    for k in ("qz_m", "recon", "kl_z", "kl_l"):
        gk = gold[f"f64/out/{k}"]
        assert rel_err(out[k].cpu().numpy().reshape(gk.shape), gk) < TOL_CELL, k
    # a training step afterwards invalidates the export (its backward overwrites the logits)
    step.run(x.float().cuda(), eps_z.float().cuda(), eps_l.float().cuda())
    with pytest.raises(RuntimeError):
        step.px_rate()


@pytest.mark.parametrize("case", [E2E_CASES[1], E2E_CASES[2], E2E_CASES[3]], ids=lambda c: c["name"])
def test_hidden_dropout_matches_oracle_with_the_same_masks(case):
    """Row f4: nn.Dropout after the z-encoder / decoder LayerNorms (modules.py:115) on the device's Philox
    stream.  torch's draws cannot be reproduced, so parity is: the HOST regenerates the documented masks and
    the fp64 oracle runs with them.  1-layer = in-kernel masks of the fused middle, 2-layer = dropout kernels
    between the unfused launches (both against the same oracle)."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    from helpers import dropout_multiplier
    case = copy.deepcopy(case)
    params, x, eps_z, eps_l, targets, mask = build_case_inputs(case)
    p_drop, seed, rng_step = 0.5, 987654321012345, 17
    B, H, Hd, nl = case["B"], case["H"], case["Hd"], case["n_layers"]
    masks = {"z_encoder": [torch.from_numpy(dropout_multiplier(p_drop, seed, rng_step, 4 * n, B, H)).double() for n in range(nl)],
             "decoder": [torch.from_numpy(dropout_multiplier(p_drop, seed, rng_step, 4 * n + 2, B, Hd)).double() for n in range(nl)]}
    kw = dict(n_layers=nl, likelihood=case["likelihood"], l_mean=case["l_mean"], l_var=case["l_var"],
              kl_weight=case["kl_weight"], gene_loss_coeff=case["gene_loss_coeff"])
    if case["use_cell_loss"]:
        kw.update(cell_properties=case["cell_properties"], grad_reverse=case["grad_reverse"], cell_targets=targets.double(),
                  cell_mask=mask.double())
    o_out, o_grads = O.training_step_with_grads(params, x, eps_z, eps_l, drop_masks=masks, **kw)
    o_plain, _ = O.training_step_with_grads(params, x, eps_z, eps_l, **kw)
    assert abs(float(o_out["loss"]) - float(o_plain["loss"])) > 1e-3 * abs(float(o_plain["loss"]))   # the masks matter

    def run(train=True):
        arena = ParamArena({k: v.float() for k, v in params.items()}, nl, case["likelihood"], "cuda")
        step = ElboStep(arena, l_mean=float(np.float32(case["l_mean"])), l_var=float(np.float32(case["l_var"])),
                        kl_weight=case["kl_weight"], gene_loss_coeff=case["gene_loss_coeff"], dropout_rate=p_drop, seed=seed)
        fn = _cell_head_fn(arena, case, targets, mask) if case["use_cell_loss"] else None
        out = step.run(x.float().cuda(), eps_z.float().cuda(), eps_l.float().cuda(), cell_head_fn=fn, rng_step=rng_step,
                       train=train)
        torch.cuda.synchronize()
        return arena, {k: v.clone() for k, v in out.items()}

    arena, out = run()
    for k in ("qz_m", "z", "library", "recon", "kl_z", "kl_l"):
        g = o_out[k].numpy()
        assert rel_err(out[k].cpu().numpy().reshape(g.shape), g) < TOL_CELL, k
    assert abs(float(out["loss"]) - float(o_out["loss"])) < TOL_CELL * abs(float(o_out["loss"]))
    n = 0
    for k, v in arena.grads().items():
        if v is None or o_grads[k] is None:
            continue
        assert rel_err(v.cpu().numpy(), o_grads[k].numpy()) < TOL_GRAD, k
        n += 1
    assert n >= 15
    # evaluation mode: no masks, the plain forward
    _, ev = run(train=False)
    assert abs(float(ev["loss"]) - float(o_plain["loss"])) < TOL_CELL * abs(float(o_plain["loss"]))


@pytest.mark.parametrize("n_layers,counts", [(1, "f32"), (2, "f32"), (1, "u8")])
def test_gene_dropout_matches_oracle_with_the_same_mask(n_layers, counts):
    """Row f4: GeneDropout on the z-encoder input (modules.py:190-209) plus hidden dropout = the shipped
    configuration's regularisers.  Two training steps: the running non-zero statistic, the keep probabilities
    (against the reference's formula in fp64) and, with the host-regenerated masks, the whole step vs the oracle."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    from helpers import dropout_multiplier, gene_dropout_keep
    B, G, H, Hd, Hl, L = 48, 400, 64, 48, 32, 6
    p_in, p_hid, alpha, seed = 0.5, 0.25, 0.95, 20240607
    p = O.init_params(G, H, Hd, Hl, L, n_layers, seed=9)
    arena = ParamArena(p, n_layers, "zinb", "cuda")
    gen = torch.Generator().manual_seed(10)
    step = ElboStep(arena, l_mean=5.0, l_var=1.5, dropout_rate=p_hid, seed=seed, input_dropout_rate=p_in,
                    input_dropout_alpha=alpha)
    running = np.zeros(G)
    for it in range(2):
        x = O.synthetic_counts(B, G, seed=11 + it, zero_inflate=0.4).clamp(max=255)
        eps_z, eps_l = torch.randn(B, L, generator=gen), torch.randn(B, 1, generator=gen)
        xd = x.to(torch.uint8).cuda() if counts == "u8" else x.cuda()
        out = step.run(xd, eps_z.cuda(), eps_l.cuda(), rng_step=100 + it)
        torch.cuda.synchronize()
        # the reference's statistic and keep probabilities (modules.py:202-205), fp64
        running = alpha * running + (1 - alpha) * (x.numpy() > 0).mean(axis=0)
        pg = np.clip(p_in * running / running.mean(), 0.0, 0.9)
        assert rel_err(step.gene_running.cpu().numpy(), running) < 1e-6
        thr = out["gene_threshold"].cpu().numpy().view(np.uint32)
        assert np.max(np.abs(thr.astype(np.float64) - pg * 2.0 ** 32)) < 2.0 ** 32 * 2e-6
        assert abs(float(out["gene_p_mean"]) - pg.mean()) < 1e-6
        keep = gene_dropout_keep(thr, seed, 100 + it, B)
        assert abs(keep.mean() - pg.mean()) < 0.02
        x_z = x.double() * torch.from_numpy(keep).double() / pg.mean()          # modules.py:209
        masks = {"z_encoder": [torch.from_numpy(dropout_multiplier(p_hid, seed, 100 + it, 4 * n, B, H)).double() for n in range(n_layers)],
                 "decoder": [torch.from_numpy(dropout_multiplier(p_hid, seed, 100 + it, 4 * n + 2, B, Hd)).double() for n in range(n_layers)]}
        o_out, o_grads = O.training_step_with_grads({k: v.double() for k, v in p.items()}, x.double(), eps_z.double(),
                                                    eps_l.double(), n_layers=n_layers, likelihood="zinb", l_mean=5.0, l_var=1.5,
                                                    drop_masks=masks, x_z=x_z)
        for k in ("qz_m", "z", "library", "recon", "kl_z", "kl_l"):
            g = o_out[k].numpy()
            assert rel_err(out[k].cpu().numpy().reshape(g.shape), g) < TOL_CELL, (it, k)
        assert abs(float(out["loss"]) - float(o_out["loss"])) < TOL_CELL * abs(float(o_out["loss"]))
        n = 0
        for k, v in arena.grads().items():
            if v is None or o_grads[k] is None:
                continue
            assert rel_err(v.cpu().numpy(), o_grads[k].numpy()) < TOL_GRAD, (it, k)
            n += 1
        assert n >= 15
    # evaluation: no masks, no state update
    before = step.gene_running.clone()
    ev = step.run(xd, eps_z.cuda(), eps_l.cuda(), backward=False)
    o_plain = O.training_step({k: v.double() for k, v in p.items()}, x.double(), eps_z.double(), eps_l.double(),
                              n_layers=n_layers, likelihood="zinb", l_mean=5.0, l_var=1.5)
    assert abs(float(ev["loss"]) - float(o_plain["loss"])) < TOL_CELL * abs(float(o_plain["loss"]))
    assert torch.equal(before, step.gene_running) and "gene_threshold" not in ev


def test_bf16_mode_against_bf16_oracle():
    """bf16 GEMM operands / fp32 accumulate: same rounding points as the oracle's bf16 mode;
    stated tolerance 2e-3 on per-cell terms, 2e-2 on gradients (bf16 has 8 mantissa bits and the
    two sides accumulate in different orders, so roundings of intermediate activations differ)."""
    from oracle import vae_oracle as O
    case = copy.deepcopy(E2E_CASES[2])
    params, x, eps_z, eps_l, targets, mask = build_case_inputs(case)
    o_out, o_grads = O.training_step_with_grads(
        {k: v.float() for k, v in params.items()}, x.float(), eps_z.float(), eps_l.float(), n_layers=case["n_layers"],
        likelihood=case["likelihood"], l_mean=case["l_mean"], l_var=case["l_var"], kl_weight=case["kl_weight"],
        gene_loss_coeff=case["gene_loss_coeff"], gemm_dtype=torch.bfloat16)
    _, arena, out = run_step(E2E_CASES[2], precision=1)
    assert rel_err(out["recon"].cpu().numpy(), o_out["recon"].numpy()) < 2e-3
    assert rel_err(out["kl_z"].cpu().numpy(), o_out["kl_z"].numpy()) < 2e-2
    for k, v in arena.grads().items():
        if v is None or o_grads[k] is None:
            continue
        assert rel_err(v.cpu().numpy(), o_grads[k].numpy()) < 3e-2, k


@pytest.mark.parametrize("precision", FP32_MODES, ids=["bf16x3", "f16x2"])
@pytest.mark.parametrize("likelihood", ["zinb", "nb"])
def test_midsize_against_oracle(likelihood, precision):
    """B=96, G=2000 (cfg-1 shape): HIP step vs the fp64 oracle on seeded synthetic data."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    B, G, H, Hd, Hl, L = 96, 2000, 128, 128, 128, 10
    p = O.init_params(G, H, Hd, Hl, L, 1, seed=5)
    x = O.synthetic_counts(B, G, seed=5, zero_inflate=0.3 if likelihood == "zinb" else 0.0)
    g = torch.Generator().manual_seed(6)
    eps_z, eps_l = torch.randn(B, L, generator=g), torch.randn(B, 1, generator=g)
    lm, lv = O.library_prior(x)
    o_out, o_grads = O.training_step_with_grads({k: v.double() for k, v in p.items()}, x.double(), eps_z.double(),
                                                eps_l.double(), n_layers=1, likelihood=likelihood, l_mean=lm, l_var=lv)
    arena = ParamArena(p, 1, likelihood, "cuda")
    step = ElboStep(arena, l_mean=lm, l_var=lv, precision=precision)
    out = step.run(x.cuda(), eps_z.cuda(), eps_l.cuda())
    rec = out["recon"].cpu().double().numpy()
    assert np.max(np.abs(rec - o_out["recon"].numpy()) / np.abs(o_out["recon"].numpy())) < TOL_CELL
    assert abs(float(out["loss"]) - float(o_out["loss"])) < TOL_CELL * abs(float(o_out["loss"]))
    for k, v in arena.grads().items():
        if o_grads[k] is None:
            continue
        assert rel_err(v.cpu().numpy(), o_grads[k].numpy()) < TOL_GRAD, k


@pytest.mark.parametrize("counts", ["integer", "fractional", "large", "tiny", "one_bad_cell", "ragged_genes"])
def test_optimistic_count_product_and_its_repeat(counts):
    """A two-f16-term step on the fused middle: no launch reads the counts ahead of the first encoder product
    (step.hip, xoptimistic).  The product takes them as exact in one unscaled f16 term -- true of integer counts below 2048
    -- and gathers the cells' sums, maxima and exactness flags on the way; the gated launch behind it finishes the means and
    sum_g W and REPEATS the product in the scaled two-term form when a flag came up: fractional counts (mixup,
    data.py:457-477), counts beyond f16's exact integers / range, a batch of tiny values, or a single offending cell in a
    row tile of its own.  Every case against the fp64 oracle at the step tests' tolerances."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    B, G, H, Hd, Hl, L = 160, 4096, 128, 128, 128, 10
    if counts == "ragged_genes":
        G = 4100      # G % 32 = 4: the last slab ends in a partial K step, whose counts enter the statistics too
    p = O.init_params(G, H, Hd, Hl, L, 1, seed=25)
    x = O.synthetic_counts(B, G, seed=25, zero_inflate=0.3)
    if counts == "fractional":
        x = 0.75 * x + 0.25 * x.roll(1, 0)
    elif counts == "large":
        x[3, 17], x[100, 4000], x[159, 5] = 70000.0, 3001.0, 2049.0
    elif counts == "tiny":
        x = x * 2.0 ** -12
    elif counts == "one_bad_cell":
        x[150, 1234] = 0.5     # the second row tile (cells 128..159), one slab
    g = torch.Generator().manual_seed(26)
    eps_z, eps_l = torch.randn(B, L, generator=g), torch.randn(B, 1, generator=g)
    lm, lv = O.library_prior(x)
    o_out, o_grads = O.training_step_with_grads({k: v.double() for k, v in p.items()}, x.double(), eps_z.double(),
                                                eps_l.double(), n_layers=1, likelihood="zinb", l_mean=lm, l_var=lv)
    arena = ParamArena(p, 1, "zinb", "cuda")
    step = ElboStep(arena, l_mean=lm, l_var=lv, precision=3)
    assert step.counts_path(B) == 2 or os.environ.get("VAE_NO_XOPT") or os.environ.get("VAE_NO_XDIRECT")
    xd, ezd, eld = x.cuda(), eps_z.cuda(), eps_l.cuda()
    for rep in range(2):   # the second call finds the first one's statistics in the workspace
        out = step.run(xd, ezd, eld)   # (the step overwrites or clears every gradient itself)
        rec = out["recon"].cpu().double().numpy()
        assert np.max(np.abs(rec - o_out["recon"].numpy()) / np.abs(o_out["recon"].numpy())) < TOL_CELL
        for k in ("qz_m", "library", "kl_z", "kl_l"):
            ref = o_out[k].double().numpy()
            assert rel_err(out[k].cpu().numpy().reshape(ref.shape), ref) < TOL_CELL, k
        for k, v in arena.grads().items():
            if o_grads[k] is None:
                continue
            assert rel_err(v.cpu().numpy(), o_grads[k].numpy()) < TOL_GRAD, (k, rep)
    if counts in ("integer", "ragged_genes"):   # the same batch as uint8 rows: the products expand them in registers -- the same bits
        xc = x.clamp(max=255)
        out_f = step.run(xc.cuda(), ezd, eld)
        ref = {k: out_f[k].clone() for k in ("recon", "qz_m", "library", "kl_z")}
        g_ref = arena.flat_grad.clone()
        out_8 = step.run(xc.to(torch.uint8).cuda(), ezd, eld)
        for k, v in ref.items():
            assert torch.equal(out_8[k], v), k
        assert torch.equal(arena.flat_grad, g_ref)
    if counts == "fractional":   # the hint that skips the optimistic attempt (vae_step_config.fractional_counts): same step
        ref = {k: out[k].clone() for k in ("recon", "qz_m", "library")}
        g_ref = arena.flat_grad.clone()
        out_h = step.run(xd, ezd, eld, fractional_counts=True)
        for k, v in ref.items():
            assert rel_err(out_h[k].cpu().numpy(), v.cpu().numpy()) < 1e-5, k
        assert rel_err(arena.flat_grad.cpu().numpy(), g_ref.cpu().numpy()) < 1e-5
    # a clean batch after an offending one: stale flags must not survive
    x2 = O.synthetic_counts(B, G, seed=27, zero_inflate=0.3)
    o2, _ = O.training_step_with_grads({k: v.double() for k, v in p.items()}, x2.double(), eps_z.double(), eps_l.double(),
                                       n_layers=1, likelihood="zinb", l_mean=lm, l_var=lv)
    out2 = step.run(x2.cuda(), ezd, eld)
    assert rel_err(out2["recon"].cpu().numpy(), o2["recon"].numpy()) < TOL_CELL
    assert rel_err(out2["qz_m"].cpu().numpy(), o2["qz_m"].numpy()) < TOL_CELL


@pytest.mark.parametrize("widths", [(256, 128, 256, 30), (192, 64, 320, 17), (512, 256, 576, 32), (128, 128, 128, 10)],
                         ids=["cfg2_widths", "ragged_lanes", "wide", "narrow"])
@pytest.mark.parametrize("precision", [0, 3, 1], ids=["bf16x3", "f16x2", "bf16"])
@pytest.mark.parametrize("split", [False, True], ids=["one_pass", "cell_chunks"])
def test_wave_per_cell_middle_against_oracle(widths, precision, split, monkeypatch):
    """cell_chunks: the small weight-gradient reductions cut into chunks of 512 cells + the combine launch, the form that
    batches from 2048 cells on run (middle.hip, small_wgrad_kernel), forced here at B = 1024.
    From 768 cells on, a 1-layer step runs the wave-per-cell middle kernels (middle.hip: one wave owns a cell, weights
    staged once per workgroup, DPP reductions); the workgroup-per-cell kernels serve smaller batches and are what the
    other step tests see.  B = 1024 with hidden dropout and a gradient arriving from cell heads (dz_extra), every
    instantiated (elements per lane) class, vs the oracle fed the host-regenerated masks."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    from helpers import dropout_multiplier
    H, Hl, Hd, L = widths
    B, G = 1024, 1600
    if split:
        monkeypatch.setenv("VAE_SMALL_WGRAD_SPLIT_MIN_B", "512")
    p_drop, seed, rng_step = 0.25, 424242, 3
    props = {"Dx": {"discrete": True, "values": [0, 1, 2], "stop_grad": False}}
    p = O.init_params(G, H, Hd, Hl, L, 1, seed=15, cell_properties=props)
    x = O.synthetic_counts(B, G, seed=15, zero_inflate=0.3)
    g = torch.Generator().manual_seed(16)
    eps_z, eps_l = torch.randn(B, L, generator=g), torch.randn(B, 1, generator=g)
    targets = torch.randint(0, 3, (B, 1), generator=g).float()
    mask = (torch.rand(B, 1, generator=g) > 0.2).float()
    lm, lv = O.library_prior(x)
    masks = {"z_encoder": [torch.from_numpy(dropout_multiplier(p_drop, seed, rng_step, 0, B, H))],
             "decoder": [torch.from_numpy(dropout_multiplier(p_drop, seed, rng_step, 2, B, Hd))]}
    bf16 = precision == 1
    cast = (lambda t: t.float()) if bf16 else (lambda t: t.double())
    arena = ParamArena(p, 1, "zinb", "cuda")
    step = ElboStep(arena, l_mean=lm, l_var=lv, precision=precision, dropout_rate=p_drop, seed=seed)
    kw = dict(n_layers=1, likelihood="zinb", l_mean=lm, l_var=lv, cell_properties=props, grad_reverse={}, cell_targets=cast(targets),
              cell_mask=cast(mask), drop_masks={k: [cast(m) for m in v] for k, v in masks.items()})
    if bf16:
        kw.update(gemm_dtype=torch.bfloat16, logits_bf16=step.bf16_activations(B, train=True))
    o_out, o_grads = O.training_step_with_grads({k: cast(v) for k, v in p.items()}, cast(x), cast(eps_z), cast(eps_l), **kw)
    from vae_amd.modules import CellDecoder
    from vae_amd.task import cell_loss
    dec = CellDecoder(L, props, grad_reverse_dict={}, use_hidden_layer=False).cuda()
    for leaf in ("weight", "bias"):
        q = getattr(dec.cell_mlp["Dx"], leaf)
        q.data, q.grad = arena.p(f"cell_decoder.cell_mlp.Dx.{leaf}"), arena.g(f"cell_decoder.cell_mlp.Dx.{leaf}")
    t, m = targets.cuda(), mask.cuda()
    out = step.run(x.cuda(), eps_z.cuda(), eps_l.cuda(), cell_head_fn=lambda z: cell_loss(dec(z), props, t, m), rng_step=rng_step)
    torch.cuda.synchronize()
    tc, tg = (2e-3, 3e-2) if bf16 else (TOL_CELL, TOL_GRAD)
    for k in ("qz_m", "z", "library", "recon", "kl_z", "kl_l"):
        ref = o_out[k].double().numpy()
        kt = tc if not bf16 else (10 * tc if k in ("kl_z", "kl_l") else 3 * tc)   # bf16: latent entries near zero vs the max-norm
        assert rel_err(out[k].cpu().numpy().reshape(ref.shape), ref) < kt, k
    assert abs(float(out["loss"]) - float(o_out["loss"])) < tc * abs(float(o_out["loss"]))
    n = 0
    for k, v in arena.grads().items():
        if v is None or o_grads[k] is None or k.startswith("cell_decoder"):
            continue
        assert rel_err(v.cpu().numpy(), o_grads[k].double().numpy()) < tg, k
        n += 1
    assert n >= 20


# ------------------------------------------------------------------------------------------------
# BASELINE.json configs[1] at FULL size (B=512, G=20000, hidden 256/128, latent 30, ZINB, fp32):
# direct comparison with the fp64 oracle plus size-independent properties of the step.
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module", params=FP32_MODES, ids=["bf16x3", "f16x2"])
def full_cfg2(request):
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    B, G, H, Hd, Hl, L = 512, 20000, 256, 256, 128, 30
    p = O.init_params(G, H, Hd, Hl, L, 1, seed=43)
    x = O.synthetic_counts(B, G, seed=43, zero_inflate=0.3)
    g = torch.Generator().manual_seed(44)
    eps_z, eps_l = torch.randn(B, L, generator=g), torch.randn(B, 1, generator=g)
    lm, lv = O.library_prior(x)
    arena = ParamArena(p, 1, "zinb", "cuda")
    step = ElboStep(arena, l_mean=lm, l_var=lv, precision=request.param)
    return dict(O=O, p=p, x=x, eps_z=eps_z, eps_l=eps_l, lm=lm, lv=lv, arena=arena, step=step, B=B, L=L,
                xd=x.cuda(), ezd=eps_z.cuda(), eld=eps_l.cuda())


def test_full_size_cfg2_against_fp64_oracle(full_cfg2):
    c = full_cfg2
    torch.set_num_threads(16)
    o_out, o_grads = c["O"].training_step_with_grads(
        {k: v.double() for k, v in c["p"].items()}, c["x"].double(), c["eps_z"].double(), c["eps_l"].double(),
        n_layers=1, likelihood="zinb", l_mean=c["lm"], l_var=c["lv"])
    out = c["step"].run(c["xd"], c["ezd"], c["eld"])
    rec = out["recon"].cpu().double().numpy()
    assert np.max(np.abs(rec - o_out["recon"].numpy()) / np.abs(o_out["recon"].numpy())) < TOL_CELL
    for k in ("kl_z", "kl_l", "z", "library"):
        assert rel_err(out[k].cpu().numpy().reshape(o_out[k].shape), o_out[k].numpy()) < TOL_CELL, k
    assert abs(float(out["loss"]) - float(o_out["loss"])) < TOL_CELL * abs(float(o_out["loss"]))
    n = 0
    for k, v in c["arena"].grads().items():
        if o_grads[k] is None:
            continue
        assert rel_err(v.cpu().numpy(), o_grads[k].numpy()) < TOL_GRAD, k
        n += 1
    assert n >= 15


def test_full_size_step_is_bit_reproducible(full_cfg2):
    """No float atomics on the 1-layer path and every gradient overwritten: a second run over a
    poisoned gradient arena gives identical bits in every gradient tensor of the ELBO path."""
    c = full_cfg2
    c["step"].run(c["xd"], c["ezd"], c["eld"])
    first = {k: v.clone() for k, v in c["arena"].grads().items() if v is not None and not k.startswith("cell_decoder")}
    c["arena"].flat_grad.fill_(7.0)
    out = c["step"].run(c["xd"], c["ezd"], c["eld"])
    l2 = out["loss"].clone()
    second = c["arena"].grads()
    assert len(first) >= 15
    for k, v in first.items():
        assert torch.equal(v, second[k]), k
    out = c["step"].run(c["xd"], c["ezd"], c["eld"])
    assert torch.equal(l2, out["loss"])


def test_full_size_cells_are_independent(full_cfg2):
    """Per-cell terms of the first 64 cells do not depend on the other 448 cells of the batch."""
    c = full_cfg2
    full = {k: v.clone() for k, v in c["step"].run(c["xd"], c["ezd"], c["eld"], backward=False).items()
            if k in ("recon", "kl_z", "kl_l", "z")}
    part = c["step"].run(c["xd"][:64], c["ezd"][:64].contiguous(), c["eld"][:64].contiguous(), backward=False)
    for k, v in full.items():
        assert rel_err(part[k].cpu().numpy(), v[:64].cpu().numpy()) < TOL_CELL, k


def test_full_size_loss_scale_is_linear(full_cfg2):
    """loss_scale = 4 (a power of two) scales every gradient of the ELBO path by 4."""
    c = full_cfg2
    c["step"].run(c["xd"], c["ezd"], c["eld"])
    g1 = {k: v.clone() for k, v in c["arena"].grads().items() if v is not None and not k.startswith("cell_decoder")}
    c["step"].run(c["xd"], c["ezd"], c["eld"], loss_scale=4.0)
    g4 = c["arena"].grads()
    for k, v in g1.items():
        assert rel_err(g4[k].cpu().numpy(), 4.0 * v.cpu().numpy()) < 1e-6, k


# ------------------------------------------------------------------------------------------------
# uint8 counts (the reference stores counts as uint8 and expands them on the host,
# create_dataset.py:131-142 / data.py:406-415; here the expansion happens inside the kernels)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("likelihood,n_layers", [("zinb", 1), ("nb", 2)])
def test_uint8_counts_give_the_same_bits_as_float_counts(likelihood, n_layers):
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    B, G, H, Hd, Hl, L = 64, 2000, 64, 64, 32, 10
    p = O.init_params(G, H, Hd, Hl, L, n_layers, seed=9)
    x = O.synthetic_counts(B, G, seed=9, zero_inflate=0.3).clamp_(max=255)
    g = torch.Generator().manual_seed(10)
    eps_z, eps_l = torch.randn(B, L, generator=g).cuda(), torch.randn(B, 1, generator=g).cuda()
    lm, lv = O.library_prior(x)
    res = []
    for xin in (x.cuda(), x.to(torch.uint8).cuda()):
        arena = ParamArena(p, n_layers, likelihood, "cuda")
        out = ElboStep(arena, l_mean=lm, l_var=lv).run(xin, eps_z, eps_l)
        res.append(({k: v.clone() for k, v in out.items() if k in ("recon", "kl_z", "kl_l", "loss", "z")},
                    {k: v.clone() for k, v in arena.grads().items() if v is not None and not k.startswith("cell_decoder")}))
    for k, v in res[0][0].items():
        assert torch.equal(v, res[1][0][k]), k
    if likelihood == "zinb":
        for k, v in res[0][1].items():
            assert torch.equal(v, res[1][1][k]), k
    else:   # the NB backward's two instantiations (float / u8 counts) contract their multiply-adds differently: last-bit
            # differences in dlogits (measured 4e-7 of a tensor's max), nothing to do with the counts themselves
        for k, v in res[0][1].items():
            assert rel_err(res[1][1][k].cpu().numpy(), v.cpu().numpy()) < 1e-6, k


@pytest.mark.parametrize("n_layers,n_cov,B", [(2, 0, 700), (3, 0, 64), (1, 5, 1500)])
def test_unfused_step_is_bit_reproducible(n_layers, n_cov, B):
    """The launch sequence of multi-layer / covariate steps: a second run over a poisoned gradient arena gives identical bits
    in every gradient tensor (column sums per workgroup + one ordered reduction, every gradient overwritten)."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    G, H, Hd, Hl, L = 1200, 128, 96, 64, 10
    p = O.init_params(G, H, Hd, Hl, L, n_layers, seed=5)
    if n_cov:   # widen the three first-layer weights by the covariate columns (modules.py:102-107)
        gc = torch.Generator().manual_seed(7)
        for k in ("z_encoder.encoder.fc_layers.0.weight", "l_encoder.encoder.fc_layers.0.weight", "decoder.px_decoder.fc_layers.0.weight"):
            p[k] = torch.cat((p[k], torch.randn(p[k].shape[0], n_cov, generator=gc) * 0.05), dim=1).contiguous()
    x = O.synthetic_counts(B, G, seed=5, zero_inflate=0.3).cuda()
    g = torch.Generator().manual_seed(6)
    eps_z, eps_l = torch.randn(B, L, generator=g).cuda(), torch.randn(B, 1, generator=g).cuda()
    cov = None
    if n_cov:
        cov = torch.zeros(B, n_cov)
        cov[torch.arange(B), torch.randint(0, n_cov, (B,), generator=g)] = 1.0
        cov = cov.cuda()
    arena = ParamArena(p, n_layers, "zinb", "cuda")
    step = ElboStep(arena, l_mean=5.0, l_var=1.5)
    step.run(x, eps_z, eps_l, batch_cov=cov)
    first = {k: v.clone() for k, v in arena.grads().items() if v is not None and not k.startswith("cell_decoder")}
    assert all(torch.isfinite(v).all() for v in first.values())
    arena.flat_grad.fill_(7.0)
    step.run(x, eps_z, eps_l, batch_cov=cov)
    second = arena.grads()
    assert len(first) >= 15
    for k, v in first.items():
        assert torch.equal(v, second[k]), k


def test_uint8_counts_need_vector_alignment():
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    p = O.init_params(77, 12, 10, 6, 3, 1, seed=3)      # G = 77 is not a multiple of 4
    arena = ParamArena(p, 1, "zinb", "cuda")
    x = torch.ones(5, 77, dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="u8"):
        ElboStep(arena, l_mean=4.0, l_var=1.0).run(x, torch.zeros(5, 3, device="cuda"), torch.zeros(5, 1, device="cuda"))


def test_weight_maxima_from_the_optimiser_replace_the_scan():
    """VERDICT r1 #6b: the fused optimiser leaves the partial maxima of the two gene-sized weight matrices behind while it
    stores the updated weights (vae_adamw_step_wmax); the two-f16-term step then skips its 72 MB scan.  The maxima must
    equal max|w| of the updated weights exactly, and the step must give the SAME BITS with either source."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena, _enc
    from vae_amd.optim import FusedAdamW
    from vae_amd.step import ElboStep
    B, G, H, Hd, Hl, L = 128, 4000, 128, 128, 64, 10
    p = O.init_params(G, H, Hd, Hl, L, 1, seed=21)
    x = O.synthetic_counts(B, G, seed=21, zero_inflate=0.3).cuda()
    g = torch.Generator().manual_seed(22)
    ez, el = torch.randn(B, L, generator=g).cuda(), torch.randn(B, 1, generator=g).cuda()
    arena = ParamArena(p, 1, "zinb", "cuda")
    step = ElboStep(arena, l_mean=6.0, l_var=1.0)
    opt = FusedAdamW(arena, lr=1e-2, weight_decay=0.1, max_grad_norm=0.5, track_weight_maxima=True)
    w_enc = arena.pcat(_enc("z_encoder", 0, "w"), _enc("l_encoder", 0, "w"))
    w_heads = arena.pcat("decoder.px_scale_decoder.0.weight", "decoder.px_dropout_decoder.weight")
    for it in range(3):
        assert float(opt.weight_maxima[0].max()) == float(w_enc.abs().max())
        assert float(opt.weight_maxima[1].max()) == float(w_heads.abs().max())
        assert float(opt.weight_maxima.min()) >= 0.0
        step.use_weight_maxima(None)
        a = step.run(x, ez, el)
        la, ga = a["loss"].clone(), arena.flat_grad.clone()
        step.use_weight_maxima(opt)
        b = step.run(x, ez, el)
        assert torch.equal(la, b["loss"]) and torch.equal(ga[:arena.used_end], arena.flat_grad[:arena.used_end]), it
        opt.step()                                   # lr 1e-2: the weights -- and their maxima -- really move
    arena.flat.mul_(3.0)                             # a change behind the optimiser's back ...
    arena.touch()                                    # ... is announced (broadcast_parameters / the load hook do), and the
    step.run(x, ez, el)                              # next step that consumes the maxima refreshes them
    assert float(opt.weight_maxima[1].max()) == float(w_heads.abs().max())
    arena.flat.mul_(0.5)
    opt.refresh_weight_maxima()                      # or explicitly
    assert float(opt.weight_maxima[0].max()) == float(w_enc.abs().max())


@pytest.mark.parametrize("n_layers,n_cov", [(1, 0), (2, 0), (1, 5)])
def test_bf16_weight_images_from_the_optimiser_replace_the_cast(n_layers, n_cov):
    """The bf16-mode twin of the test above: vae_adamw_step_bf16 leaves bf16 images (nearest even) of the two gene-sized
    weight matrices while it stores the updated weights; a bf16-storage step then skips its own cast launch.  The images
    must equal torch's bf16 rounding of the updated weights bit for bit -- row padding of a covariate model included in the
    layout -- the update itself must be the plain vae_adamw_step's, and the step must give the SAME BITS with either source."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena, _enc
    from vae_amd.optim import FusedAdamW
    from vae_amd.step import ElboStep
    B, G, H, Hd, Hl, L = 256, 4000, 128, 128, 64, 10
    p = O.init_params(G, H, Hd, Hl, L, n_layers, seed=23)
    cov = None
    if n_cov:
        gc = torch.Generator().manual_seed(7)
        for k in ("z_encoder.encoder.fc_layers.0.weight", "l_encoder.encoder.fc_layers.0.weight", "decoder.px_decoder.fc_layers.0.weight"):
            p[k] = torch.cat((p[k], torch.randn(p[k].shape[0], n_cov, generator=gc) * 0.05), dim=1).contiguous()
        cov = torch.zeros(B, n_cov)
        cov[torch.arange(B), torch.randint(0, n_cov, (B,), generator=gc)] = 1.0
        cov = cov.cuda()
    x = O.synthetic_counts(B, G, seed=23, zero_inflate=0.3).cuda()
    g = torch.Generator().manual_seed(24)
    ez, el = torch.randn(B, L, generator=g).cuda(), torch.randn(B, 1, generator=g).cuda()
    arena, twin = ParamArena(p, n_layers, "zinb", "cuda"), ParamArena(p, n_layers, "zinb", "cuda")
    step = ElboStep(arena, l_mean=6.0, l_var=1.0, precision=1)
    assert step.bf16_activations(B)
    opt = FusedAdamW(arena, lr=1e-2, weight_decay=0.1, max_grad_norm=0.5, track_bf16_images=True)
    plain = FusedAdamW(twin, lr=1e-2, weight_decay=0.1, max_grad_norm=0.5)
    spans = opt._max_spans
    assert spans[0][1] - spans[0][0] == (H + Hl) * arena.ld_enc_w and spans[1][1] - spans[1][0] == 2 * G * Hd
    with pytest.raises(ValueError):
        FusedAdamW(arena, track_weight_maxima=True, track_bf16_images=True)
    with pytest.raises(ValueError):
        step.use_weight_images((opt.weight_images[0], opt.weight_images[1][:-8]))
    for it in range(3):
        for img, (b, e) in zip(opt.weight_images, spans):
            assert torch.equal(img.view(torch.int16), arena.flat[b:e].to(torch.bfloat16).view(torch.int16)), it
        step.use_weight_images(None)
        a = step.run(x, ez, el, batch_cov=cov)
        la, ga = a["loss"].clone(), arena.flat_grad.clone()
        step.use_weight_images(opt)
        b_ = step.run(x, ez, el, batch_cov=cov)
        assert torch.equal(la, b_["loss"]) and torch.equal(ga[:arena.used_end], arena.flat_grad[:arena.used_end]), it
        twin.flat_grad.copy_(arena.flat_grad)
        opt.step()                                   # lr 1e-2: the weights -- and their images -- really move
        plain.step()
        assert torch.equal(arena.flat, twin.flat), it
    arena.flat.mul_(3.0)                             # a change behind the optimiser's back needs a refresh
    opt.refresh_weight_images()
    b, e = spans[1]
    assert torch.equal(opt.weight_images[1].view(torch.int16), arena.flat[b:e].to(torch.bfloat16).view(torch.int16))


@pytest.mark.parametrize("balanced", [False, True], ids=["plain", "balance_classes"])
@pytest.mark.parametrize("B", [37, 700, 2304])
def test_native_cell_heads_match_torch_heads(B, balanced):
    """vae_cell_heads_step (linear CellDecoder heads + LitVAE._cell_loss, modules.py:297-324, task.py:419-451) against the
    torch modules under autograd in fp64: loss, d z and every head's weight / bias gradient -- two discrete heads (one
    behind a gradient-reversal layer), a stop_grad head that must leave d z alone, a continuous head, ignore_index rows,
    masked cells, a loss scale; B = 2304 runs the chunked reduction + combine launch.  balance_classes: per-class
    cross-entropy weights clip(1 / freq, 0.1, 10) (task.py:129-132) on the discrete heads."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.modules import CellDecoder
    from vae_amd.step import CellHeads
    from vae_amd.task import cell_loss, class_balance_weights
    L = 30
    props = {"Dx": {"discrete": True, "values": [0, 1, 2], "stop_grad": False, "freq": [0.7, 0.25, 0.05]},
             "Adv": {"discrete": True, "values": list(range(7)), "stop_grad": False, "freq": [0.02, 0.3, 0.2, 0.2, 0.1, 0.1, 0.08]},
             "Frozen": {"discrete": True, "values": [0, 1], "stop_grad": True, "freq": [0.5, 0.5]},
             "Age": {"discrete": False, "values": [-1], "stop_grad": False}}
    rev = {"Adv": 0.3}
    cw = class_balance_weights(props) if balanced else None
    if balanced:   # 1 / 0.02 = 50 clips to 10; 1 / 0.7 = 1.43 stays
        assert float(cw["Adv"][0]) == 10.0 and abs(float(cw["Dx"][0]) - 1 / 0.7) < 1e-6 and "Age" not in cw
    p = O.init_params(64, 16, 16, 8, L, 1, seed=21, cell_properties=props)
    arena = ParamArena(p, 1, "zinb", "cuda")
    g = torch.Generator().manual_seed(22)
    z = torch.randn(B, L, generator=g)
    targets = torch.stack([torch.randint(0, len(v["values"]), (B,), generator=g).float() if v["discrete"] else torch.randn(B, generator=g)
                           for v in props.values()], dim=1)
    targets[::5, 1] = -100.0
    targets[1::9, 0] = -100.0
    mask = (torch.rand(B, len(props), generator=g) > 0.2).float()
    scale = 1.7
    # torch side, fp64
    dec = CellDecoder(L, props, grad_reverse_dict=rev, use_hidden_layer=False).double()
    for k in props:
        dec.cell_mlp[k].weight.data = p[f"cell_decoder.cell_mlp.{k}.weight"].double()
        dec.cell_mlp[k].bias.data = p[f"cell_decoder.cell_mlp.{k}.bias"].double()
    z64 = z.double().requires_grad_(True)
    ref = cell_loss(dec(z64), props, targets.double(), mask.double(), class_weights=cw)
    (ref * scale).backward()
    heads = CellHeads(arena, props, rev, class_weights=cw)
    dz = torch.full((B, L), float("nan"), device="cuda")
    loss = heads.run(z.cuda(), targets.cuda(), mask.cuda(), scale, dz)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(ref)) < 2e-6 * abs(float(ref))
    assert rel_err(dz.cpu().numpy(), z64.grad.numpy()) < 1e-5
    for k in props:
        for leaf in ("weight", "bias"):
            got = arena.g(f"cell_decoder.cell_mlp.{k}.{leaf}").cpu().numpy()
            assert rel_err(got, getattr(dec.cell_mlp[k], leaf).grad.numpy()) < 1e-5, (k, leaf)
    # forward only: the loss alone, no gradient written
    before = arena.flat_grad.clone()
    loss2 = heads.run(z.cuda(), targets.cuda(), mask.cuda())
    torch.cuda.synchronize()
    assert abs(float(loss2) - float(ref)) < 2e-6 * abs(float(ref)) and torch.equal(before, arena.flat_grad)
    # a head with a hidden layer is refused (it runs through cell_head_fn)
    p2 = dict(p)
    p2["cell_decoder.cell_mlp.Dx.0.weight"] = p2.pop("cell_decoder.cell_mlp.Dx.weight")
    with pytest.raises((ValueError, KeyError)):
        CellHeads(ParamArena(p2, 1, "zinb", "cuda"), props, rev)
