"""CPU: host-side mirror of the reference interface -- constructor / state_dict contract, the
parameter arena, losses, LR schedule.  (No kernels run here.)"""
import io
import contextlib

import numpy as np
import pytest
import torch

from golden_inputs import E2E_CASES, INIT_CASE, build_case_inputs
from helpers import load_golden


def make_vae(c, **kw):
    from vae_amd.networks import VAE
    return VAE(c["G"], n_hidden=c["H"], n_hidden_decoder=c["Hd"], n_hidden_library=c["Hl"], n_latent=c["L"],
               n_latent_cell_decoder=c["L"], n_layers=c["n_layers"], cell_properties=c["cell_properties"],
               grad_reverse_dict=c["grad_reverse"], cell_decoder_hidden_layer=False, **kw)


def test_initial_weights_equal_the_reference_under_seed_43():
    """Same sub-module creation order and initialisers as networks.py:162-231: identical RNG use."""
    gold = load_golden("init_seed43.npz")
    torch.manual_seed(43)
    net = make_vae(INIT_CASE, dropout_rate=0.1, input_dropout_rate=0.0)
    sd = net.state_dict()
    assert list(sd.keys()) == list(gold.files)            # names AND order of the state_dict
    for k in gold.files:
        assert torch.equal(sd[k], torch.from_numpy(gold[k])), k


def test_state_dict_keys_match_goldens():
    c = E2E_CASES[3]
    gold = load_golden(f"e2e_{c['name']}.npz")
    ref_keys = {k.split("/", 2)[2] for k in gold.files if k.startswith("f32/grad")}
    net = make_vae(c, dropout_rate=0.0)
    assert ref_keys == {k for k, _ in net.named_parameters()}


def test_bad_dispersion_raises_value_error():
    from vae_amd.networks import VAE
    with pytest.raises(ValueError):
        VAE(8, cell_properties={"A": {"discrete": True, "values": [0, 1], "stop_grad": False}}, dispersion="nope")
    with pytest.raises(TypeError):
        VAE(8, cell_properties=None)                      # the reference raises here too (networks.py:179)


def test_arena_layout_and_views():
    from vae_amd.arena import ParamArena
    c = dict(E2E_CASES[5])
    params, *_ = build_case_inputs(c)
    a = ParamArena({k: v.float() for k, v in params.items()}, c["n_layers"], c["likelihood"], "cpu")
    for k, v in params.items():
        assert torch.equal(a.p(k), v.float())
    w = a.pcat("decoder.px_scale_decoder.0.weight", "decoder.px_dropout_decoder.weight")
    assert w.shape == (2 * c["G"], c["Hd"]) and torch.equal(w[c["G"]:], params["decoder.px_dropout_decoder.weight"].float())
    e = a.pcat("z_encoder.encoder.fc_layers.0.weight", "l_encoder.encoder.fc_layers.0.weight")
    assert e.shape == (c["H"] + c["Hl"], c["G"])
    with pytest.raises(RuntimeError):
        a.pcat("px_r", "z_encoder.mean_encoder.bias")
    g = a.grads()
    assert g["decoder.px_r_decoder.weight"] is None and g["decoder.px_dropout_decoder.weight"] is not None
    a.flat_grad.fill_(1.0)
    a.zero_small_grads()
    assert float(a.g("px_r").sum()) == 0.0 and float(a.g("decoder.px_scale_decoder.0.weight").min()) == 1.0
    nb = ParamArena({k: v.float() for k, v in params.items()}, 1, "nb", "cpu")
    assert nb.grads()["decoder.px_dropout_decoder.bias"] is None
    with pytest.raises(KeyError):
        ParamArena({"px_r": torch.zeros(3)}, 1, "zinb", "cpu")


def test_lr_schedule_and_optimizer_groups():
    from vae_amd.task import LitVAE
    c = E2E_CASES[3]
    net = make_vae(c, dropout_rate=0.0)
    task = LitVAE(net, cell_properties=c["cell_properties"], learning_rate=5e-4, warmup_steps=1000.0, weight_decay=0.1,
                  library_mean=1.0, library_var=1.0, n_epochs_kl_warmup=None)
    cfg = task.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    assert opt.defaults["betas"] == (0.9, 0.998) and opt.defaults["eps"] == 1e-7
    wd = {round(g["weight_decay"], 6) for g in opt.param_groups}
    assert wd == {0.1, 0.0}
    n_enc = sum(1 for n, _ in net.named_parameters() if "z_encoder" in n or "cell_decoder" in n)
    assert len(opt.param_groups[0]["params"]) == n_enc
    w = 1000.0
    for s in (0, 10, 999, 1000, 5000):
        expect = np.sqrt(w) * min((s + 1) ** -0.5, (s + 1) * w ** -1.5)
        assert abs(sched.lr_lambda(s) - expect) < 1e-12
    assert task.kl_weight == 1.0
    task.n_epochs_kl_warmup, task.current_epoch = 20, 3
    assert abs(task.kl_weight - 0.2) < 1e-12


def test_losses_api():
    from vae_amd.losses import FocalLoss, GroupDRO, Poly1FocalLoss
    logits = torch.tensor([[2.0, -1.0, 0.5], [0.1, 0.2, 0.3]])
    tgt = torch.tensor([0, 2])
    fl = FocalLoss(3, gamma=2.0)(logits, tgt)
    logp = torch.log_softmax(logits, -1)
    smooth = torch.tensor([[0.95, 0.05, 0.0], [0.0, 0.05, 0.95]])
    assert torch.allclose(fl, (-smooth * logp * (1 - logp.exp()) ** 2).sum(-1), atol=1e-6)
    assert Poly1FocalLoss(3)(logits, tgt).shape == (2,)
    g = GroupDRO()
    g.update_batch(torch.tensor([1.0, 3.0, 2.0]), torch.tensor([1.0, 1.0, 1.0]), [0, 0, 61])
    g.update_weights()
    assert g.batch_count[0] == 0 and abs(g.running_loss[0] - 0.02) < 1e-9
    assert g.get_weights([0, 61]).shape == (2,)


def test_cell_loss_matches_oracle():
    """Product CellDecoder + cell_loss against the oracle's restatement (masked CE with
    ignore_index, MSE, grad-reverse, stop_grad) -- torch ops, runs on CPU."""
    from oracle import vae_oracle as O
    from vae_amd.modules import CellDecoder
    from vae_amd.task import cell_loss
    c = dict(E2E_CASES[3])
    params, x, eps_z, eps_l, targets, mask = build_case_inputs(c)
    with contextlib.redirect_stdout(io.StringIO()):
        dec = CellDecoder(c["L"], c["cell_properties"], grad_reverse_dict=c["grad_reverse"], use_hidden_layer=False)
    p32 = {k: v.float() for k, v in params.items()}
    for k in c["cell_properties"]:
        dec.cell_mlp[k].weight.data = p32[f"cell_decoder.cell_mlp.{k}.weight"].clone()
        dec.cell_mlp[k].bias.data = p32[f"cell_decoder.cell_mlp.{k}.bias"].clone()
    z = eps_z.float().clone().requires_grad_(True)
    loss = cell_loss(dec(z), c["cell_properties"], targets, mask)
    loss.backward()
    z2 = eps_z.float().clone().requires_grad_(True)
    ref = O.cell_loss(p32, z2, c["cell_properties"], c["grad_reverse"], targets, mask)
    ref.backward()
    assert torch.allclose(loss, ref, atol=1e-6)
    assert torch.allclose(z.grad, z2.grad, atol=1e-6)


def test_group_dro_cell_loss_matches_the_reference_sequence():
    """`use_gdro=True` (task.py:84-101, 419-451): seven consecutive `_cell_loss` calls across the step-500 switch-on, against
    a fixture made by running the reference's `losses.GroupDRO` through the same sequence
    (tests/golden/make_goldens_gdro.py): every step's loss and both trackers' weights / running losses after it."""
    from vae_amd.networks import VAE
    from vae_amd.task import LitVAE
    gold = load_golden("gdro_sequence.npz")
    props = {"Dementia": {"discrete": True, "values": [0, 1]}, "BRAAK_AD": {"discrete": False, "values": [-1]},
             "Sex": {"discrete": True, "values": [0, 1, 2]}}
    with contextlib.redirect_stdout(io.StringIO()):
        net = VAE(32, n_hidden=16, n_hidden_decoder=16, n_hidden_library=16, n_latent=4, n_latent_cell_decoder=4, n_layers=1,
                  cell_properties=props, grad_reverse_dict={})
    lit = LitVAE(net, cell_properties=props, use_gdro=True, library_mean=5.0, library_var=1.0)
    groups = [int(g) for g in gold["groups"]]
    for t, train_step in enumerate(gold["train_steps"]):
        lit.train_step = int(train_step)
        pred = {k: torch.from_numpy(gold[f"s{t}/pred/{k}"]) for k in props}
        loss = lit._cell_loss(pred, torch.from_numpy(gold[f"s{t}/vals"]), torch.from_numpy(gold[f"s{t}/mask"]),
                              [int(g) for g in gold[f"s{t}/group_idx"]])
        assert abs(float(loss) - float(gold[f"s{t}/cell_loss"])) < 1e-12 * abs(float(gold[f"s{t}/cell_loss"])), t
        for k in ("Dementia", "BRAAK_AD"):
            np.testing.assert_allclose([lit.gdro[k].weights[g] for g in groups], gold[f"s{t}/weights/{k}"], rtol=1e-13, atol=0)
            np.testing.assert_allclose([lit.gdro[k].running_loss[g] for g in groups], gold[f"s{t}/running/{k}"], rtol=1e-13, atol=0)
    # the weights did switch on: the last steps' losses differ from the unweighted sum
    from vae_amd.task import cell_loss
    t = len(gold["train_steps"]) - 1
    plain = cell_loss({k: torch.from_numpy(gold[f"s{t}/pred/{k}"]) for k in props}, props, torch.from_numpy(gold[f"s{t}/vals"]),
                      torch.from_numpy(gold[f"s{t}/mask"]))
    assert abs(float(plain) - float(gold[f"s{t}/cell_loss"])) > 1e-3
