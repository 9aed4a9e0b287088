"""CPU: the oracle (oracle/vae_oracle.py) reproduces the golden vectors made by the REFERENCE.

This is what pins the oracle (task prompt ③): every e2e case and the likelihood grid were
produced by importing /root/reference/src in the build container (tests/golden/make_goldens.py).
"""
import copy

import numpy as np
import pytest
import torch

from golden_inputs import E2E_CASES, WIDE_CASE, build_case_inputs, LIKELIHOOD_GRID
from helpers import load_golden, rel_err, sample_like_golden
from oracle import vae_oracle as O

ALL_CASES = E2E_CASES + [WIDE_CASE]


def run_oracle(case, dtype, gemm_dtype=None):
    case = copy.deepcopy(case)
    params, x, eps_z, eps_l, targets, mask = build_case_inputs(case)
    p = {k: v.to(dtype) for k, v in params.items()}
    return O.training_step_with_grads(
        p, x.to(dtype), eps_z.to(dtype), eps_l.to(dtype),
        n_layers=case["n_layers"], likelihood=case["likelihood"],
        l_mean=float(np.float32(case["l_mean"])), l_var=float(np.float32(case["l_var"])),
        kl_weight=case["kl_weight"], gene_loss_coeff=case["gene_loss_coeff"],
        cell_properties=case["cell_properties"] if case["use_cell_loss"] else None,
        grad_reverse=case["grad_reverse"], cell_targets=targets, cell_mask=mask.to(dtype),
        gemm_dtype=gemm_dtype,
    )


@pytest.mark.parametrize("case", ALL_CASES, ids=[c["name"] for c in ALL_CASES])
@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-11), ("f32", torch.float32, 2e-5)])
def test_oracle_matches_reference_goldens(case, tag, dtype, tol):
    gold = load_golden(f"e2e_{case['name']}.npz")
    out, grads = run_oracle(case, dtype)
    G, stride = case["G"], case["sample_stride"]
    for k, v in out.items():
        g = gold[f"{tag}/out/{k}"]
        assert rel_err(sample_like_golden(v, G, stride), g) < tol, k
    n_checked = 0
    for k, v in grads.items():
        if f"{tag}/gradnone/{k}" in gold.files:
            assert v is None, f"{k}: reference has no grad here"
            continue
        g = gold[f"{tag}/grad/{k}"]
        assert v is not None, k
        assert rel_err(sample_like_golden(v, G, stride), g) < tol * 5, k
        n_checked += 1
    assert n_checked >= 15


def test_unused_parameters_have_no_grad():
    """px_r_decoder.* never gets a gradient; px_dropout_decoder.* none under NB."""
    _, grads = run_oracle(E2E_CASES[0], torch.float32)
    assert grads["decoder.px_r_decoder.weight"] is None
    assert grads["decoder.px_dropout_decoder.weight"] is None
    _, grads = run_oracle(E2E_CASES[1], torch.float32)
    assert grads["decoder.px_dropout_decoder.weight"] is not None


@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-12), ("f32", torch.float32, 1e-6)])
def test_likelihood_grid(tag, dtype, tol):
    gold = load_golden("likelihood_grid.npz")
    x = torch.tensor(gold["x"], dtype=dtype)
    mu = torch.tensor(gold["mu"], dtype=dtype, requires_grad=True)
    th = torch.tensor(gold["theta"], dtype=dtype, requires_grad=True)
    pi = torch.tensor(gold["pi"], dtype=dtype, requires_grad=True)
    z = O.log_zinb_positive(x, mu, th, pi)
    z.sum().backward()
    np.testing.assert_allclose(z.detach().numpy(), gold[f"{tag}/zinb"], rtol=tol, atol=tol)
    np.testing.assert_allclose(mu.grad.numpy(), gold[f"{tag}/zinb_dmu"], rtol=tol * 10, atol=tol)
    np.testing.assert_allclose(th.grad.numpy(), gold[f"{tag}/zinb_dtheta"], rtol=tol * 10, atol=tol)
    np.testing.assert_allclose(pi.grad.numpy(), gold[f"{tag}/zinb_dpi"], rtol=tol * 10, atol=tol)
    mu.grad = None; th.grad = None
    n = O.log_nb_positive(x, mu, th)
    n.sum().backward()
    np.testing.assert_allclose(n.detach().numpy(), gold[f"{tag}/nb"], rtol=tol, atol=tol)
    np.testing.assert_allclose(mu.grad.numpy(), gold[f"{tag}/nb_dmu"], rtol=tol * 10, atol=tol)
    np.testing.assert_allclose(th.grad.numpy(), gold[f"{tag}/nb_dtheta"], rtol=tol * 10, atol=tol)


def test_bf16_mode_is_close_to_fp32():
    """The bf16-GEMM-operand oracle stays within a stated loose tolerance of fp32 (2e-2 on loss)."""
    o32, _ = run_oracle(E2E_CASES[1], torch.float32)
    o16, _ = run_oracle(E2E_CASES[1], torch.float32, gemm_dtype=torch.bfloat16)
    assert abs(float(o16["loss"]) - float(o32["loss"])) / abs(float(o32["loss"])) < 2e-2


def test_host_philox_known_answers():
    """tests/helpers.philox_word0 (the host restatement of the library's dropout stream) against the
    Philox4x32-10 known-answer vectors published with Random123 (kat_vectors: all-zero and all-ones
    counter / key)."""
    import numpy as np
    from helpers import dropout_multiplier, philox_word0
    assert int(philox_word0(0, 0, 0, np.array([0], dtype=np.uint64))[0]) == 0x6627E8D5
    ones = np.array([0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
    assert int(philox_word0(0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, ones)[0]) == 0x408F276D
    m = dropout_multiplier(0.5, 1234, 7, 0, 512, 256)
    assert set(np.unique(m)) == {0.0, 2.0} and abs((m > 0).mean() - 0.5) < 0.01


def test_oracle_with_batch_covariates_matches_the_reference():
    """Row f4: masked one-hot batch covariates in both encoders and the decoder (modules.py:121-130), fp64, against the
    reference's own run (tests/golden/make_goldens_f4.py)."""
    import copy
    from golden_inputs import F4_CASES, build_f4_inputs
    case = copy.deepcopy(F4_CASES[0])
    gold = load_golden("f4_batchcov.npz")
    x, eps_z, eps_l, bl, bm, targets, mask = build_f4_inputs(case)
    p = {k[len("param/"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("param/")}
    cov = O.batch_covariates(bl, bm, [len(v["values"]) for v in case["batch_properties"].values()]).double()
    assert cov.shape == (case["B"], 5) and float(cov[2].abs().sum()) == 0.0 and float(cov[1, :3].abs().sum()) == 0.0
    out, grads = O.training_step_with_grads(p, x, eps_z[0], eps_l, n_layers=case["n_layers"], likelihood="zinb",
                                            l_mean=case["l_mean"], l_var=case["l_var"], gene_loss_coeff=case["gene_loss_coeff"],
                                            cell_properties=case["cell_properties"], grad_reverse=case["grad_reverse"],
                                            cell_targets=targets.double(), cell_mask=mask.double(), cov=cov)
    for k in ("qz_m", "z", "library", "recon", "kl_z", "kl_l", "loss", "cell_loss"):
        g = gold[f"out/{k}"]
        assert rel_err(out[k].numpy().reshape(g.shape), g) < 1e-11, k
    n = 0
    for k, g in grads.items():
        if f"gradnone/{k}" in gold.files:
            assert g is None, k
            continue
        assert rel_err(g.numpy(), gold[f"grad/{k}"]) < 1e-10, k
        n += 1
    assert n >= 30
