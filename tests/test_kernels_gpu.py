"""GPU parity tests of the individual HIP kernels, called through the C ABI (vae_amd.ops),
against fp64 CPU computations and the reference-generated likelihood grid."""
import numpy as np
import pytest
import torch

from helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from vae_amd import ops as o
    return o


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


GEMM_SHAPES = [
    # (M, N, K)  -- ragged on purpose, plus split-K shapes (skinny M,N with long K)
    (8, 16, 64), (5, 31, 30), (129, 130, 33), (512, 384, 2000), (96, 257, 4101), (300, 40, 7), (1, 1, 1),
    # K % 32 == 0 and 16-byte rows: the LDS-DMA pipeline (ragged tiles, split-K slabs, single pass)
    (132, 260, 4096), (130, 262, 4096), (640, 1280, 256), (1024, 2048, 128), (512, 384, 2048),
    # K % 4 == 0 but not % 32 (30 000 genes): the same pipeline with a zero-filled last K step, with and without split-K
    (260, 136, 4100), (256, 640, 3004), (1024, 1024, 36),
]


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", GEMM_SHAPES)
def test_gemm_fp32(ops, layout, shape):
    M, N, K = shape
    a64 = _rand((M, K), 1)
    b64 = _rand((K, N), 2)
    bias64 = _rand((N,), 3)
    ref = a64 @ b64 + bias64
    A = (a64 if layout != ops.TN else a64.t().contiguous()).float().cuda()
    Bm = (b64.t().contiguous() if layout == ops.NT else b64).float().cuda()
    out = ops.gemm(layout, A, Bm, bias=bias64.float().cuda())
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 2e-6 * max(1, np.sqrt(K) / 8)
    # accumulate=True adds on top
    out2 = ops.gemm(layout, A, Bm, out=out.clone(), accumulate=True)
    assert rel_err(out2.cpu().numpy(), (2 * ref - bias64).numpy()) < 4e-6 * max(1, np.sqrt(K) / 8)


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(132, 260, 4096), (640, 1280, 256), (512, 384, 2048), (96, 257, 4101)])
def test_gemm_split_precision_is_fp32_grade(ops, layout, shape):
    """precision 2 (and the default fp32 route for aligned shapes): operands split exactly into three
    bf16 terms, six bf16 MFMA products -- must be as accurate as an fp32 GEMM, on operands with a
    wide dynamic range (rows scaled by 2^-20 .. 2^20)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(31)
    scale = torch.exp2(torch.randint(-20, 21, (M, 1), generator=g).double())
    a64 = _rand((M, K), 6) * scale
    b64 = _rand((K, N), 7, 0.05)
    A = (a64 if layout != ops.TN else a64.t().contiguous()).float().cuda()
    Bm = (b64.t().contiguous() if layout == ops.NT else b64).float().cuda()
    ref = (A.double().cpu() if layout != ops.TN else A.double().cpu().t()) @ (Bm.double().cpu().t() if layout == ops.NT else Bm.double().cpu())
    mag = (a64.abs() @ b64.abs()).numpy()
    for prec in (ops.FP32, ops.BF16X3):
        out = ops.gemm(layout, A, Bm, precision=prec).double().cpu().numpy()
        assert np.max(np.abs(out - ref.numpy()) / mag) < 4e-7 * max(1.0, np.sqrt(K) / 16), prec


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(512, 384, 2048), (64, 31, 256), (640, 1280, 256)])
def test_gemm_slabs_sum_to_the_product(ops, layout, shape):
    M, N, K = shape
    a64, b64 = _rand((M, K), 8), _rand((K, N), 9)
    A = (a64 if layout != ops.TN else a64.t().contiguous()).float().cuda()
    Bm = (b64.t().contiguous() if layout == ops.NT else b64).float().cuda()
    slabs = ops.gemm_slabs(layout, A, Bm)
    assert slabs.dim() == 3 and slabs.shape[1:] == (M, N) and slabs.shape[0] >= 1
    assert rel_err(slabs.double().sum(0).cpu().numpy(), (a64 @ b64).numpy()) < 2e-6 * max(1, np.sqrt(K) / 8)


@pytest.mark.parametrize("shape", [(40000, 256, 512), (8192, 256, 2304), (1280, 384, 512), (2048, 256, 512)])
def test_gemm_rowscaled_keeps_every_row_fp32_grade(ops, shape):
    """vae_gemm_rowscaled (the step's decoder-head weight gradient, dW = dlogits^T px): two f16 terms with one power-of-two
    scale per OUTPUT ROW.  Columns of A span 2^-40 .. 1 (a gene's dlogits column follows its softmax mass): every row of
    the product is held to fp64 relative to that row's own sum |a||b| -- a single operand scale (vae_gemm_scaled) would lose
    the quiet rows entirely.  Column maxima arrive per chunk of the K rows (3 chunks; the epilogue backward leaves one per
    cell chunk); 160 x 128, 128 x 160 and 128 x 128 tile shapes."""
    M, N, K = shape
    g = torch.Generator().manual_seed(51)
    col_scale = torch.exp2(torch.randint(-40, 1, (1, M), generator=g).double())
    a64 = _rand((K, M), 18) * col_scale
    b64 = _rand((K, N), 19, 0.7)
    A, Bm = a64.float().cuda(), b64.float().cuda()
    a64, b64 = A.double().cpu(), Bm.double().cpu()          # the operands as the kernel sees them
    ref = (a64.t() @ b64).numpy()
    mag = (a64.abs().t() @ b64.abs()).numpy()
    bounds = [0, K // 3, 2 * K // 3, K]
    col_max = torch.stack([A[bounds[i]:bounds[i + 1]].abs().amax(dim=0) for i in range(3)])
    out = ops.gemm_rowscaled(A, Bm, col_max, Bm.abs().amax().expand(8).contiguous()).double().cpu().numpy()
    err = np.max(np.abs(out - ref) / mag, axis=1)
    assert err.max() < 4e-7 * max(1.0, np.sqrt(K) / 16), (err.max(), int(err.argmax()))
    out1 = ops.gemm_rowscaled(A, Bm).double().cpu().numpy()                      # default maxima: the same product
    assert np.array_equal(out, out1)
    with pytest.raises(ValueError):
        ops.gemm_rowscaled(A[:, :30].contiguous(), Bm)                           # outside the tile pipeline


@pytest.mark.parametrize("counts", ["integer", "fractional", "large"])
@pytest.mark.parametrize("B,G,HE", [(512, 20000, 384), (384, 4096, 256), (2304, 4096, 384)])
def test_count_operand_products(ops, B, G, HE, counts):
    """vae_row_stats + vae_gemm_counts: the encoder's first layer and its weight gradient over the RAW counts,
    (x W^T) / m - sum W and (dpre / m)^T x - colsum(dpre) 1^T (modules.py:264,267).  Integer counts below 2048 run on ONE
    f16 term (flags all zero); fractional counts (mixup, data.py:457-477) or counts from 2048 on must make the kernels
    take the two-term loop -- the same fp32-grade accuracy against fp64 in every case (a one-term product of inexact
    counts would be off by 2^-11)."""
    g = torch.Generator().manual_seed(61)
    x = torch.poisson(torch.rand(B, G, generator=g) * 3.0, generator=g) * (torch.rand(B, G, generator=g) > 0.5)
    x[:, 0] += 1.0
    if counts == "fractional":
        x = x * 0.63 + 0.11 * torch.rand(B, G, generator=g)
    if counts == "large":
        x[3, 5], x[7, 9] = 2049.0, 40001.0         # 12 significant bits; beyond 2^15
    w = _rand((HE, G), 62, 0.05).float()
    dpre = (_rand((B, HE), 63, 1e-3) * torch.exp2(torch.randint(-6, 1, (B, 1), generator=g).double())).float()
    xd, wd, dd = x.cuda(), w.cuda(), dpre.cuda()
    mean, amax, flags, wsum = ops.row_stats(xd, wd)
    assert torch.equal(mean.cpu(), x.sum(dim=1) / G) or rel_err(mean.cpu().numpy(), x.double().mean(dim=1).numpy()) < 1e-6
    assert float(amax.max()) == float(x.abs().max())
    assert bool(flags.max() > 0) == (counts != "integer")
    assert rel_err(wsum.cpu().numpy(), w.double().sum(dim=1).numpy()) < 1e-6
    wmax = wd.abs().amax().expand(4).contiguous()
    # forward: x W^T
    out = ops.gemm_counts(ops.NT, xd, wd, amax, wmax, flags).double().cpu()
    ref, mag = x.double() @ w.double().t(), x.double().abs() @ w.double().abs().t()
    assert float(((out - ref).abs() / mag).max()) < (2e-6 if counts == "large" else 4e-7 * max(1.0, np.sqrt(G) / 16))
    # ... and what the middle makes of it equals the Linear over x / mean - 1
    pre = out / mean.double().cpu()[:, None] - wsum.double().cpu()[None, :]
    xn = x.double() / x.double().mean(dim=1, keepdim=True) - 1.0
    assert rel_err(pre.numpy(), (xn @ w.double().t()).numpy()) < 2e-6
    # weight gradient: (dpre / m)^T x - colsum(dpre) 1^T
    dps = (dd / mean[:, None]).contiguous()
    dmax = dps.abs().amax().expand(4).contiguous()
    colsum = dd.double().sum(dim=0).float()
    gw = ops.gemm_counts(ops.TN, dps, xd, dmax, amax, flags, row_sub=colsum).double().cpu()
    ref_w = dpre.double().t() @ xn
    assert rel_err(gw.numpy(), ref_w.numpy()) < 2e-6
    mag_w = dps.double().cpu().abs().t() @ x.double().abs() + colsum.double().cpu().abs()[:, None]
    ref_s = dps.double().cpu().t() @ x.double() - colsum.double().cpu()[:, None]
    # ("large": one term of 40001 dominates its column's sum from its k step on, and every later fp32 accumulation rounds
    # at that magnitude -- the accumulator's own behaviour, ~sqrt(96 accumulations) x 2^-25; not a property of the operands)
    assert float(((gw - ref_s).abs() / mag_w).max()) < (2e-6 if counts == "large" else 4e-7 * max(1.0, np.sqrt(B) / 16))


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(132, 260, 4096), (640, 1280, 256), (512, 384, 2048), (260, 136, 4100), (96, 257, 4101)])
def test_gemm_two_f16_terms(ops, layout, shape):
    """vae_gemm_scaled: operands scaled by a power of two and split into two f16 terms, three MFMA products -- error
    against fp64 below the fp32 accumulation's own (rows of different magnitude inside f16's reach; ragged shapes take
    the fp32 kernel)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(41)
    scale = torch.exp2(torch.randint(-6, 1, (M, 1), generator=g).double())
    a64 = _rand((M, K), 16) * scale
    b64 = _rand((K, N), 17, 0.05)
    A = (a64 if layout != ops.TN else a64.t().contiguous()).float().cuda()
    Bm = (b64.t().contiguous() if layout == ops.NT else b64).float().cuda()
    bias = _rand((N,), 18).float().cuda()
    ref = (A.double().cpu() if layout != ops.TN else A.double().cpu().t()) @ (Bm.double().cpu().t() if layout == ops.NT else Bm.double().cpu())
    mag = (a64.abs() @ b64.abs()).numpy()
    out = ops.gemm_scaled(layout, A, Bm, bias=bias)
    err = np.abs(out.double().cpu().numpy() - (ref + bias.double().cpu()).numpy()) / (mag + 1.0)
    assert err.max() < 4e-7 * max(1.0, np.sqrt(K) / 16)


@pytest.mark.parametrize("shape", [(5, 4100, 128), (130, 8192, 256), (512, 40000, 256), (300, 6000, 512), (2100, 4100, 256),
                                   (2048, 8192, 128)])
def test_gemm_decoder_head_shape(ops, shape):
    """NT products with a small K and a wide N (the decoder heads), whose time is dominated by the output tile write
    (16-byte stores through the per-wave LDS patch; ragged N takes the scalar path for the last column block): bf16
    mode equals the product of bf16-rounded operands, the two-f16-term mode is fp32-grade; bias, ragged M / N,
    accumulate, a strided output."""
    M, N, K = shape
    a64, b64, bias64 = _rand((M, K), 21), _rand((N, K), 22, 0.1), _rand((N,), 23)
    A, Bm, bias = a64.float().cuda(), b64.float().cuda(), bias64.float().cuda()
    ref_b = A.bfloat16().double().cpu() @ Bm.bfloat16().double().cpu().t() + bias64
    out = ops.gemm(ops.NT, A, Bm, bias=bias, precision=ops.BF16)
    assert rel_err(out.cpu().numpy(), ref_b.numpy()) < 1e-5
    ref = A.double().cpu() @ Bm.double().cpu().t() + bias64
    mag = (A.double().cpu().abs() @ Bm.double().cpu().abs().t()).numpy() + 1.0
    big = torch.full((M, N + 8), 7.0, device="cuda")
    out2 = ops.gemm_scaled(ops.NT, A, Bm, bias=bias, out=big[:, :N])
    assert np.max(np.abs(out2.double().cpu().numpy() - ref.numpy()) / mag) < 4e-7
    assert float(big[:, N:].min()) == 7.0 and float(big[:, N:].max()) == 7.0      # nothing written past N
    out3 = ops.gemm_scaled(ops.NT, A, Bm, out=out2.clone(), accumulate=True)
    assert np.max(np.abs(out3.double().cpu().numpy() - (2 * ref - bias64).numpy()) / mag) < 8e-7
    out4 = ops.gemm(ops.NT, A, Bm, out=out.clone(), accumulate=True, precision=ops.BF16)
    assert rel_err(out4.cpu().numpy(), (2 * ref_b - bias64).numpy()) < 2e-5


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("shape", [(130, 200, 1000), (520, 384, 2004)])
def test_gemm_bf16(ops, layout, shape):
    M, N, K = shape
    a64, b64 = _rand((M, K), 4), _rand((K, N), 5)
    ar, br = a64.float().bfloat16().double(), b64.float().bfloat16().double()
    ref = ar @ br
    A = (a64 if layout != ops.TN else a64.t().contiguous()).float().cuda()
    Bm = (b64.t().contiguous() if layout == ops.NT else b64).float().cuda()
    out = ops.gemm(layout, A, Bm, precision=ops.BF16)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-5  # operands rounded identically; fp32 accumulate


@pytest.mark.parametrize("case", [
    # (layout, io, M, N, K): the five gene-sized products of a bf16-mode step, reduced; ragged M / N edges, K tails, split-K
    (0, 1, 520, 384, 4100), (0, 4, 516, 4104, 256), (0, 4, 2304, 4000, 128), (2, 1, 4104, 256, 516), (1, 1, 516, 256, 4104),
    (2, 2, 384, 4100, 520), (0, 1, 384, 388, 1028), (2, 2, 260, 2052, 520),
    # both operands bf16 (the step's bf16 images of weights / small activations), and all three
    (0, 3, 520, 384, 4100), (0, 7, 516, 4104, 256), (1, 3, 516, 256, 4104), (2, 3, 4104, 256, 516), (2, 3, 384, 4100, 520),
    # ... with every extent a multiple of 8: the LDS-DMA kernel (K tails of 8 / 40 / 56, ragged tiles, split-K, all three tile shapes)
    (0, 3, 520, 384, 4104), (2, 3, 4104, 256, 520), (2, 3, 384, 4104, 512), (1, 3, 520, 256, 4152), (0, 7, 2304, 4000, 1032),
    (0, 3, 136, 4104, 4040), (2, 3, 136, 392, 4096),
])
def test_gemm_bf16_storage(ops, case):
    """vae_gemm_bf16_storage: an operand that is ALREADY bf16 in HBM gives the bits of vae_gemm(VAE_BF16) on its fp32
    image (rounding at the write = rounding at the read); a bf16 output is vae_gemm(VAE_BF16)'s fp32 result + bias rounded
    to nearest even -- exact, because both run the same kernel with the same tiles."""
    layout, io, M, N, K = case
    a64, b64, bias64 = _rand((M, K), 51), _rand((K, N), 52, 0.1), _rand((N,), 53)
    A = (a64 if layout != ops.TN else a64.t().contiguous()).float().cuda()
    Bm = (b64.t().contiguous() if layout == ops.NT else b64).float().cuda()
    bias = bias64.float().cuda()
    if io in (3, 7):
        A16, B16 = A.bfloat16(), Bm.bfloat16()
        want = ops.gemm(layout, A16.float(), B16.float(), bias=bias, precision=ops.BF16)
        if io == 7:
            want = want.bfloat16()
        got = ops.gemm_bf16_storage(layout, A16, B16, io, bias=bias)
    elif io == ops.IO_A_BF16:
        A16 = A.bfloat16()
        want = ops.gemm(layout, A16.float(), Bm, bias=bias, precision=ops.BF16)
        got = ops.gemm_bf16_storage(layout, A16, Bm, io, bias=bias)
        # a strided bf16 operand (rows padded), as the step's workspace holds it
        pad = torch.zeros(A16.shape[0], A16.shape[1] + 8, dtype=torch.bfloat16, device="cuda")
        pad[:, :A16.shape[1]] = A16
        got2 = ops.gemm_bf16_storage(layout, pad[:, :A16.shape[1]], Bm, io, bias=bias)
        assert torch.equal(got2, got)
    elif io == ops.IO_B_BF16:
        B16 = Bm.bfloat16()
        want = ops.gemm(layout, A, B16.float(), bias=bias, precision=ops.BF16)
        got = ops.gemm_bf16_storage(layout, A, B16, io, bias=bias)
    else:
        want = ops.gemm(layout, A, Bm, bias=bias, precision=ops.BF16).bfloat16()
        big = torch.full((M, N + 16), 7.0, dtype=torch.bfloat16, device="cuda")
        got = ops.gemm_bf16_storage(layout, A, Bm, io, bias=bias, out=big[:, :N])
        assert float(big[:, N:].float().min()) == 7.0 and float(big[:, N:].float().max()) == 7.0      # nothing past N
    torch.cuda.synchronize()
    assert got.dtype == want.dtype
    if io & 3 == 3:   # the LDS-DMA kernel cuts split-K slabs at 64-deep tiles: same products, fp32 sums in another order
        assert rel_err(got.float().cpu().numpy(), want.float().cpu().numpy()) < (4e-3 if io & 4 else 2e-6)
    else:
        assert torch.equal(got, want)
    ref = A.bfloat16().double().cpu() if layout != ops.TN else A.bfloat16().double().cpu().t()
    ref = ref @ (Bm.bfloat16().double().cpu().t() if layout == ops.NT else Bm.bfloat16().double().cpu()) + bias64
    assert rel_err(got.float().cpu().numpy(), ref.numpy()) < (4e-3 if io & ops.IO_C_BF16 else 1e-5)


def test_gemm_bf16_storage_refuses_what_it_cannot_run(ops):
    a = torch.zeros(32, 64, device="cuda")
    with pytest.raises(ValueError):      # small shapes are outside the tile pipeline
        ops.gemm_bf16_storage(ops.NT, a.bfloat16(), a, ops.IO_A_BF16)
    b = torch.zeros(512, 512, device="cuda")
    with pytest.raises(ValueError):      # B alone as bf16 exists for the wgrad layout only
        ops.gemm_bf16_storage(ops.NT, b, b.bfloat16(), ops.IO_B_BF16)
    with pytest.raises(TypeError):
        ops.gemm_bf16_storage(ops.NT, b, b, ops.IO_A_BF16)


@pytest.mark.parametrize("layout", [0, 2])
def test_gemm_normalised_counts(ops, layout):
    """x/mean - 1 applied on the fly to the count operand (modules.py:264)."""
    Bc, G, H = 37, 1003, 24
    g = torch.Generator().manual_seed(7)
    x = torch.poisson(torch.rand(Bc, G, generator=g) * 3, generator=g)
    x[:, 0] += 1
    mean = ops.row_mean(x.cuda())
    assert torch.equal(mean.cpu(), x.mean(dim=-1))
    xn = (x.double() / x.double().mean(-1, keepdim=True)) - 1
    if layout == ops.NT:
        w = _rand((H, G), 8, 0.05)
        ref = xn @ w.t()
        out = ops.gemm(ops.NT, x.cuda(), w.float().cuda(), x_mean=mean)
    else:
        dy = _rand((Bc, H), 9)
        ref = dy.t() @ xn
        out = ops.gemm(ops.TN, dy.float().cuda(), x.cuda(), x_mean=mean)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 5e-6


@pytest.mark.parametrize("B,G", [(37, 1003), (8, 20000), (3, 64)])
def test_normalize_rows_is_bit_exact(ops, B, G):
    """x / mean - 1 with IEEE division, as torch evaluates Encoder.forward's normalisation."""
    g = torch.Generator().manual_seed(5)
    x = torch.poisson(torch.rand(B, G, generator=g) * 3, generator=g)
    x[:, 0] += 1
    mean, xn = ops.normalize_rows(x.cuda())
    ref_mean = x.double().mean(dim=1)
    assert rel_err(mean.cpu().double().numpy(), ref_mean.numpy()) < 1e-6
    m32 = mean.cpu()
    assert torch.equal(xn.cpu(), x / m32[:, None] - 1)


def test_gemm_rejects_bad_arguments(ops):
    a = torch.zeros(4, 8, device="cuda")
    with pytest.raises(ValueError):
        ops.gemm(ops.NT, a, torch.zeros(4, 9, device="cuda"))
    with pytest.raises(RuntimeError):
        ops.gemm(ops.NT, torch.zeros(4, 8), torch.zeros(4, 8))
    with pytest.raises(TypeError):
        ops.gemm(ops.NT, a.double(), a.double())


@pytest.mark.parametrize("B,W", [(7, 12), (64, 128), (33, 256), (5, 1000)])
def test_gelu_ln_fwd_bwd(ops, B, W):
    pre = _rand((B, W), 11, 2.0).requires_grad_(True)
    gamma = (1 + _rand((W,), 12, 0.3)).requires_grad_(True)
    beta = _rand((W,), 13, 0.3).requires_grad_(True)
    dout = _rand((B, W), 14)
    y = torch.nn.functional.layer_norm(torch.nn.functional.gelu(pre), (W,), gamma, beta, 1e-5)
    y.backward(dout)
    out, mean, rstd = ops.gelu_ln_fwd(pre.detach().float().cuda(), gamma.detach().float().cuda(), beta.detach().float().cuda())
    assert rel_err(out.cpu().numpy(), y.detach().numpy()) < 3e-6
    dg = torch.zeros(W, device="cuda"); db = torch.zeros(W, device="cuda"); dbias = torch.zeros(W, device="cuda")
    dpre = ops.gelu_ln_bwd(dout.float().cuda(), pre.detach().float().cuda(), mean, rstd, gamma.detach().float().cuda(), dg, db, dbias)
    assert rel_err(dpre.cpu().numpy(), pre.grad.numpy()) < 1e-5
    assert rel_err(dg.cpu().numpy(), gamma.grad.numpy()) < 1e-5
    assert rel_err(db.cpu().numpy(), beta.grad.numpy()) < 1e-5
    assert rel_err(dbias.cpu().numpy(), pre.grad.sum(0).numpy()) < 1e-5


@pytest.mark.parametrize("B,L,pm,pv", [(9, 4, 0.0, 1.0), (130, 30, 0.0, 1.0), (17, 1, 7.3, 0.21)])
def test_reparam_kl_fwd_bwd(ops, B, L, pm, pv):
    from torch.distributions import Normal, kl_divergence
    heads = _rand((B, L + 1), 21, 3.0)
    heads[0, L] = 25.0  # softplus threshold branch
    heads.requires_grad_(True)
    eps = _rand((B, L), 22, 2.0)
    qm, qv = heads[:, :L], torch.nn.functional.softplus(heads[:, L:]) + 1e-4
    z = qm + eps * qv.sqrt()
    kl = kl_divergence(Normal(qm, qv.sqrt()), Normal(torch.tensor(pm, dtype=torch.float64),
                                                     torch.tensor(pv, dtype=torch.float64).sqrt())).sum(1)
    dz = _rand((B, L), 23)
    gk = _rand((B,), 24)
    (z * dz).sum().backward(retain_graph=True)
    (0.37 * (kl * gk).sum()).backward()
    hq = heads.detach().float().cuda()
    qv_g, z_g, kl_g = ops.reparam_kl_fwd(hq, eps.float().cuda(), pm, pv)
    assert rel_err(qv_g.cpu().numpy(), qv.detach().numpy()[:, 0]) < 2e-6
    assert rel_err(z_g.cpu().numpy(), z.detach().numpy()) < 2e-6
    assert rel_err(kl_g.cpu().numpy(), kl.detach().numpy()) < 5e-6
    dbias = torch.zeros(L + 1, device="cuda")
    dh = ops.reparam_kl_bwd(dz.float().cuda(), gk.float().cuda(), 0.37, hq, eps.float().cuda(), qv_g, pm, pv, dbias=dbias)
    assert rel_err(dh.cpu().numpy(), heads.grad.numpy()) < 1e-5
    assert rel_err(dbias.cpu().numpy(), heads.grad.sum(0).numpy()) < 1e-5


def test_colsum_and_elbo_reduce(ops):
    m = _rand((301, 77), 31)
    out = torch.zeros(77, device="cuda")
    ops.colsum_into(m.float().cuda(), out)
    assert rel_err(out.cpu().numpy(), m.sum(0).numpy()) < 1e-5
    ll, kz, kl_ = -_rand((1000,), 32, 500).abs(), _rand((1000,), 33).abs(), _rand((1000,), 34).abs()
    s = ops.elbo_reduce(ll.float().cuda(), kz.float().cuda(), kl_.float().cuda(), 0.7, 5e-3).cpu().double()
    gl = (-ll).mean() + 0.7 * (kz + kl_).mean()
    ref = torch.stack([(-ll).mean(), kz.mean(), kl_.mean(), gl, 5e-3 * gl])
    assert rel_err(s.numpy(), ref.numpy()) < 1e-6


def _grid_condition_scales(gold):
    """Magnitudes of the terms each quantity is a sum/difference of: an fp32 evaluation of
    the reference formulas (the reference itself included: its own fp32 run sits at <= 2.7
    eps*T on this grid) cannot be closer to the fp64 truth than a few eps32 * these."""
    from scipy.special import gammaln, digamma
    x, mu, th, pi = gold["x"], gold["mu"], gold["theta"], gold["pi"]
    e = 1e-8
    T = (np.abs(gammaln(x + th)) + np.abs(gammaln(th)) + np.abs(gammaln(x + 1)) + np.abs(x * np.log(mu + e))
         + np.abs((x + th) * np.log(th + mu + e)) + np.abs(th * np.log(th + e)) + np.abs(pi) + 1)
    Tm = x / (mu + e) + (x + th) / (th + mu + e) + 1e-3
    Tt = (np.abs(digamma(x + th)) + np.abs(digamma(th)) + np.abs(np.log((th + e) / (th + mu + e))) + 2
          + x / (th + mu + e))
    lr = np.abs(np.log((th + e) / (th + mu + e)))
    Tp = 0.25 * (th * (np.abs(np.log(th + e)) + np.abs(np.log(th + mu + e))) + np.abs(pi)) + 1
    # zero branch of ZINB: sigmoid(pi_theta_log) carries the rounding of theta*log-ratio into dmu / dtheta
    Tm = Tm + Tp * th / (th + mu + e)
    Tt = Tt + Tp * (lr + 2)
    return T, Tm, Tt, Tp


@pytest.mark.parametrize("zinb", [True, False])
def test_likelihood_grid_elementwise(ops, zinb):
    """log_zinb_positive / log_nb_positive and their gradients on the reference-made grid
    (x up to 5e4, mu 1e-8..1e4, theta 1e-3..1e4, softplus-threshold edges at |pi| ~ 20).
    Truth = the reference run in fp64; bound = 6 eps32 x the condition scale of each quantity."""
    gold = load_golden("likelihood_grid.npz")
    n = gold["x"].size
    cols = 512
    rows = n // cols
    assert rows * cols == n

    def dev(a):
        return torch.tensor(a, dtype=torch.float32).view(rows, cols).cuda()

    x, mu, th, pi = dev(gold["x"]), dev(gold["mu"]), dev(gold["theta"]), dev(gold["pi"])
    key = "zinb" if zinb else "nb"
    T, Tm, Tt, Tp = _grid_condition_scales(gold)
    eps32 = 2.0 ** -24
    out = ops.loglik_elem_fwd(zinb, x, mu, th, pi).cpu().numpy().ravel().astype(np.float64)
    assert np.all(np.isfinite(out))
    assert np.max(np.abs(out - gold[f"f64/{key}"]) / (eps32 * T)) < 6.0
    dmu, dth, dpi = ops.loglik_elem_bwd(zinb, torch.ones_like(x), x, mu, th, pi)
    for got, name, sc in ((dmu, "dmu", Tm), (dth, "dtheta", Tt)) + (((dpi, "dpi", Tp),) if zinb else ()):
        g_ = got.cpu().numpy().ravel().astype(np.float64)
        assert np.all(np.isfinite(g_)), name
        assert np.max(np.abs(g_ - gold[f"f64/{key}_{name}"]) / (eps32 * sc)) < 8.0, name


# ------------------------------------------------------------------------------------------------
# fused optimiser step against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW,
# the optimiser the reference configures (task.py:534-553, train.py:54), run on the CPU in fp32.
# ------------------------------------------------------------------------------------------------
def _tiny_arena(likelihood="zinb", n_cov=0):
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    p = O.init_params(77, 12, 10, 6, 3, 1, seed=3)
    if n_cov:   # batch-covariate columns (modules.py:102-107): the arena stores the encoder rows with a padded stride
        g = torch.Generator().manual_seed(4)
        for k in ("z_encoder.encoder.fc_layers.0.weight", "l_encoder.encoder.fc_layers.0.weight", "decoder.px_decoder.fc_layers.0.weight"):
            p[k] = torch.cat((p[k], torch.randn(p[k].shape[0], n_cov, generator=g) * 0.1), dim=1).contiguous()
    return p, ParamArena(p, 1, likelihood, "cuda")


@pytest.mark.parametrize("max_norm,wd,n_cov", [(0.5, 0.1, 0), (0.0, 0.0, 0), (1e9, 0.05, 0), (0.5, 0.1, 5)])
def test_fused_adamw_matches_torch_trajectory(max_norm, wd, n_cov):
    from vae_amd.optim import FusedAdamW, reference_weight_decay
    p0, arena = _tiny_arena(n_cov=n_cov)
    if n_cov:
        assert arena.ld_enc_w == 88 and arena.p("z_encoder.encoder.fc_layers.0.weight").stride() == (88, 1)
    warm = 4.0
    lam = lambda s: (warm ** 0.5) * min((s + 1) ** -0.5, (s + 1) * warm ** -1.5)  # noqa: E731
    skip = [k for k in arena.offsets if k.startswith("cell_decoder")]
    fused = FusedAdamW(arena, lr=1e-2, weight_decay=wd, max_grad_norm=max_norm, lr_lambda=lam, skip=skip)
    # torch side: same parameters, same groups, grad=None for what the arena marks unused / skipped
    names = [k for k in arena.offsets if k not in arena.no_grad and k not in skip]
    tp = {k: p0[k].clone().float().requires_grad_(True) for k in names}
    groups = {}
    for k in names:
        groups.setdefault(reference_weight_decay(k, wd), []).append(tp[k])
    opt = torch.optim.AdamW([{"params": v, "weight_decay": w} for w, v in groups.items()], lr=1e-2, betas=(0.9, 0.998), eps=1e-7)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lam)
    before_unused = {k: arena.p(k).clone() for k in arena.offsets if k not in names}
    g = torch.Generator().manual_seed(11)
    for it in range(4):
        for k in names:
            gr = torch.randn(tp[k].shape, generator=g) * (10.0 if it == 1 else 0.01)   # step 1 gets clipped hard
            tp[k].grad = gr.clone()
            arena.g(k).copy_(gr)
        if max_norm > 0:
            total = torch.nn.utils.clip_grad_norm_(list(tp.values()), max_norm)
        opt.step()
        sched.step()
        fused.step()
        torch.cuda.synchronize()
        if max_norm > 0:
            assert abs(float(fused.grad_norm) - float(total)) <= 2e-6 * float(total)
        for k in names:
            assert rel_err(arena.p(k).cpu().numpy(), tp[k].detach().numpy()) < 2e-6, (it, k)
    for k, v in before_unused.items():   # parameters without a gradient are never touched
        assert torch.equal(arena.p(k), v), k
    if n_cov:   # the padding columns of the strided rows stay zero through clipping, decay and the update
        assert float(arena.flat[:18 * 88].view(18, 88)[:, 77 + n_cov:].abs().max()) == 0.0


def test_fused_adamw_rejects_bad_arguments():
    from vae_amd.optim import FusedAdamW
    _, arena = _tiny_arena()
    opt = FusedAdamW(arena)
    opt.steps = -1          # -> step 0 reaches the library
    with pytest.raises(RuntimeError, match="step counts from 1"):
        opt.step()


def test_normalize_rows_uint8_equals_float(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.poisson(torch.rand(33, 2000, generator=g) * 3, generator=g).clamp_(max=255)
    x[:, 0] += 1
    m_f, xn_f = ops.normalize_rows(x.cuda())
    m_u, xn_u = ops.normalize_rows(x.to(torch.uint8).cuda())
    assert torch.equal(m_f, m_u) and torch.equal(xn_f, xn_u)


@pytest.mark.parametrize("G", [1003, 2048])
def test_px_rate_from_saved_row_statistics(ops, G):
    """px_rate = exp(library) * softmax(logits) (modules.py:401,404) from the rowmax / rowsum of nbzinb_fwd."""
    B = 19
    g = torch.Generator().manual_seed(31)
    logits = (torch.randn(B, G, generator=g) * 3).cuda()
    library = (torch.randn(B, generator=g) + 6).cuda()
    x = torch.poisson(torch.rand(B, G, generator=g) * 2, generator=g).cuda()
    table = ops.gene_table(torch.zeros(G, device="cuda"))
    _, rowmax, rowsum, _ = ops.nbzinb_fwd(False, logits, None, x, library, table)
    got = ops.px_rate(logits, library, rowmax, rowsum)
    ref = torch.exp(library.double())[:, None] * torch.softmax(logits.double(), dim=-1)
    assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    out = torch.full((B, G + 8), 7.0, device="cuda")[:, :G]        # strided destination, padding untouched
    ops.px_rate(logits, library, rowmax, rowsum, out=out)
    assert torch.equal(out, got)
    with pytest.raises(ValueError):
        ops.px_rate(logits, library[:-1], rowmax, rowsum)


@pytest.mark.parametrize("p", [0.5, 0.1, 0.93])
def test_dropout_stream_matches_host_restatement(ops, p):
    """The dropout masks are the documented Philox4x32-10 stream (include/vae_amd.h): regenerated on the host
    by tests/helpers.py, whose Philox is pinned by the Random123 known-answer vectors (test_oracle_golden)."""
    from helpers import dropout_multiplier
    B, W, seed, step, site = 37, 200, 0x1234567890ABCDEF, 41, 6
    x = torch.ones(B, W, device="cuda")
    ops.dropout_apply(x, p, seed, step, site)
    ref = dropout_multiplier(p, seed, step, site, B, W)
    assert np.array_equal(x.cpu().numpy(), ref)
    assert abs((ref > 0).mean() - (1 - p)) < 0.02
    # a strided view: element index = row * width + column of the VIEW; the padding is untouched
    buf = torch.ones(B, W + 8, device="cuda")
    ops.dropout_apply(buf[:, :W], p, seed, step, site)
    assert np.array_equal(buf[:, :W].cpu().numpy(), ref) and bool((buf[:, W:] == 1).all())
    # other step / site / seed -> other masks; p = 0 -> untouched
    y = torch.ones(B, W, device="cuda")
    assert not torch.equal(ops.dropout_apply(y.clone(), p, seed, step + 1, site), x)
    assert not torch.equal(ops.dropout_apply(y.clone(), p, seed, step, site + 4), x)
    assert torch.equal(ops.dropout_apply(y.clone(), 0.0, seed, step, site), y)
    with pytest.raises(RuntimeError):
        ops.dropout_apply(y, 1.0, seed, step, site)


@pytest.mark.parametrize("streaming", [False, True], ids=["per_cell", "streaming"])
@pytest.mark.parametrize("zinb", [True, False])
@pytest.mark.parametrize("G", [1000, 1003, 4100])
@pytest.mark.parametrize("counts", ["f32", "u8"])
def test_fused_epilogue_against_the_oracle_likelihood(ops, zinb, G, counts, streaming):
    """vae_nbzinb_fwd / _bwd alone vs the oracle's softmax -> exp(library) -> log_(zi)nb_positive -> sum
    (modules.py:362,404, distributions.py:22-90, task.py:156-165) in fp64 autograd, on counts that hit every branch:
    zeros, table counts (< 16), Stirling counts (16..255, thousands) and -- float counts only -- fractional ones;
    extreme dropout logits on both sides of the softplus threshold."""
    if counts == "u8" and G % 4:
        pytest.skip("uint8 counts need the vector path")
    _epilogue_against_the_oracle(ops, zinb, 9, G, counts, streaming)


@pytest.mark.parametrize("B", [490, 508])
def test_fused_epilogue_with_evenly_dealt_cell_chunks(ops, B):
    """The same comparison at shapes whose (gene strip x cell chunk) grid would overflow one generation of workgroups, so
    that the kernels deal the cells out evenly instead (ll_grid / chunk_cells in epilogue.hip: the first B % chunks
    chunks take one cell more): 20 992 genes = 21 strips of the forward, 41 of the two-gene backward.  B = 490 -> the
    forward runs 48 chunks of 10 or 11 cells; B = 508 -> the backward runs 31 chunks of 16 or 17.  (The bench shapes
    B = 2048 / 4096 take this path; the step tests' small gene counts do not.)"""
    _epilogue_against_the_oracle(ops, True, B, 20992, "f32", True)


def _epilogue_against_the_oracle(ops, zinb, B, G, counts, streaming):
    from oracle import vae_oracle as O
    g = torch.Generator().manual_seed(77)
    x = torch.poisson(torch.rand(B, G, generator=g) * 3, generator=g)
    x[torch.rand(B, G, generator=g) < 0.4] = 0
    rep = lambda v: torch.tensor(v).repeat((B + 8) // 9)[:B]
    x[:, 5] = rep([0, 1, 15, 16, 17, 100, 255, 40, 3.0])
    if counts == "f32":
        x[:, 6] = rep([2500.0, 0.5, 7.25, 1e-9, 30000.0, 15.5, 16.5, 0.0, 1.0])   # large, fractional, below eps
    ls = torch.randn(B, G, generator=g) * 2
    lp = torch.randn(B, G, generator=g) * 3
    lp[:, 7] = rep([-30.0, 30.0, -21.0, 21.0, -19.0, 19.0, 0.0, -50.0, 50.0])
    lib = torch.randn(B, generator=g) * 0.5 + 6
    px_r = torch.randn(G, generator=g) * 0.7
    leaves = [t.double().requires_grad_(True) for t in (ls, lp, lib, px_r)]
    mu = torch.exp(leaves[2])[:, None] * torch.softmax(leaves[0], dim=-1)
    th = torch.exp(leaves[3])
    ll64 = (O.log_zinb_positive(x.double(), mu, th, leaves[1]) if zinb else O.log_nb_positive(x.double(), mu, th)).sum(-1)
    w = torch.linspace(0.5, 1.5, B, dtype=torch.float64)
    (ll64 * w * -3e-3).sum().backward()

    xd = x.to(torch.uint8).cuda() if counts == "u8" else x.cuda()
    lsd, lpd = ls.cuda(), lp.cuda()
    table = ops.gene_table(px_r.cuda())
    ll, rmax, rsum, S = ops.nbzinb_fwd(zinb, lsd, lpd if zinb else None, xd, lib.cuda(), table, streaming=streaming)
    assert np.max(np.abs(ll.cpu().double().numpy() - ll64.detach().numpy()) / np.abs(ll64.detach().numpy())) < 1e-5
    # softmax statistics: log-sum-exp of every row against fp64
    lse = (rmax.double() + rsum.double().log()).cpu().numpy()
    assert np.max(np.abs(lse - torch.logsumexp(ls.double(), dim=-1).numpy())) < 2e-6
    dls, dlp = torch.empty_like(lsd), torch.empty_like(lpd)
    dlib = torch.empty(B, device="cuda")
    dpxr, dbs, dbp = (torch.zeros(G, device="cuda") for _ in range(3))
    ops.nbzinb_bwd(zinb, lsd, lpd if zinb else None, xd, lib.cuda(), table, rmax, rsum, S, w.float().cuda(), -3e-3, dls,
                   dlp if zinb else None, dlib, dpxr, dbs, dbp if zinb else None)
    assert rel_err(dls.cpu().numpy(), leaves[0].grad.numpy()) < 1e-4
    assert rel_err(dlib.cpu().numpy(), leaves[2].grad.numpy()) < 1e-4
    assert rel_err(dpxr.cpu().numpy(), leaves[3].grad.numpy()) < 1e-4
    assert rel_err(dbs.cpu().numpy(), leaves[0].grad.sum(0).numpy()) < 1e-4
    if zinb:
        assert rel_err(dlp.cpu().numpy(), leaves[1].grad.numpy()) < 1e-4
        assert rel_err(dbp.cpu().numpy(), leaves[1].grad.sum(0).numpy()) < 1e-4


@pytest.mark.parametrize("zinb", [True, False])
@pytest.mark.parametrize("counts", ["f32", "u8"])
def test_fused_epilogue_over_bf16_logits(ops, zinb, counts):
    """vae_nbzinb_fwd_ws_bf16 / vae_nbzinb_bwd_bf16 (what a bf16-mode step runs on the logits its decoder-head product
    stored as bf16): the arithmetic is the fp32 kernels' on the widened logits, so the forward results are THEIR bits and
    every gradient element is their fp32 gradient rounded to bf16 (nearest even); in place, as the step calls it."""
    B, G = 70, 2048 + 1024 + 8      # three gene strips, the last nearly empty; two cell chunks
    g = torch.Generator().manual_seed(78)
    x = torch.poisson(torch.rand(B, G, generator=g) * 3, generator=g)
    x[torch.rand(B, G, generator=g) < 0.4] = 0
    x[:9, 5] = torch.tensor([0, 1, 15, 16, 17, 100, 255, 40, 3.0])
    ls16 = (torch.randn(B, G, generator=g) * 2).bfloat16().cuda()
    lp16 = (torch.randn(B, G, generator=g) * 3).bfloat16().cuda()
    lib = (torch.randn(B, generator=g) * 0.5 + 6).cuda()
    table = ops.gene_table((torch.randn(G, generator=g) * 0.7).cuda())
    xd = x.to(torch.uint8).cuda() if counts == "u8" else x.cuda()
    w = torch.linspace(0.5, 1.5, B).cuda()
    ls32, lp32 = ls16.float(), lp16.float()

    def run(ls, lp):
        ll, rmax, rsum, S = ops.nbzinb_fwd(zinb, ls, lp if zinb else None, xd, lib, table, streaming=True)
        dls, dlp = torch.empty_like(ls), torch.empty_like(lp)
        dlib = torch.empty(B, device="cuda")
        dpxr, dbs, dbp = (torch.full((G,), 7.0 if ls.dtype == torch.bfloat16 else 0.0, device="cuda") for _ in range(3))
        ops.nbzinb_bwd(zinb, ls, lp if zinb else None, xd, lib, table, rmax, rsum, S, w, -3e-3, dls, dlp if zinb else None, dlib,
                       dpxr, dbs, dbp if zinb else None)
        return [t.clone() for t in (ll, rmax, rsum, S, dls, dlp, dlib, dpxr, dbs, dbp)]

    r32, r16 = run(ls32, lp32), run(ls16, lp16)
    for i in (0, 1, 2, 3, 6, 7, 8) + ((9,) if zinb else ()):      # ll, statistics, S_b, dlibrary, d px_r, d bias: the fp32 kernels' bits
        assert torch.equal(r16[i], r32[i]), i
    assert torch.equal(r16[4], r32[4].bfloat16())
    if zinb:
        assert torch.equal(r16[5], r32[5].bfloat16())
    # in place: the gradients overwrite the logits they were computed from
    a, b2 = ls16.clone(), lp16.clone()
    ll, rmax, rsum, S = ops.nbzinb_fwd(zinb, a, b2 if zinb else None, xd, lib, table, streaming=True)
    dpxr, dbs, dbp = (torch.zeros(G, device="cuda") for _ in range(3))
    ops.nbzinb_bwd(zinb, a, b2 if zinb else None, xd, lib, table, rmax, rsum, S, w, -3e-3, a, b2 if zinb else None,
                   torch.empty(B, device="cuda"), dpxr, dbs, dbp if zinb else None)
    assert torch.equal(a, r16[4]) and (not zinb or torch.equal(b2, r16[5])) and torch.equal(dpxr, r16[7])


def test_poisson_log_prob_matches_torch():
    """reconstruction_loss "poisson" (task.py:167-168): vae_amd.distributions.Poisson.log_prob on the HIP kernel
    vs torch.distributions.Poisson in fp64, values and gradient w.r.t. the rate."""
    from vae_amd.distributions import Poisson
    g = torch.Generator().manual_seed(5)
    rate = (torch.rand(13, 301, generator=g) * 20 + 1e-3)
    x = torch.poisson(rate, generator=g)
    x[0, :4] = torch.tensor([0.0, 1.0, 170.0, 3000.0])
    r64 = rate.double().requires_grad_(True)
    ref = torch.distributions.Poisson(r64).log_prob(x.double())
    w = torch.rand(13, 301, generator=g).double()
    (ref * w).sum().backward()
    rd = rate.cuda().requires_grad_(True)
    out = Poisson(rd).log_prob(x.cuda())
    (out * w.float().cuda()).sum().backward()
    assert np.max(np.abs(out.detach().cpu().double().numpy() - ref.detach().numpy()) / (1 + np.abs(ref.detach().numpy()))) < 1e-5   # fp32 rounding of terms ~20x the result
    assert rel_err(rd.grad.cpu().numpy(), r64.grad.numpy()) < 1e-5
    # the recon loss of the task: -log_prob summed over genes
    assert rel_err((-out).sum(-1).detach().cpu().numpy(), (-ref).sum(-1).detach().numpy()) < 1e-5


@pytest.mark.parametrize("zinb", [True, False])
def test_streaming_epilogue_forward_at_full_width(ops, zinb):
    """The streaming forward (vae_nbzinb_fwd_ws: the kernels the fused step runs) at 20 000 and 30 000 genes and a batch
    that fills several cell chunks, against the one-workgroup-per-cell kernel (itself held to the fp64 oracle above and in
    the step tests): per-cell ll to 2e-6 relative, S_b to 1e-5 of its scale, statistics to 1e-6."""
    from oracle import vae_oracle as O
    g = torch.Generator(device="cuda").manual_seed(3)
    for B, G in ((300, 20000), (70, 30000), (5, 50000)):
        x = torch.poisson(torch.rand(B, G, generator=g, device="cuda") * 1.5, generator=g)
        x[torch.rand(B, G, generator=g, device="cuda") < 0.4] = 0
        x[:, 3] = 40.0
        ls = torch.randn(B, G, generator=g, device="cuda") * 2
        lp = torch.randn(B, G, generator=g, device="cuda") * 3
        lib = torch.randn(B, generator=g, device="cuda") * 0.5 + 8
        px_r = torch.randn(G, generator=g, device="cuda") * 0.7
        table = ops.gene_table(px_r)
        b = ops.nbzinb_fwd(zinb, ls, lp if zinb else None, x, lib, table, streaming=True)
        c = ops.nbzinb_fwd(zinb, ls, lp if zinb else None, x.to(torch.uint8), lib, table, streaming=True)
        if G <= 32768:        # the per-cell kernel parks a row in LDS: 32 768 genes at most
            a = ops.nbzinb_fwd(zinb, ls, lp if zinb else None, x, lib, table)
            assert float(((a[0] - b[0]).abs() / a[0].abs()).max()) < 2e-6, (B, G)
            assert float(((a[1] + a[2].log()) - (b[1] + b[2].log())).abs().max()) < 1e-6
            assert float((a[3] - b[3]).abs().max() / a[3].abs().max()) < 1e-5
        else:                 # beyond it: straight against the fp64 oracle
            with pytest.raises(RuntimeError):
                ops.nbzinb_fwd(zinb, ls, lp if zinb else None, x, lib, table)
            mu = torch.exp(lib.double().cpu())[:, None] * torch.softmax(ls.double().cpu(), dim=-1)
            th = torch.exp(px_r.double().cpu())
            ll64 = (O.log_zinb_positive(x.double().cpu(), mu, th, lp.double().cpu()) if zinb
                    else O.log_nb_positive(x.double().cpu(), mu, th)).sum(-1)
            assert float(((b[0].cpu().double() - ll64).abs() / ll64.abs()).max()) < 1e-5
        for u, v in zip(b, c):
            assert torch.equal(u, v)                      # uint8 counts: the same bits
        assert torch.equal(b[0], ops.nbzinb_fwd(zinb, ls, lp if zinb else None, x, lib, table, streaming=True)[0])   # run to run
