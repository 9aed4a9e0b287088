"""CPU, world_size 2 over gloo: the data-parallel gradient path (vae_amd/parallel.py).

The kernels cannot run here, so each rank fills its gradient arena with the ORACLE's gradients
of its own half batch; after the bucketed all-reduce every rank must hold the gradient of the
mean loss over the union batch (DDP semantics, reference train.py:57), parameters must be
broadcast from rank 0, and unused parameters must be skipped."""
import copy
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, payload, likelihood, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from golden_inputs import E2E_CASES, build_case_inputs
        from oracle import vae_oracle as O
        from vae_amd.arena import ParamArena
        from vae_amd.parallel import GradAllReducer, broadcast_parameters
        torch.set_num_threads(1)
        case = copy.deepcopy(E2E_CASES[1] if likelihood == "zinb" else E2E_CASES[0])
        params, x, eps_z, eps_l, _, _ = build_case_inputs(case)
        p32 = {k: v.float() for k, v in params.items()}
        # rank 1 starts from different parameters: the broadcast must fix that
        start = p32 if rank == 0 else {k: v + 1.0 for k, v in p32.items()}
        arena = ParamArena(start, case["n_layers"], case["likelihood"], "cpu")
        broadcast_parameters(arena)
        assert all(torch.equal(arena.p(k), p32[k]) for k in p32)

        kw = dict(n_layers=case["n_layers"], likelihood=case["likelihood"], l_mean=case["l_mean"], l_var=case["l_var"],
                  kl_weight=case["kl_weight"], gene_loss_coeff=case["gene_loss_coeff"])
        B = x.shape[0]
        half = slice(rank * B // world, (rank + 1) * B // world)
        _, g_local = O.training_step_with_grads(p32, x[half].float(), eps_z[half].float(), eps_l[half].float(), **kw)
        _, g_full = O.training_step_with_grads(p32, x.float(), eps_z.float(), eps_l.float(), **kw)
        for k, g in g_local.items():
            if g is not None:
                arena.g(k).copy_(g)
        sentinel = arena.flat_grad[arena.used_end:].clone().fill_(7.0)
        arena.flat_grad[arena.used_end:] = sentinel
        red = GradAllReducer(arena, payload=payload)
        BUCKET_ORDER = red.bucket_tags()
        spans = sorted(sp for t in BUCKET_ORDER for sp in red.buckets[t] if sp[1] > sp[0])
        covered = sum(hi - lo for lo, hi in spans)      # the buckets tile every gradient the loss reaches, once
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[0][0] == 0 and spans[-1][1] == arena.used_end
        n_grad = sum(arena.g(k).numel() for k, g in g_local.items() if g is not None)
        assert n_grad <= covered <= n_grad + 3 * len(arena.offsets)    # + alignment padding only
        for tag in BUCKET_ORDER:
            red.bucket_ready(tag)
        red.finish()
        tol = 1e-6 if payload == "fp32" else 2e-2
        worst = 0.0
        for k, g in g_full.items():
            if g is None:
                continue
            err = float((arena.g(k) - g).abs().max() / g.abs().max().clamp_min(1e-12))
            worst = max(worst, err)
        assert worst < tol, worst
        assert torch.equal(arena.flat_grad[arena.used_end:], sentinel)   # unused parameters are not touched
        q.put((rank, "ok", worst))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "fail", repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("payload,likelihood", [("fp32", "zinb"), ("bf16", "zinb"), ("fp32", "nb")])
def test_gradient_allreduce_matches_full_batch(payload, likelihood):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + {"fp32": 0, "bf16": 1}[payload] + (7 if likelihood == "nb" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, payload, likelihood, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_encoder_bucket_row_slabs():
    """the (H + Hl, G) encoder weight gradient -- the last, exposed bucket -- cut into row slabs that the
    native backward releases one by one.  Slabs are whole 128-row tiles, at least 8 MB each, contiguous, and tile the
    bucket exactly; small models keep one bucket."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.parallel import GradAllReducer
    G = 16384
    arena = ParamArena(O.init_params(G, 256, 32, 128, 8, 1, seed=1), 1, "zinb", "cpu")
    one = GradAllReducer(arena)
    assert one.enc_slabs == 1 and one.buckets["enc_w0"] == [(0, 384 * G)] and one.bucket_tags() == ["dec_heads", "small", "enc_w0"]
    for want, bounds in ((3, [0, 128, 256, 384]), (2, [0, 256, 384]), (16, [0, 128, 256, 384])):   # 16: 8 MB = 128 rows here
        red = GradAllReducer(arena, enc_slabs=want)
        assert red.enc_row_bounds == bounds and red.enc_slabs == len(bounds) - 1
        tags = [t for t in red.bucket_tags() if t.startswith("enc_w0")]
        assert tags == [f"enc_w0:{i}" for i in range(len(bounds) - 1)] and "enc_w0" not in red.buckets
        spans = [red.buckets[t][0] for t in tags]
        assert spans[0][0] == 0 and spans[-1][1] == one.buckets["enc_w0"][0][1]
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all((hi - lo) * 4 >= (8 << 20) and lo % (128 * G) == 0 for lo, hi in spans)
    small = ParamArena(O.init_params(512, 64, 32, 32, 8, 1, seed=1), 1, "zinb", "cpu")
    assert GradAllReducer(small, enc_slabs=4).enc_slabs == 1


def _seed_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vae_amd.train import rank_seed
        gen = torch.Generator().manual_seed(rank_seed(43))
        idx = torch.randint(0, 100000, (256,), generator=gen)        # the batch draw of train.main
        perm = torch.randperm(1000, generator=torch.Generator().manual_seed(rank_seed(43)))[:64]   # RandomSampler's
        q.put((rank, rank_seed(43), idx.tolist(), perm.tolist()))
    finally:
        dist.destroy_process_group()


def test_ranks_draw_different_cells():
    """every rank seeded 43 drew the SAME batch, so N GPUs averaged N copies of one gradient.  The reference's
    DDP gives each rank its own sampler shard (train.py:57); here the batch generators are seeded 43 + rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 23
    procs = [ctx.Process(target=_seed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == [43, 44]
    assert res[0][2] != res[1][2] and len(set(res[0][2]) & set(res[1][2])) < 16
    assert res[0][3] != res[1][3]
    from vae_amd.train import rank_seed
    assert rank_seed(43) == 43                                    # single process: the reference's seed
