"""GPU side of the data-parallel path (vae_amd/parallel.py):
  * one process: the two HIP events the native backward records release each gradient bucket only once it is final
    (a side stream snapshots every bucket behind its event; the snapshots must equal the final gradients);
  * two ranks over RCCL ("nccl"), one GPU each: skipped on a box with one GPU, so the first box with two exercises it;
    every rank must end with the oracle's gradient of the mean loss over the union batch (DDP semantics, train.py:57)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _SnapshotReducer:
    """Stands in for GradAllReducer: 'reducing' a bucket = copying it, on a side stream, behind the event."""
    world, on_gpu = 2, True

    def __init__(self, arena, enc_slabs=0):
        from vae_amd.parallel import GradAllReducer
        self.arena = arena
        real = GradAllReducer(arena, enc_slabs=enc_slabs)
        self.buckets, self.enc_slabs, self.enc_row_bounds = real.buckets, real.enc_slabs, real.enc_row_bounds
        self.stream = torch.cuda.Stream()
        self.snap, self.had_event = {}, {}

    def bucket_ready(self, tag, after_event=None):
        self.had_event[tag] = after_event is not None
        ev = after_event
        if ev is None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            self.snap[tag] = [self.arena.flat_grad[lo:hi].clone() for lo, hi in self.buckets[tag]]


@pytest.mark.parametrize("enc_slabs,precision", [(0, 3), (3, 3), (3, 0), (3, 1)], ids=["one_bucket", "slabs_f16x2", "slabs_bf16x3", "slabs_bf16"])
@pytest.mark.parametrize("n_layers", [1, 2])
def test_native_events_release_buckets_only_when_final(n_layers, enc_slabs, precision):
    """enc_slabs = 3: the encoder weight gradient produced and released in three 128-row slabs
    (vae_elbo_step_backward_slabs), in every precision mode; its values must not depend on the slabbing beyond summation order."""
    from oracle import vae_oracle as O
    from vae_amd.arena import ParamArena
    from vae_amd.step import ElboStep
    B, G, H, Hd, Hl, L = 512, 20000, 256, 256, 128, 30
    p = O.init_params(G, H, Hd, Hl, L, n_layers, seed=3)
    x = O.synthetic_counts(B, G, seed=3, zero_inflate=0.3).cuda()
    g = torch.Generator().manual_seed(4)
    ez, el = torch.randn(B, L, generator=g).cuda(), torch.randn(B, 1, generator=g).cuda()
    arena = ParamArena(p, n_layers, "zinb", "cuda")
    step = ElboStep(arena, l_mean=7.0, l_var=1.0, precision=precision)
    red = _SnapshotReducer(arena, enc_slabs)
    step.run(x, ez, el)
    whole = arena.flat_grad.clone()                  # one launch over all rows
    for it in range(3):
        arena.flat_grad.fill_(float("nan"))      # a bucket released too early would be snapshotted with NaNs in it
        torch.cuda.synchronize()
        step.run(x, ez, el, reducer=red)
        torch.cuda.synchronize()
        if enc_slabs:
            assert red.had_event == {"dec_heads": True, "small": True, "enc_w0:0": True, "enc_w0:1": True, "enc_w0:2": True}
        else:
            assert red.had_event == {"dec_heads": True, "small": True, "enc_w0": False}
        real = torch.zeros(arena.total, dtype=torch.bool, device="cuda")     # gradient elements (not alignment padding)
        for k, (off, shape) in arena.offsets.items():
            if k not in arena.no_grad:
                real[off:off + arena.g(k).numel()] = True
        seen = 0
        for tag, spans in red.buckets.items():
            for (lo, hi), snap in zip(spans, red.snap[tag]):
                m = real[lo:hi]
                assert torch.isfinite(snap[m]).all(), (it, tag)
                assert torch.equal(snap[m], arena.flat_grad[lo:hi][m]), (it, tag)
                seen += int(m.sum())
        assert seen == int(real.sum())                                      # the buckets cover every gradient
        # ... and slabbing changes only the summation order (a 128-row launch may pick another split of the cell axis)
        diff = (arena.flat_grad[real] - whole[real]).abs().max() / whole[real].abs().max()
        assert float(diff) == 0.0 if not enc_slabs else float(diff) < 1e-5


def _nccl_worker(rank, world, port, payload, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from oracle import vae_oracle as O
        from vae_amd.arena import ParamArena
        from vae_amd.parallel import GradAllReducer, broadcast_parameters
        from vae_amd.step import ElboStep
        B, G, H, Hd, Hl, L = 256, 4000, 128, 128, 64, 10
        p = O.init_params(G, H, Hd, Hl, L, 1, seed=11)
        x = O.synthetic_counts(B, G, seed=11, zero_inflate=0.3)
        g = torch.Generator().manual_seed(12)
        ez, el = torch.randn(B, L, generator=g), torch.randn(B, 1, generator=g)
        lm, lv = O.library_prior(x)
        start = p if rank == 0 else {k: v + 1.0 for k, v in p.items()}     # the broadcast must repair rank 1
        arena = ParamArena(start, 1, "zinb", f"cuda:{rank}")
        broadcast_parameters(arena)
        step = ElboStep(arena, l_mean=lm, l_var=lv)
        red = GradAllReducer(arena, payload=payload)
        half = slice(rank * B // world, (rank + 1) * B // world)
        for _ in range(3):                                                   # repeated steps reuse events and wire buffers
            step.run(x[half].cuda(), ez[half].cuda(), el[half].cuda(), reducer=red)
            red.finish()
        torch.cuda.synchronize()
        _, g_full = O.training_step_with_grads({k: v.double() for k, v in p.items()}, x.double(), ez.double(), el.double(),
                                               n_layers=1, likelihood="zinb", l_mean=lm, l_var=lv)
        tol = 1e-4 if payload == "fp32" else 2e-2
        worst = 0.0
        for k, gf in g_full.items():
            if gf is None or k.startswith("cell_decoder"):
                continue
            worst = max(worst, float((arena.g(k).cpu().double() - gf).abs().max() / gf.abs().max().clamp_min(1e-30)))
        q.put((rank, "ok" if worst < tol else "fail", worst))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "fail", repr(e)))
    finally:
        dist.destroy_process_group()


def _rccl_one_rank_worker(port, q):
    """One rank, backend nccl (= RCCL), collectives forced on: every call of the product path executes on the card --
    the native events, the high-priority side stream, in-place all_reduce(AVG) on slices of the gradient arena, the bf16
    wire copy, broadcast of the parameter arena -- and must leave the single rank's own gradients behind."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)      # a stuck collective names its line instead of timing the test out
    import torch.distributed as dist
    torch.cuda.set_device(0)
    try:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)      # as bench.py creates the group
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)
    except (AttributeError, TypeError):
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from oracle import vae_oracle as O
        from vae_amd.arena import ParamArena
        from vae_amd.parallel import GradAllReducer, broadcast_parameters
        from vae_amd.step import ElboStep
        B, G, H, Hd, Hl, L = 512, 4000, 256, 128, 128, 10
        p = O.init_params(G, H, Hd, Hl, L, 1, seed=21)
        x = O.synthetic_counts(B, G, seed=21, zero_inflate=0.3).cuda()
        g = torch.Generator().manual_seed(22)
        ez, el = torch.randn(B, L, generator=g).cuda(), torch.randn(B, 1, generator=g).cuda()
        arena = ParamArena(p, 1, "zinb", "cuda")
        broadcast_parameters(arena)
        dist.broadcast(arena.flat, src=0)                                   # (the call a larger group makes)
        step = ElboStep(arena, l_mean=7.0, l_var=1.0)
        step.run(x, ez, el)
        torch.cuda.synchronize()
        want = arena.flat_grad[:arena.used_end].clone()
        res = {}
        order = (("fp32", 0), ("bf16", 0), ("fp32s", 3), ("bf16s", 3))
        if os.environ.get("VAE_TEST_VARIANT") == "order":   # diagnostic: is it the slabs, or the third reducer's stream?
            order = order[2:] + order[:2]
        for payload, slabs in order:
            red = GradAllReducer(arena, payload=payload[:4], force_collectives=True, enc_slabs=slabs, enc_slab_min_bytes=1 << 20)
            assert red.active and red.avg_in_collective and not red.host_staged and red.enc_slabs == max(1, slabs)
            if os.environ.get("VAE_TEST_VARIANT") == "nonative" and slabs:   # diagnostic: five collectives, no native slab events
                red.enc_slabs = 1
            print("rccl one-rank:", payload, "slabs", red.enc_slabs, red.enc_row_bounds, flush=True)
            for it in range(5):                                             # several steps in flight, no synchronisation
                arena.flat_grad.fill_(float("nan"))
                step.run(x, ez, el, reducer=red)
                red.finish()
                if os.environ.get("VAE_TEST_SYNC_EACH") == "dev":
                    torch.cuda.synchronize()
                    print("  step", it, "done", flush=True)
                elif os.environ.get("VAE_TEST_SYNC_EACH"):   # diagnostic: name what is stuck instead of hanging
                    import time
                    t_end = time.time() + 20
                    while time.time() < t_end and not (torch.cuda.current_stream().query() and red.comm_stream.query()):
                        time.sleep(0.05)
                    evs = {"heads": step._ev_heads, "small": step._ev_small}
                    evs.update({f"slab{i}": e for i, e in enumerate(step._ev_slabs or [])})
                    print("  step", it, "compute done", torch.cuda.current_stream().query(), "comm done", red.comm_stream.query(),
                          {k: e.query() for k, e in evs.items()}, flush=True)
                    if not (torch.cuda.current_stream().query() and red.comm_stream.query()):
                        os._exit(3)
            torch.cuda.synchronize()
            real = torch.zeros(arena.used_end, dtype=torch.bool, device="cuda")     # gradient elements (not alignment padding)
            for k, (off, shape) in arena.offsets.items():
                if k not in arena.no_grad and off < arena.used_end:
                    real[off:off + arena.g(k).numel()] = True
            got = arena.flat_grad[:arena.used_end]
            res[payload] = float((got[real] - want[real]).abs().max() / want[real].abs().max())
        # (row slabs: a 128-row launch may split the cell axis differently -- summation order only)
        ok = res["fp32"] == 0.0 and res["bf16"] < 1e-2 and res["fp32s"] < 1e-5 and res["bf16s"] < 1e-2
        q.put(("ok" if ok else "fail", res))
    except Exception as e:  # noqa: BLE001
        q.put(("fail", repr(e)))
    finally:
        dist.destroy_process_group()


def test_rccl_path_executes_with_one_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_rccl_one_rank_worker, args=(29500 + (os.getpid() % 2000) + 77, q))
    proc.start()
    res = q.get(timeout=240)
    proc.join(timeout=60)
    assert res[0] == "ok", res


@pytest.mark.parametrize("payload", ["fp32", "bf16"])
def test_two_ranks_rccl_match_the_full_batch_gradient(payload):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL, one rank per GPU)")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (41 if payload == "bf16" else 40)
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, payload, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
