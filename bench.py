#!/usr/bin/env python3
"""Benchmark of the VAE training hot path (fwd + bwd ELBO) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg1]

One "step" = one fused forward+backward ELBO pass over one batch of synthetic cells that is
already resident in HBM (plus, for N > 1, the gradient all-reduce).  The default workload is
BASELINE.json configs[1]: 20k genes, ZINB, hidden 256, latent 30, batch 512, fp32.
Rank 0 prints ONE JSON line (metric: cells/sec, whole job) with `roofline` for the dominant
kernel (measured with HIP events on the launch stream) and `cpu_baseline` (the oracle's
plain-PyTorch step timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: G, H, Hd, Hl, L, n_layers, likelihood, per-GPU batch, precision, resident cells per GPU
    "cfg1": dict(G=2000, H=128, Hd=128, Hl=128, L=10, n_layers=1, likelihood="nb", B=512, precision="fp32", cells=10_000,
                 desc="10k cells x 2k genes, NB, hidden 128, latent 10"),
    "cfg2": dict(G=20000, H=256, Hd=256, Hl=128, L=30, n_layers=1, likelihood="zinb", B=512, precision="fp32",
                 cells=100_000, desc="100k cells x 20k genes, ZINB, hidden 256, latent 30, batch 512, fp32"),
    "cfg3": dict(G=20000, H=256, Hd=256, Hl=128, L=30, n_layers=1, likelihood="zinb", B=4096, precision="bf16",
                 cells=100_000, desc="20k-gene ZINB, bf16 MFMA GEMMs + fp32 likelihood, batch 4096"),
    "cfg4": dict(G=20000, H=256, Hd=256, Hl=128, L=30, n_layers=2, likelihood="zinb", B=4096, precision="bf16",
                 cells=131_072, desc="20k genes, 2-layer encoder/decoder, batch 4096 per GPU, data parallel"),
    # BASELINE.json configs[4]: the multi-task model (cell-property heads on z, gradient reversal) at 30k genes,
    # widths of the reference's shipped config.py (2 x 512, latent 32, bf16-mixed)
    "cfg5": dict(G=30000, H=512, Hd=512, Hl=128, L=32, n_layers=2, likelihood="zinb", B=2048, precision="bf16",
                 cells=65_536, heads=True, desc="multi-task VAE with cell-property heads, 30k genes, 2 x 512, batch 2048 per GPU"),
}
HEADS = {   # cell properties of the multi-task configuration: two discrete, one adversarial, one continuous
    "Dementia": {"discrete": True, "values": [0, 1], "stop_grad": False},
    "BRAAK_AD": {"discrete": True, "values": [0, 1, 2, 3, 4, 5, 6], "stop_grad": False},
    "Sex": {"discrete": True, "values": [0, 1], "stop_grad": False},
    "Age": {"discrete": False, "values": [-1], "stop_grad": False},
}
HEADS_GRAD_REVERSE = {"Sex": 0.2}
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
# dense TFLOP/s.  fp32 GEMMs run as six bf16 MFMA products per fp32 multiply-add (exact 3-way operand
# split, gemm.hip), so their ceiling in ALGORITHMIC fp32 flops is the bf16 peak / 6 -- above the
# 157.3 TFLOP/s of the fp32 MFMA instruction that the first version of the path used.
MFMA_PEAK = {"fp32": 2500.0 / 3.0, "fp32x2": 2500.0 / 2.0, "fp32x3": 2500.0 / 6.0, "bf16": 2500.0}
MFMA_BASIS = {"fp32x2": "f16 dense 2500 TFLOP/s / 2 products per fp32 multiply-add (integer counts on one f16 term x two-term weights / gradients)",
              "fp32": "f16 dense 2500 TFLOP/s / 3 products per fp32 multiply-add (two scaled f16 terms); the fp32 MFMA peak is 157.3",
              "fp32x3": "bf16 dense 2500 TFLOP/s / 6 products per fp32 multiply-add (three bf16 terms); the fp32 MFMA peak is 157.3",
              "bf16": "bf16 dense 2500 TFLOP/s"}
DTYPE_TEXT = {"fp32": "f32 (the five gene-sized GEMMs: operands scaled by a power of two -- the decoder weight gradient: one per gene -- "
                      "and split into 2 f16 terms, 3 f16 MFMA products, f32 accumulate: error vs f64 below the f32 MFMA's own; "
                      "everything else f32)",
              "fp32x3": "f32 (GEMM operands split exactly into 3 bf16 terms, 6 bf16 MFMA products, f32 accumulate; everything else f32)",
              "bf16": "bf16 GEMM operands, f32 accumulate and likelihood arithmetic; xn, logits and logit gradients stored as bf16 (the reference's autocast storage) where config.bf16_activations says so"}


def synth_counts_device(n, G, device, seed, zero_inflate):
    """SURVEY.md §8(d) synthetic counts generated on the device in chunks: Poisson(lambda_g * s_c),
    optional zero inflation, no all-zero rows.  Stored fp32 (the boundary hands fp32 counts)."""
    g = torch.Generator(device=device).manual_seed(seed)
    lam = torch.exp(torch.randn(G, generator=g, device=device) * 1.5 - 1.5).clamp(0, 50)
    out = torch.empty(n, G, dtype=torch.float32, device=device)
    for i in range(0, n, 8192):
        m = min(8192, n - i)
        s = torch.exp(torch.randn(m, 1, generator=g, device=device) * 0.25)
        x = torch.poisson(lam.unsqueeze(0) * s, generator=g)
        if zero_inflate > 0:
            x = x * (torch.rand(m, G, generator=g, device=device) >= zero_inflate)
        x[:, 0] += (x.sum(dim=1) == 0).float()
        out[i:i + m] = x
    return out


def time_kernel(fn, iters=5):
    """Average device time (ms) of fn() measured with HIP events on the current stream."""
    fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


def kernel_breakdown(arena, cfg, x, precision, bf16_storage=False, counts_path=1):
    """Per-kernel timings of the big stages at the bench shapes, on the launch stream.  bf16_storage: the step of this
    configuration keeps xn / logits / logit gradients as bf16 and streams bf16 images of the weights: the
    same kernels are timed here on such operands, and the likelihood passes are priced at their bytes (8 / 12 per
    (cell, gene) for ZINB instead of 12 / 20)."""
    from vae_amd import ops
    from vae_amd.arena import _enc
    B, G, H, Hl, Hd = x.shape[0], cfg["G"], cfg["H"], cfg["Hl"], cfg["Hd"]
    zinb = cfg["likelihood"] == "zinb"
    dev = x.device
    w_enc = arena.pcat(_enc("z_encoder", 0, "w"), _enc("l_encoder", 0, "w"))
    if zinb:
        w_heads = arena.pcat("decoder.px_scale_decoder.0.weight", "decoder.px_dropout_decoder.weight")
    else:
        w_heads = arena.p("decoder.px_scale_decoder.0.weight")
    NH = w_heads.shape[0]
    mean, xn = ops.normalize_rows(x)  # the step streams x/mean-1 materialised once per batch
    px = torch.randn(B, Hd, device=dev)
    dpre = torch.randn(B, H + Hl, device=dev) * 1e-3
    logits = torch.randn(B, NH, device=dev)
    lib = torch.full((B,), 9.0, device=dev)
    table = ops.gene_table(arena.p("px_r"))
    dl = torch.empty_like(logits)
    dlib, dpxr = torch.empty(B, device=dev), torch.zeros(G, device=dev)
    dbs, dbp = torch.zeros(G, device=dev), torch.zeros(G, device=dev)
    gw_heads, gw_enc = torch.empty_like(w_heads), torch.empty_like(w_enc)
    P = precision
    ks = {}
    fl = lambda m, n, k: 2.0 * m * n * k  # noqa: E731
    if bf16_storage:
        h = lambda t: t.bfloat16()  # noqa: E731
        xn16, w_enc16, px16, w_heads16, dpre16 = h(xn), h(w_enc), h(px), h(w_heads), h(dpre)
        logits = logits.bfloat16()
        dl = torch.empty_like(logits)
        mm = None
        ks["gemm_enc_fwd"] = (time_kernel(lambda: ops.gemm_bf16_storage(ops.NT, xn16, w_enc16, 3)), "mfma", fl(B, H + Hl, G))
        ks["gemm_dec_fwd"] = (time_kernel(lambda: ops.gemm_bf16_storage(ops.NT, px16, w_heads16, 7, out=logits)), "mfma", fl(B, NH, Hd))
        ks["gemm_dec_wgrad"] = (time_kernel(lambda: ops.gemm_bf16_storage(ops.TN, logits, px16, 3, out=gw_heads)), "mfma", fl(NH, Hd, B))
        ks["gemm_dec_dgrad"] = (time_kernel(lambda: ops.gemm_bf16_storage(ops.NN, logits, w_heads16, 3)), "mfma", fl(B, Hd, NH))
        ks["gemm_enc_wgrad"] = (time_kernel(lambda: ops.gemm_bf16_storage(ops.TN, dpre16, xn16, 3, out=gw_enc)), "mfma", fl(H + Hl, G, B))
    elif P == ops.F16X2:   # the step's producers hand the operand maxima to the GEMMs; here: computed once, outside the timing
        sc = {id(t): ops.absmax_scale(t) for t in (xn, w_enc, px, w_heads, logits, dpre)}
        mm = lambda lay, a, b, out=None: ops.gemm_scaled(lay, a, b, sc[id(a)], sc[id(b)], out=out)  # noqa: E731
    else:
        mm = lambda lay, a, b, out=None: ops.gemm(lay, a, b, out=out, precision=P)  # noqa: E731
    if mm is not None and P == ops.F16X2 and not os.environ.get("VAE_NO_XDIRECT"):
        # the two encoder products of this mode stream the RAW counts (vae_gemm_counts): one f16 term for the integer counts
        _, xmax, xflags, _ = ops.row_stats(x, w_enc)
        wmax = w_enc.abs().amax().expand(4).contiguous()
        dps = (dpre / mean[:, None]).contiguous()
        dmax, colsum = dps.abs().amax().expand(4).contiguous(), dpre.sum(dim=0)
        if counts_path != 2:   # (path 2: no statistics launch -- the forward product gathers them itself, step.hip xoptimistic;
            #  the product timed below is then the one-term kernel WITHOUT that gathering: 3-4 us less than in the step)
            ks["row_stats"] = (time_kernel(lambda: ops.row_stats(x, w_enc)), "hbm", 4.0 * (B + H + Hl) * G)
        ks["gemm_enc_fwd"] = (time_kernel(lambda: ops.gemm_counts(ops.NT, x, w_enc, xmax, wmax, xflags)), "mfma", fl(B, H + Hl, G), "fp32x2")
    elif mm is not None:
        ks["gemm_enc_fwd"] = (time_kernel(lambda: mm(ops.NT, xn, w_enc)), "mfma", fl(B, H + Hl, G))
    if mm is not None:
        ks["gemm_dec_fwd"] = (time_kernel(lambda: mm(ops.NT, px, w_heads, logits)), "mfma", fl(B, NH, Hd))
        if P == ops.F16X2:
            sc[id(logits)] = ops.absmax_scale(logits)
        # the decoder wgrad of the two-f16-term mode carries one scale per gene (vae_gemm_rowscaled; step.hip: the per-gene
        # dynamic range of dlogits); the column maxima come from the backward epilogue in the step, here: computed once
        if P == ops.F16X2:
            cmax, bmax = logits.abs().amax(dim=0, keepdim=True).contiguous(), px.abs().amax().expand(4).contiguous()
            wg = lambda: ops.gemm_rowscaled(logits, px, cmax, bmax, out=gw_heads)  # noqa: E731
        else:
            wg = lambda: mm(ops.TN, logits, px, gw_heads)  # noqa: E731
        ks["gemm_dec_wgrad"] = (time_kernel(wg), "mfma", fl(NH, Hd, B))
        if P == ops.F16X2 and cfg["n_layers"] == 1:
            # the fused middle of a one-layer step sums the split-K slabs of this product itself: time the
            # product as the step launches it -- slabs left in the workspace, no reduce launch behind it
            import ctypes as C
            from vae_amd import _lib
            clib = _lib.load()
            nb = clib.vae_gemm_slabs_workspace_bytes(B, Hd, NH)
            wsl = ops.workspace(nb, dev)
            nsl = C.c_int(0)
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            dg = lambda: _lib.check(clib.vae_gemm_scaled_slabs(ops.NN, B, Hd, NH, logits.data_ptr(), logits.stride(0), sc[id(logits)].data_ptr(),  # noqa: E731
                                                              w_heads.data_ptr(), w_heads.stride(0), sc[id(w_heads)].data_ptr(),
                                                              wsl.data_ptr(), nb, C.byref(nsl), st), "gemm_scaled_slabs")
        else:
            dg = lambda: mm(ops.NN, logits, w_heads)  # noqa: E731
        ks["gemm_dec_dgrad"] = (time_kernel(dg), "mfma", fl(B, Hd, NH))
        if ks["gemm_enc_fwd"][-1] == "fp32x2":
            ks["gemm_enc_wgrad"] = (time_kernel(lambda: ops.gemm_counts(ops.TN, dps, x, dmax, xmax, xflags, row_sub=colsum, out=gw_enc)),
                                    "mfma", fl(H + Hl, G, B), "fp32x2")
        else:
            ks["gemm_enc_wgrad"] = (time_kernel(lambda: mm(ops.TN, dpre, xn, gw_enc)), "mfma", fl(H + Hl, G, B))
    logits.normal_()
    lb = 2 if bf16_storage else 4                 # bytes per stored logit
    per = 4 + (2 if zinb else 1) * lb
    # the step's likelihood forward: the streaming pass with the softmax statistics handed in (the decoder-head GEMM's
    # epilogue produces them); the statistics-only pass that serves shapes outside that GEMM path is timed beside it
    fo = torch.stack(ops.nbzinb_fwd(zinb, logits[:, :G], logits[:, G:] if zinb else None, x, lib, table, streaming=True))
    ks["nbzinb_fwd"] = (time_kernel(lambda: ops.nbzinb_fwd(zinb, logits[:, :G], logits[:, G:] if zinb else None, x, lib, table,
                                                           streaming=True, out=fo, have_stats=True, partials_only=True)),
                        "hbm", float(per) * B * G)
    ks["nbzinb_fwd_with_own_statistics"] = (
        time_kernel(lambda: ops.nbzinb_fwd(zinb, logits[:, :G], logits[:, G:] if zinb else None, x, lib, table, streaming=True,
                                           out=fo)), "hbm", float(per + lb) * B * G)
    rmax, rsum, S = fo[1], fo[2], fo[3]
    per_b = 4 + (2 if zinb else 1) * 2 * lb       # counts + logits read, gradients written
    ks["nbzinb_bwd"] = (time_kernel(lambda: ops.nbzinb_bwd(zinb, logits[:, :G], logits[:, G:] if zinb else None, x, lib, table,
                                                           rmax, rsum, S, None, -1e-5, dl[:, :G], dl[:, G:] if zinb else None,
                                                           dlib, dpxr, dbs, dbp if zinb else None)),
                        "hbm", float(per_b) * B * G)
    return ks


def pmc_traffic(config, precision, B, kernel):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json: FETCH_SIZE x 2
    per the gfx950 correction + WRITE_SIZE, separate --pmc runs of this same script's kernels; counters cannot be read
    from inside the timed process).  None when no pass matches this configuration."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    rec = table.get(f"{config}/{precision}/B{B}", {}).get(kernel)
    if not rec:
        return None
    return {"bytes": rec["fetch_bytes"] + rec["write_bytes"], "fetch_bytes": rec["fetch_bytes"], "write_bytes": rec["write_bytes"],
            "unit": "B per launch", "mfma_busy_frac": rec.get("mfma_util"), "source": rec.get("source")}


def cpu_baseline(cfg, threads):
    """The oracle's plain-PyTorch fwd+bwd step on the host cores (bounded: ~10-30 s)."""
    from oracle import vae_oracle as O
    torch.set_num_threads(threads)
    B = 512 if cfg["G"] >= 20000 else 256
    p = O.init_params(cfg["G"], cfg["H"], cfg["Hd"], cfg["Hl"], cfg["L"], cfg["n_layers"], seed=43)
    x = O.synthetic_counts(B, cfg["G"], seed=43, zero_inflate=0.3 if cfg["likelihood"] == "zinb" else 0.0)
    g = torch.Generator().manual_seed(44)
    ez, el = torch.randn(B, cfg["L"], generator=g), torch.randn(B, 1, generator=g)
    lm, lv = O.library_prior(x)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}

    def one():
        for v in leaves.values():
            v.grad = None
        out = O.training_step(leaves, x, ez, el, n_layers=cfg["n_layers"], likelihood=cfg["likelihood"], l_mean=lm, l_var=lv)
        out["loss"].backward()

    for _ in range(2):
        one()
    times = []
    t_end = time.time() + 20.0
    while len(times) < 10 and (time.time() < t_end or len(times) < 3):
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), model)
    except OSError:
        pass
    try:
        import psutil
        physical = psutil.cpu_count(logical=False)
    except Exception:  # noqa: BLE001
        physical = None
    return dict(value=B / med, unit="cells/s", cores=threads, kind="port", host_cpu=model, host_physical_cores=physical,
                host_logical_cpus=os.cpu_count(),
                sample=f"{len(times)} fwd+bwd steps of batch {B} of the same config (oracle/vae_oracle.py, fp32, "
                       f"median {med * 1e3:.1f} ms/step, torch.set_num_threads({threads}): the job's CPU share on the box)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults long enough to leave the load-onset transient of the board behind (the first ~40 steps after
    # any idle gap run 3-10 % slower); still 0.15 s of timed work
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--precision", default="", choices=["", "fp32", "fp32x3", "bf16"],
                    help="fp32 = fp32-grade products on two scaled f16 terms / three MFMA products (default of cfg1/cfg2); "
                         "fp32x3 = three bf16 terms / six products; bf16 = operands rounded to bf16")
    ap.add_argument("--grad-payload", default="auto", choices=["auto", "fp32", "bf16"],
                    help="wire format of the gradient all-reduce (N > 1); auto = bf16 where the step's products already round "
                         "their operands to bf16 (cfg4/cfg5), fp32 for the fp32-grade configurations")
    ap.add_argument("--enc-slabs", type=int, default=-1,
                    help="release the first-layer encoder weight gradient (the last bucket) in up to this many row slabs "
                         "(vae_elbo_step_backward_slabs); default: 3 at N > 1, none at N = 1, where a value > 1 makes an inert reducer and shows what "
                         "slabbing costs (cfg2: nothing measurable, 0.4350 vs 0.4353 ms on one box)")
    ap.add_argument("--counts", default="f32", choices=["f32", "u8"],
                    help="dtype of the resident count matrix: f32 as the reference's loader delivers it (default, the "
                         "headline), or u8 = the reference's on-disk format expanded inside the kernels (row f2)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo only to rehearse the N>1 path on a box with fewer GPUs than ranks")
    ap.add_argument("--scan-weights", action="store_true",
                    help="let every forward scan the two weight matrices for its GEMM operand scales (72 MB, ~15 us; bf16 mode: "
                         "cast them to bf16) instead of taking the maxima / bf16 images the fused optimiser maintains while it "
                         "writes the weights")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched bare: start one rank per GPU ourselves, as fresh child processes of torch.distributed.run, BEFORE
        # anything in this process touches the GPU (importing torch does not); rank 0 of the children prints the JSON line
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    if os.environ.get("VAE_BENCH_WATCHDOG"):  # dump every thread's stack if the run is still alive after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["VAE_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())  # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            # high-priority RCCL stream: the collective's kernels must get CUs while the backward still fills the chip
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                dist.init_process_group("nccl", device_id=dev, pg_options=opts)
            except (AttributeError, TypeError):
                dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from vae_amd import ops
    from vae_amd.arena import ParamArena
    from vae_amd.networks import VAE
    from vae_amd.parallel import GradAllReducer, broadcast_parameters
    from vae_amd.step import ElboStep

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    if args.precision:
        cfg["precision"] = args.precision
    B, G, L = cfg["B"], cfg["G"], cfg["L"]
    precision = {"bf16": ops.BF16, "fp32x3": ops.FP32, "fp32": ops.F16X2}[cfg["precision"]]

    # model: the reference architecture with torch default init under seed 43 (train.py:20)
    torch.manual_seed(43)
    net = VAE(G, n_hidden=cfg["H"], n_hidden_decoder=cfg["Hd"], n_hidden_library=cfg["Hl"], n_latent=L,
              n_latent_cell_decoder=L, n_layers=cfg["n_layers"], dropout_rate=0.0, input_dropout_rate=0.0,
              cell_properties=HEADS if cfg.get("heads") else {"Dementia": HEADS["Dementia"]},
              grad_reverse_dict=HEADS_GRAD_REVERSE if cfg.get("heads") else {}, reconstruction_loss=cfg["likelihood"],
              cell_decoder_hidden_layer=False)
    arena = ParamArena(dict(net.named_parameters()), cfg["n_layers"], cfg["likelihood"], dev)
    cell_decoder = net.cell_decoder.to(dev) if cfg.get("heads") else None
    if cell_decoder is not None:   # the heads' parameters and gradients are arena views too
        for k, q in cell_decoder.named_parameters():
            q.data, q.grad = arena.p("cell_decoder." + k), arena.g("cell_decoder." + k)
    del net
    broadcast_parameters(arena)

    # data: this rank's shard of the synthetic matrix, resident in HBM
    n_cells = max(B, (cfg["cells"] // B) * B)
    n_cells = min(n_cells, 64 * B)  # bounded resident pool: 64 distinct batches per rank are plenty
    x_all = synth_counts_device(n_cells, G, dev, seed=43 + rank, zero_inflate=0.3 if cfg["likelihood"] == "zinb" else 0.0)
    lib = torch.log1p(x_all.sum(dim=1).double())
    l_mean, l_var = float(lib.mean()), float(lib.var(unbiased=False))
    if args.counts == "u8":
        x_all = x_all.clamp_(max=255).to(torch.uint8)
    n_batches = n_cells // B
    gen = torch.Generator(device=dev).manual_seed(43 + rank)
    eps_z = torch.randn(B, L, generator=gen, device=dev)
    eps_l = torch.randn(B, 1, generator=gen, device=dev)

    step = ElboStep(arena, l_mean=l_mean, l_var=l_var, kl_weight=1.0, gene_loss_coeff=5e-3, precision=precision)
    # Training pairs the step with the fused optimiser (vae_amd/optim.py), which leaves the maxima of the two
    # gene-sized weight matrices behind while it stores them; the optimiser step itself is outside the headline (§8d) and
    # the weights do not change in this loop, so the maxima it would maintain are computed once here.
    fopt = None
    if precision == ops.F16X2 and G % 4 == 0 and not args.scan_weights:
        from vae_amd.optim import FusedAdamW
        fopt = FusedAdamW(arena, lr=5e-4, weight_decay=0.1, max_grad_norm=0.5, track_weight_maxima=True)
        step.use_weight_maxima(fopt)
    if precision == ops.BF16 and G % 8 == 0 and not args.scan_weights:   # bf16 mode: the optimiser keeps the bf16 images
        from vae_amd.optim import FusedAdamW
        fopt = FusedAdamW(arena, lr=5e-4, weight_decay=0.1, max_grad_norm=0.5, track_bf16_images=True)
        step.use_weight_images(fopt)
    if args.grad_payload == "auto":
        args.grad_payload = "bf16" if precision == ops.BF16 else "fp32"
    if args.enc_slabs < 0:
        args.enc_slabs = 3 if world > 1 else 0
    reducer = (GradAllReducer(arena, payload=args.grad_payload, enc_slabs=args.enc_slabs)
               if world > 1 or args.enc_slabs > 1 else None)
    head_fn = heads = None
    if cell_decoder is not None:   # CellDecoder + _cell_loss (modules.py:312-324, task.py:419-451) on z
        from vae_amd.step import CellHeads
        from vae_amd.task import cell_loss
        targets = torch.stack([torch.randint(0, len(v["values"]), (B,), generator=gen, device=dev).float() if v["discrete"]
                               else torch.randn(B, generator=gen, device=dev) for v in HEADS.values()], dim=1)
        mask = (torch.rand(B, len(HEADS), generator=gen, device=dev) > 0.1).float()
        if os.environ.get("VAE_BENCH_TORCH_HEADS"):   # A/B: the heads through torch autograd on the (B, L) latent
            head_fn = lambda z: cell_loss(cell_decoder(z), HEADS, targets, mask)  # noqa: E731
        else:                                         # linear heads (the reference's shipped setting): native kernels
            heads = (CellHeads(arena, HEADS, HEADS_GRAD_REVERSE), targets, mask)

    def one_step(i):
        xb = x_all[(i % n_batches) * B:(i % n_batches + 1) * B]
        out = step.run(xb, eps_z, eps_l, cell_head_fn=head_fn, cell_heads=heads, reducer=reducer)
        if reducer is not None:
            reducer.finish()
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # A full collection of the interpreter's garbage collector stalls the host for tens of milliseconds -- more than the
    # launches queued ahead cover -- at a step that moves with every unrelated allocation of the process (measured at cfg5:
    # one 40 ms hole somewhere in steps 50-120).  Collect now, keep the collector off until the timed region ends.
    import gc
    gc.collect()                     # (before the warm-up: nothing but the fence sits between warm-up and timed steps)
    gc_was_on = gc.isenabled()
    gc.disable()
    # The headline takes the GEMM operand scales of the two weight matrices (bf16 mode: their bf16 images) from the fused
    # optimiser, whose step is outside the timed region (ADVICE / VERDICT r2).  The self-contained figure -- every forward
    # scans / casts the weights itself -- is measured by the same K steps right here, BEFORE the headline loop: the board's
    # load-onset transient (the first ~40 steps after an idle gap run 3-10 % slow) then sits in this loop's
    # two untimed steps and early timed ones, and the two figures are compared at the same clock state.  The headline
    # loop is repeated after itself as a stability check (`ms_per_step_headline_repeat`).
    def timed_loop():
        fence()
        t1 = time.perf_counter()
        for i in range(args.steps):
            one_step(args.warmup + i)
        fence()
        return (time.perf_counter() - t1) / args.steps * 1e3
    ms_scan = None
    if fopt is not None and world == 1:
        keep_m, keep_i = step.weight_maxima, step.weight_images
        step.use_weight_maxima(None)
        step.use_weight_images(None)
        for i in range(2):
            one_step(i)
        ms_scan = timed_loop()
        step.use_weight_maxima(fopt if keep_m is not None else None)
        step.use_weight_images(fopt if keep_i is not None else None)
    for i in range(args.warmup):
        one_step(i)
    fence()
    host_trace = [] if os.environ.get("VAE_BENCH_TRACE_HOST") else None   # diagnostic: when the host returned from each step
    gpu_trace = []                                                        # ... and an event behind each step
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = one_step(args.warmup + i)
        if host_trace is not None:
            host_trace.append(time.perf_counter() - t0)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            gpu_trace.append(ev)
    fence()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    if host_trace is not None and rank == 0:
        gaps = sorted(((host_trace[i] - (host_trace[i - 1] if i else 0.0)) * 1e3, i) for i in range(len(host_trace)))[-5:]
        print("host trace: last step returned at %.1f ms of %.1f; longest host steps (ms, index): %s"
              % (host_trace[-1] * 1e3, elapsed * 1e3, [(round(g, 2), i) for g, i in gaps]), file=sys.stderr)
        per = [gpu_trace[i - 1].elapsed_time(gpu_trace[i]) for i in range(1, len(gpu_trace))]
        print("gpu trace (ms between the events behind consecutive steps): " + " ".join("%.3f" % v for v in per), file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    side = None
    if ms_scan is not None:
        ms_again = timed_loop()
        side = {"ms_per_step_scan_weights": ms_scan, "value_scan_weights": B / (ms_scan * 1e-3),
                "ms_per_step_headline_repeat": ms_again, "value_headline_repeat": B / (ms_again * 1e-3),
                "note": "same K steps; scan_weights runs BEFORE the headline loop (every forward scans -- bf16 mode: casts -- the "
                        "two gene-sized weight matrices itself instead of using the optimiser's side products), headline_repeat "
                        "right after it"}
    loss = float(out["loss"])
    bf16_st = step.bf16_activations(B)
    # Row f2 beside the headline: the same K steps fed the reference's ON-DISK format -- uint8 counts (create_dataset.py:131-142)
    u8_side = None
    if args.counts == "f32" and world == 1 and reducer is None and cell_decoder is None and not args.scan_weights:
        x_keep = x_all
        x_all = x_keep.clamp(max=255).to(torch.uint8)
        for i in range(2):
            one_step(i)
        ms_u8 = timed_loop()
        x_all = x_keep
        u8_side = {"ms_per_step": ms_u8, "value": B / (ms_u8 * 1e-3),
                   "note": "the same K steps right after the headline loops, batches handed over as uint8 (counts clamped to 255, as "
                           "the reference's store is); gradients are bit-identical to fp32 counts of the same values"}

    result = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        result = {
            "metric": "training cells/sec (fwd+bwd ELBO, 20k genes)" if G == 20000 else f"training cells/sec (fwd+bwd ELBO, {G} genes)",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_TEXT[cfg["precision"]],
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {cfg['desc']}", "batch_per_gpu": B, "global_batch": B * world,
                       "genes": G, "likelihood": cfg["likelihood"], "n_layers": cfg["n_layers"],
                       "parallelism": f"dp{world}", "grad_payload": args.grad_payload if world > 1 else None,
                       "enc_slabs": reducer.enc_slabs if reducer is not None else None,
                       "counts": args.counts, "bf16_activations": bf16_st,
                       "count_operand_path": ({0: "normalised matrix", 1: "raw counts behind a statistics launch",
                                               2: "raw counts, optimistic (no launch ahead of the first product)"}
                                              .get(step.counts_path(B)) + ("" if args.counts == "f32" else ", read from the uint8 rows" if step.counts_path(B) == 2 and not os.environ.get("VAE_NO_XU8")
                                                 else " over an fp32 copy of the uint8 batch (one cast launch)")),
                       "weight_operand_scales": ("scanned by every forward" if fopt is None else
                                                 "maxima kept by the fused optimiser (vae_adamw_step_wmax)" if fopt.weight_maxima is not None
                                                 else "bf16 images kept by the fused optimiser (vae_adamw_step_bf16)")},
            "final_loss": loss,
        }
        if side is not None:
            result["scan_weights"] = side
        if u8_side is not None:
            result["uint8_counts"] = u8_side
    if rank == 0 and not args.no_breakdown:
        xb = x_all[:B].float()
        ks = kernel_breakdown(arena, cfg, xb, precision, bf16_st, step.counts_path(B))
        # the dominant kernel OF THE STEP (the statistics-only likelihood variant is timed beside it, not launched by this step)
        name, rec = max(((k, v) for k, v in ks.items() if k != "nbzinb_fwd_with_own_statistics"), key=lambda kv: kv[1][0])
        t_ms, bound, work = rec[:3]
        basis = rec[3] if len(rec) > 3 else cfg["precision"]      # the product scheme this kernel runs
        if bound == "hbm":
            ach, peak, unit = work / (t_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            ach, peak, unit = work / (t_ms * 1e-3) / 1e12, MFMA_PEAK[basis], "TFLOP/s"
        result["roofline"] = {"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                              "frac": ach / peak, "traffic": pmc_traffic(args.config, cfg["precision"], B, name), "avg_launch_ms": t_ms,
                              "algorithmic_per_launch": work,
                              "peak_basis": MFMA_BASIS[basis] if bound == "mfma" else "HBM3E 8 TB/s"}
        result["kernels"] = {
            k: {"ms": v[0], "bound": v[1],
                "achieved": (v[2] / (v[0] * 1e-3) / (1e9 if v[1] == "hbm" else 1e12)),
                "unit": "GB/s" if v[1] == "hbm" else "TFLOP/s"} for k, v in ks.items()}
    if rank == 0 and not args.no_breakdown:
        # the optimiser step is excluded from the headline and reported beside it:
        # clip-by-global-norm 0.5 + AdamW over every parameter with a gradient, one fused call
        from vae_amd.optim import FusedAdamW
        if fopt is None:
            fopt = FusedAdamW(arena, lr=5e-4, weight_decay=0.1, max_grad_norm=0.5)
        t_opt = time_kernel(fopt.step, iters=10)
        n_act = sum(e - b for b, e, _ in fopt.ranges)
        result["optimizer"] = {"kernel": "vae_adamw_step (clip 0.5 + AdamW, fused)", "ms": t_opt, "bound": "hbm",
                               "achieved": n_act * 32.0 / (t_opt * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "bytes_per_param": 32, "params": n_act,
                               "cells_per_s_with_optimizer": world * B / ((elapsed / args.steps) + t_opt * 1e-3)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
        result["cpu_baseline"] = cpu_baseline(cfg, threads)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
