"""Data-parallel gradient averaging for the fused step: one process per GPU, RCCL over xGMI
(``torch.distributed`` backend "nccl"; "gloo" on CPU for tests and for rehearsals on one GPU).

Semantics = the reference's DDP (train.py:57): every rank computes the mean loss over ITS
cells, gradients are summed over ranks and divided by the world size.  Parameters the loss
never reaches (px_r_decoder.*, px_dropout_decoder.* under NB) are statically excluded -- the
arena keeps them past ``used_end`` -- which replaces ``find_unused_parameters=True``.

The gradient arena is already one flat buffer, so buckets are plain slices: no flatten /
unflatten copies.  Three buckets, in the order the backward finishes them:

  ``dec_heads``  decoder-head weights (2G x Hd: 41 MB at 20k genes)  -- final after the first wgrad GEMM
  ``small``      every other tensor except the first-layer encoder weights (hidden layers, heads,
                 biases, LayerNorm, px_r, cell heads: < 1 MB)          -- final before the last GEMM
  ``enc_w0``     [z_encoder ; l_encoder] first-layer weights ((H+Hl) x G: 31 MB) -- the last kernel

Each bucket is reduced on a side stream as soon as a HIP event recorded by the native backward
(``vae_elbo_step_backward_events``) says its gradients are final, so the 41 MB bucket travels while the
rest of the backward computes and only the last bucket is exposed.  xGMI is point-to-point, so few,
large messages are preferred over many small ones.

Backends
  * nccl (RCCL): ``all_reduce(AVG)`` of the bucket slice in place, enqueued on the side stream.
  * gloo with GPU gradients (rehearsal on fewer GPUs than ranks): the bucket is copied to a pinned
    host buffer on the side stream, reduced on the host in ``finish()`` and copied back -- gloo never
    sees a device tensor (its own CUDA path takes streams from torch's pool for every collective;
    see DESIGN.md §6 for what that did on one shared GPU).
  * gloo with CPU gradients (unit tests): in place.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .arena import ParamArena, _enc

BUCKET_ORDER = ("dec_heads", "small", "enc_w0")

# ONE communication stream per device and process, shared by every reducer.  Hardware queues are scarce
# (GPU_MAX_HW_QUEUES = 4 on the MI355X boxes) and streams beyond them share queues: with a fresh high-priority stream per
# GradAllReducer, the THIRD reducer of a process (null stream + three side streams + RCCL's own = five streams) left
# hipDeviceSynchronize() waiting forever in its second step although every stream's query() reported completion
# (tests/test_parallel_gpu.py, round 3; independent of buckets, payload and row slabs -- DESIGN 6).
_COMM_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def _comm_stream(device) -> "torch.cuda.Stream":
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _COMM_STREAMS:
        _COMM_STREAMS[idx] = torch.cuda.Stream(device=idx, priority=-1)
    return _COMM_STREAMS[idx]


class GradAllReducer:
    def __init__(self, arena: ParamArena, group: Optional["dist.ProcessGroup"] = None, payload: str = "fp32",
                 force_collectives: bool = False, enc_slabs: int = 0, enc_slab_min_bytes: int = 8 << 20):
        """``force_collectives``: run every collective although the group has ONE rank (they are identities then) --
        lets a one-GPU box execute the whole RCCL path (events, side stream, in-place AVG on arena slices).

        ``enc_slabs``: cut the last bucket -- the (H + Hl, G) first-layer encoder weight gradient, written by the last GEMM
        of the backward -- into up to this many ROW slabs (contiguous arena ranges, whole 128-row tiles, at least
        ``enc_slab_min_bytes`` each: xGMI wants few, large messages); the native backward then produces and releases them one by one (``vae_elbo_step_backward_slabs``), so that
        slab i travels while slab i + 1 is computed.  0 / 1: one bucket, one launch."""
        if payload not in ("fp32", "bf16"):
            raise ValueError("payload must be 'fp32' or 'bf16'")
        self.arena = arena
        self.group = group
        self.payload = payload
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        self.on_gpu = arena.flat_grad.is_cuda
        backend = dist.get_backend(group) if dist.is_initialized() else None
        self.host_staged = self.on_gpu and self.active and backend != "nccl"
        # RCCL averages in the collective itself (ReduceOp.AVG); gloo has no AVG: sum, then scale
        self.avg_in_collective = self.active and self.on_gpu and backend == "nccl"
        self.comm_stream = _comm_stream(arena.device) if self.on_gpu else None
        self._pending: List[tuple] = []
        # bucket boundaries (element offsets into flat_grad).  Under NB the dropout-head weights have no
        # gradient and fall between "dec_heads" and "small".
        ws, wp = "decoder.px_scale_decoder.0.weight", "decoder.px_dropout_decoder.weight"
        lo = arena.offsets[ws][0]
        hi = lo + arena.p(ws).numel() + (arena.p(wp).numel() if arena.likelihood == "zinb" else 0)
        after = arena.offsets[wp][0] + arena.p(wp).numel()
        e0 = arena.offsets[_enc("z_encoder", 0, "w")][0]
        if e0 != 0 or lo < e0:
            raise RuntimeError("GradAllReducer: unexpected arena layout (encoder weights must come first)")
        self.buckets: Dict[str, List[Tuple[int, int]]] = {
            "dec_heads": [(lo, hi)], "small": [(after, arena.used_end)], "enc_w0": [(0, lo)]}
        # row slabs of the encoder bucket: boundaries in rows of the stacked (H + Hl, ld_enc_w) matrix
        ld = int(getattr(arena, "ld_enc_w", 0)) or arena.p(_enc("z_encoder", 0, "w")).shape[1]
        rows = lo // ld if lo % ld == 0 else 0
        self.enc_row_bounds: List[int] = [0, rows]
        if enc_slabs > 1 and rows >= 256:
            per = max(128, -(-int(enc_slab_min_bytes) // (ld * 4) // 128) * 128, -(-rows // enc_slabs // 128) * 128)   # rows per slab
            b = list(range(0, rows, per)) + [rows]
            if len(b) > 2 and b[-1] - b[-2] < 128:      # a short tail joins its neighbour
                del b[-2]
            self.enc_row_bounds = b
            if len(b) > 2:
                del self.buckets["enc_w0"]
                for i in range(len(b) - 1):
                    self.buckets[f"enc_w0:{i}"] = [(b[i] * ld, b[i + 1] * ld)]
        self.enc_slabs = len(self.enc_row_bounds) - 1
        # wire / staging buffers are allocated once (no allocation in the step loop)
        self._wire: Dict[Tuple[int, int], torch.Tensor] = {}
        self._host: Dict[Tuple[int, int], torch.Tensor] = {}
        wire_dtype = torch.bfloat16 if payload == "bf16" else torch.float32
        for spans in self.buckets.values():
            for span in spans:
                n = span[1] - span[0]
                if n <= 0 or not self.active:
                    continue
                if payload == "bf16":
                    self._wire[span] = torch.empty(n, dtype=torch.bfloat16, device=arena.device)
                if self.host_staged:
                    self._host[span] = torch.empty(n, dtype=wire_dtype).pin_memory()

    def bucket_tags(self) -> List[str]:
        """Bucket names in the order the backward finishes them (the encoder bucket possibly as row slabs)."""
        return [t for t in ("dec_heads", "small") if t in self.buckets] + [t for t in self.buckets if t.startswith("enc_w0")]

    # -- called by the step right after the kernels writing a bucket were enqueued -----------
    def bucket_ready(self, tag: str, after_event=None) -> None:
        """Start reducing bucket ``tag``.  ``after_event``: an event already recorded (by the native
        step) at the point of the compute stream where the bucket became final; default = now."""
        if not self.active:
            return
        for span in self.buckets[tag]:
            lo, hi = span
            if hi <= lo:
                continue
            buf = self.arena.flat_grad[lo:hi]
            if not self.on_gpu:
                self._launch_cpu(buf)
                continue
            ev = after_event
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                wire = buf
                if self.payload == "bf16":
                    wire = self._wire[span]
                    wire.copy_(buf)
                if self.host_staged:
                    host = self._host[span]
                    host.copy_(wire, non_blocking=True)
                    landed = torch.cuda.Event()
                    landed.record(self.comm_stream)
                    self._pending.append((buf, wire, host, landed))
                else:
                    op = dist.ReduceOp.AVG if self.payload == "fp32" else dist.ReduceOp.SUM
                    work = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
                    self._pending.append((buf, wire, None, work))

    def _launch_cpu(self, buf: torch.Tensor) -> None:
        wire = buf.to(torch.bfloat16) if self.payload == "bf16" else buf
        work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append((buf, wire, None, work))

    def finish(self) -> None:
        """Wait for every bucket, apply the 1/world averaging; the compute stream then owns the grads."""
        if not self.active:
            return
        scale = 1.0 / self.world
        ctx = torch.cuda.stream(self.comm_stream) if self.on_gpu else _Null()
        with ctx:
            for buf, wire, host, handle in self._pending:
                if host is not None:               # gloo rehearsal on the GPU: host round trip
                    handle.synchronize()           # the device -> host copy has landed
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                    wire.copy_(host, non_blocking=True)
                    if wire is not buf:
                        buf.copy_(wire)
                    buf.mul_(scale)
                    continue
                handle.wait()
                if wire is not buf:
                    buf.copy_(wire)
                if not (self.avg_in_collective and wire is buf):
                    buf.mul_(scale)
        self._pending.clear()
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.comm_stream)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def broadcast_parameters(arena: ParamArena, src: int = 0, group=None) -> None:
    """Make every rank start from rank ``src``'s parameters (one flat broadcast)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if arena.flat.is_cuda and dist.get_backend(group) != "nccl":   # rehearsal: keep gloo away from device tensors
            host = arena.flat.cpu()
            dist.broadcast(host, src=src, group=group)
            arena.flat.copy_(host)
        else:
            dist.broadcast(arena.flat, src=src, group=group)
        arena.touch()   # weight maxima / bf16 images a FusedAdamW keeps are stale now
