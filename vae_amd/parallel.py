"""Data-parallel gradient averaging for the fused step: one process per GPU, RCCL over xGMI
(``torch.distributed`` backend "nccl"; "gloo" on CPU for tests).

Semantics = the reference's DDP (train.py:57): every rank computes the mean loss over ITS
cells, gradients are summed over ranks and divided by the world size.  Parameters the loss
never reaches (px_r_decoder.*, px_dropout_decoder.* under NB) are statically excluded -- the
arena keeps them past ``used_end`` -- which replaces ``find_unused_parameters=True``.

The gradient arena is already one flat buffer, so buckets are plain slices: no flatten /
unflatten copies.  Buckets are reduced on a side stream as soon as the kernels that write
them have been enqueued (the fused step calls ``bucket_ready``), so the big decoder-head
bucket travels while the encoder backward still computes.  xGMI is point-to-point, so few,
large messages are preferred (default: 2-3 buckets of tens of MB).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from .arena import ParamArena


class GradAllReducer:
    def __init__(self, arena: ParamArena, group: Optional["dist.ProcessGroup"] = None, payload: str = "fp32"):
        if payload not in ("fp32", "bf16"):
            raise ValueError("payload must be 'fp32' or 'bf16'")
        self.arena = arena
        self.group = group
        self.payload = payload
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.on_gpu = arena.flat_grad.is_cuda
        # RCCL averages in the collective itself (ReduceOp.AVG); gloo has no AVG: sum, then scale
        self.avg_in_collective = (self.world > 1 and self.on_gpu and payload == "fp32"
                                  and dist.get_backend(group) == "nccl")
        self.comm_stream = torch.cuda.Stream(device=arena.device, priority=-1) if self.on_gpu else None
        self._pending: List[Tuple[torch.Tensor, Optional[torch.Tensor], object]] = []
        # bucket boundaries (element offsets into flat_grad): decoder-head weights | everything else.
        # Under NB the dropout-head weights have no gradient and fall between the two ranges.
        ws, wp = "decoder.px_scale_decoder.0.weight", "decoder.px_dropout_decoder.weight"
        lo = arena.offsets[ws][0]
        hi = lo + arena.p(ws).numel() + (arena.p(wp).numel() if arena.likelihood == "zinb" else 0)
        after = arena.offsets[wp][0] + arena.p(wp).numel()
        self.buckets = {"dec_heads": [(lo, hi)], "rest": [(0, lo), (after, arena.used_end)]}

    # -- called by the step right after the kernels writing a bucket were enqueued -----------
    def bucket_ready(self, tag: str, after_event=None) -> None:
        """Start reducing bucket ``tag``.  ``after_event``: an event already recorded (by the native
        step) at the point of the compute stream where the bucket became final; default = now."""
        if self.world == 1:
            return
        for lo, hi in self.buckets[tag]:
            if hi <= lo:
                continue
            buf = self.arena.flat_grad[lo:hi]
            if self.on_gpu:
                ev = after_event
                if ev is None:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(ev)
                    self._launch(buf)
            else:
                self._launch(buf)

    def _launch(self, buf: torch.Tensor) -> None:
        if self.payload == "bf16":
            wire = buf.to(torch.bfloat16)
            work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((buf, wire, work))
        else:
            op = dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM
            work = dist.all_reduce(buf, op=op, group=self.group, async_op=True)
            self._pending.append((buf, None, work))

    def finish(self) -> None:
        """Wait for every bucket, apply the 1/world averaging; the compute stream then owns the grads."""
        if self.world == 1:
            return
        scale = 1.0 / self.world
        ctx = torch.cuda.stream(self.comm_stream) if self.on_gpu else _Null()
        with ctx:
            for buf, wire, work in self._pending:
                work.wait()
                if wire is not None:
                    buf.copy_(wire)
                if not (self.avg_in_collective and wire is None):
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
        dist.broadcast(arena.flat, src=src, group=group)
