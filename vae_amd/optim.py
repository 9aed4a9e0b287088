"""Fused optimiser step on the flat arenas: gradient clipping by global norm
(``Trainer(gradient_clip_val=0.5)``, train.py:54) + AdamW with the reference's groups
(``betas=(0.9, 0.998)``, ``eps=1e-7``, weight decay on ``z_encoder`` / ``cell_decoder`` only,
task.py:534-553) in two HIP launches (``vae_adamw_step``), replacing ``clip_grad_norm_`` +
``torch.optim.AdamW.step`` (dozens of multi-tensor launches, seven passes over 18 M parameters).

The learning-rate schedule stays on the host: pass ``lr_lambda`` (e.g.
``WarmupConstantAndDecaySchedule.lr_lambda``, task.py:662-676); ``step()`` evaluates it at the number
of steps already taken, exactly like ``LambdaLR`` stepped once per optimiser step.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch

from . import _lib
from .arena import ParamArena


def reference_weight_decay(name: str, weight_decay: float) -> float:
    """The reference's two groups (task.py:534-553): decay only on z_encoder.* and cell_decoder.*."""
    return weight_decay if ("z_encoder" in name or "cell_decoder" in name) else 0.0


class FusedAdamW:
    def __init__(self, arena: ParamArena, lr: float = 5e-4, betas: Tuple[float, float] = (0.9, 0.998), eps: float = 1e-7,
                 weight_decay: float = 0.0, max_grad_norm: float = 0.5,
                 decay_of: Callable[[str, float], float] = reference_weight_decay,
                 lr_lambda: Optional[Callable[[int], float]] = None, skip: Iterable[str] = (),
                 track_weight_maxima: bool = False, track_bf16_images: bool = False):
        """``skip``: parameter names that receive no gradient in this run (e.g. the cell heads when no
        cell loss is used) in addition to ``arena.no_grad``: torch leaves grad=None parameters alone.

        ``track_weight_maxima``: every step also leaves the partial maxima of the two gene-sized weight matrices
        (first-layer encoder weights, decoder-head weights) in ``self.weight_maxima`` (2, P) while it stores the updated
        weights; ``ElboStep.use_weight_maxima(opt)`` then skips the step's own 72 MB scan for the two-f16-term GEMM
        scales.  Anything else that changes those weights must call ``arena.touch()`` (``broadcast_parameters`` and
        ``LitVAE``'s state-dict load hook do) or ``refresh_weight_maxima()``.

        ``track_bf16_images``: the bf16-mode counterpart -- every step also leaves bf16 images (nearest even) of the same
        two matrices in ``self.weight_images`` while it stores the updated weights; ``ElboStep.use_weight_images(opt)``
        then skips the step's own cast launch (6 bytes per weight per step).  Same rule: ``refresh_weight_images()`` after
        anything else changed the weights.  One of the two options at a time (they belong to different precision modes)."""
        if not arena.flat.is_cuda:
            raise RuntimeError("vae_amd: FusedAdamW needs the arena on a GPU (there is no CPU fallback)")
        self.lib = _lib.load()
        self.arena = arena
        self.base_lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.max_grad_norm = float(max_grad_norm) if max_grad_norm else 0.0
        self.lr_lambda = lr_lambda
        self.steps = 0
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=arena.device)
        self._ws = torch.empty(self.lib.vae_adamw_workspace_bytes() // 4, dtype=torch.float32, device=arena.device)
        # contiguous ranges with one weight decay each, in arena order
        skipped = set(skip) | set(arena.no_grad)
        ranges: List[Tuple[int, int, float]] = []
        for name, (off, shape) in arena.offsets.items():
            if name in skipped:
                continue
            n = arena.extent(name)
            wd = float(decay_of(name, weight_decay))
            if ranges and ranges[-1][1] == off and ranges[-1][2] == wd:
                ranges[-1] = (ranges[-1][0], off + n, wd)
            else:
                ranges.append((off, off + n, wd))
        # alignment padding between groups (<= 3 floats, zero in every arena) is bridged between equal-decay neighbours --
        # but only when the gap really is padding: a skipped / gradient-less parameter of <= 3 elements sitting there
        # must stay untouched, as torch leaves grad=None parameters alone
        def is_padding(lo: int, hi: int) -> bool:
            for name, (off, shape) in arena.offsets.items():
                n = arena.extent(name)
                if off < hi and off + n > lo:
                    return False
            return True

        merged: List[Tuple[int, int, float]] = []
        for r in ranges:
            if merged and r[0] - merged[-1][1] <= 3 and merged[-1][2] == r[2] and is_padding(merged[-1][1], r[0]):
                merged[-1] = (merged[-1][0], r[1], r[2])
            else:
                merged.append(r)
        if len(merged) > 16:
            raise RuntimeError(f"vae_amd: FusedAdamW supports up to 16 ranges, this arena needs {len(merged)}")
        self.ranges = merged
        self._segs = (_lib.AdamWSegment * len(merged))(*[_lib.AdamWSegment(b, e, wd) for b, e, wd in merged])
        self.weight_maxima = None
        self.weight_images = None
        if track_weight_maxima and track_bf16_images:
            raise ValueError("vae_amd: FusedAdamW: track_weight_maxima (two-f16-term mode) or track_bf16_images (bf16 mode), not both")
        if track_weight_maxima or track_bf16_images:
            from .arena import _enc
            def span(first, last):
                b = arena.offsets[first][0]
                e = arena.offsets[last][0] + arena.extent(last)
                return b, e
            enc = span(_enc("z_encoder", 0, "w"), _enc("l_encoder", 0, "w"))
            heads = span("decoder.px_scale_decoder.0.weight",
                         "decoder.px_dropout_decoder.weight" if arena.likelihood == "zinb" else "decoder.px_scale_decoder.0.weight")
            if any(v % 4 for v in enc + heads):
                raise RuntimeError("vae_amd: FusedAdamW(track_weight_maxima / track_bf16_images): the weight matrices are "
                                   "not 16-byte multiples")
            self._max_ranges = (C.c_int64 * 4)(enc[0], enc[1], heads[0], heads[1])
            self._max_spans = (enc, heads)
        if track_weight_maxima:
            self.weight_maxima = torch.zeros(2, int(self.lib.vae_adamw_wmax_partials()), dtype=torch.float32, device=arena.device)
            self.refresh_weight_maxima()
        if track_bf16_images:
            self.weight_images = tuple(torch.empty(e - b, dtype=torch.bfloat16, device=arena.device) for b, e in self._max_spans)
            self.refresh_weight_images()
        self._seen_version = arena.version

    def sync(self) -> None:
        """Refresh the tracked side products when the arena was written by anything but ``step()`` since they were made
        (``ParamArena.touch()``: ``broadcast_parameters`` and ``LitVAE``'s load hook call it; ``ElboStep`` calls this before
        every step that consumes them -- an integer compare)."""
        if self._seen_version != self.arena.version:
            self.refresh_weight_maxima()
            self.refresh_weight_images()
            self._seen_version = self.arena.version

    def refresh_weight_images(self) -> None:
        """Recompute the bf16 images from the weights as they are now (after anything but ``step()`` changed them)."""
        if self.weight_images is None:
            return
        for img, (b, e) in zip(self.weight_images, self._max_spans):
            img.copy_(self.arena.flat[b:e])          # torch rounds to nearest even, as the kernels do

    def refresh_weight_maxima(self) -> None:
        """Recompute the tracked maxima from the weights as they are now (after anything but ``step()`` changed them)."""
        if self.weight_maxima is None:
            return
        self.weight_maxima.zero_()
        for r, (b, e) in enumerate(self._max_spans):
            self.weight_maxima[r, 0] = self.arena.flat[b:e].abs().max()

    @property
    def lr(self) -> float:
        """Learning rate of the NEXT step."""
        return self.base_lr * (float(self.lr_lambda(self.steps)) if self.lr_lambda is not None else 1.0)

    def step(self, grad_scale: float = 1.0) -> None:
        lr = self.lr
        self.steps += 1
        a = self.arena
        if self.weight_maxima is not None:
            rc = self.lib.vae_adamw_step_wmax(a.flat.data_ptr(), a.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), a.total, self._segs, len(self.ranges), lr, self.betas[0],
                                              self.betas[1], self.eps, self.steps, self.max_grad_norm, float(grad_scale),
                                              self._ws.data_ptr(), self._ws.numel() * 4, self.grad_norm.data_ptr(),
                                              self._max_ranges, self.weight_maxima.data_ptr(),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
            _lib.check(rc, "adamw_step_wmax")
            return
        if self.weight_images is not None:
            rc = self.lib.vae_adamw_step_bf16(a.flat.data_ptr(), a.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                              self.exp_avg_sq.data_ptr(), a.total, self._segs, len(self.ranges), lr, self.betas[0],
                                              self.betas[1], self.eps, self.steps, self.max_grad_norm, float(grad_scale),
                                              self._ws.data_ptr(), self._ws.numel() * 4, self.grad_norm.data_ptr(),
                                              self._max_ranges, self.weight_images[0].data_ptr(), self.weight_images[1].data_ptr(),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
            _lib.check(rc, "adamw_step_bf16")
            return
        rc = self.lib.vae_adamw_step(a.flat.data_ptr(), a.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                     self.exp_avg_sq.data_ptr(), a.total, self._segs, len(self.ranges), lr, self.betas[0],
                                     self.betas[1], self.eps, self.steps, self.max_grad_norm, float(grad_scale),
                                     self._ws.data_ptr(), self._ws.numel() * 4, self.grad_norm.data_ptr(),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "adamw_step")

    def state_dict(self) -> Dict[str, object]:
        return {"steps": self.steps, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        self.steps = int(sd["steps"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class L1Penalty:
    """``l1_lambda * sum(|p|)`` over the parameters the reference regularises (task.py:491-496: every
    ``z_encoder`` parameter whose name has no "bias" -- the Linear weights AND the LayerNorm scales), on the
    flat arena: one launch adds ``coeff * sign(p)`` to the gradients and sums ``|p|`` (``vae_l1_penalty``)."""

    def __init__(self, arena: ParamArena, selects: Callable[[str], bool] = lambda n: "z_encoder" in n and "bias" not in n):
        if not arena.flat.is_cuda:
            raise RuntimeError("vae_amd: L1Penalty needs the arena on a GPU (there is no CPU fallback)")
        self.lib = _lib.load()
        self.arena = arena
        ranges: List[Tuple[int, int]] = []
        for name, (off, shape) in arena.offsets.items():
            if name in arena.no_grad or not selects(name):
                continue
            n = arena.extent(name)
            if ranges and ranges[-1][1] == off:
                ranges[-1] = (ranges[-1][0], off + n)
            else:
                ranges.append((off, off + n))
        ranges.sort()
        if not 0 < len(ranges) <= 16:
            raise RuntimeError(f"vae_amd: L1Penalty supports 1..16 ranges, got {len(ranges)}")
        self.ranges = ranges
        self._segs = (_lib.AdamWSegment * len(ranges))(*[_lib.AdamWSegment(b, e, 0.0) for b, e in ranges])
        self._ws = torch.empty(self.lib.vae_adamw_workspace_bytes() // 4, dtype=torch.float32, device=arena.device)
        self.total = torch.zeros(1, dtype=torch.float32, device=arena.device)

    def apply(self, grad_coeff: float, backward: bool = True) -> torch.Tensor:
        """-> sum(|p|) as a 0-dim GPU tensor; gradients += grad_coeff * sign(p) when ``backward``."""
        a = self.arena
        rc = self.lib.vae_l1_penalty(a.flat.data_ptr(), a.flat_grad.data_ptr() if backward else None, a.total, self._segs,
                                     len(self.ranges), float(grad_coeff), self._ws.data_ptr(), self._ws.numel() * 4,
                                     self.total.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "l1_penalty")
        return self.total[0]
