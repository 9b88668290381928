"""Optimizers for the two flat parameters of CocoBertModel: AdamW (torch.optim.AdamW semantics; COCO steps it through
the HF Trainer, COCO/trainer.py:66-70) and LAMB as the reference implements it (ANCE/utils/lamb.py; ANCE's default,
ANCE/drivers/run_ann.py:128-133), plus ``clip_grad_norm_`` (run_ann.py:347-352) whose coefficient never leaves the
device.  The pass over ``flat_decay`` also refreshes the bf16 weight shadow the GEMMs read, so no separate cast runs."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Sequence, Tuple

import numpy as np
import torch

from . import _native as N
from ._native import check, lib, ptr, stream_ptr

__all__ = ["FlatAdamW", "FlatLamb", "clip_grad_norm_", "lamb_plan"]


def clip_grad_norm_(parameters: Iterable[torch.Tensor], max_norm: float) -> torch.Tensor:
    """Device-side ``torch.nn.utils.clip_grad_norm_``: returns a fp32 CUDA tensor ``[total_norm, clip_coef]`` and does
    NOT touch the gradients - pass it to ``FlatAdamW.step`` / ``FlatLamb.step`` as ``clip=`` and the optimizer pass
    multiplies the gradient by ``clip_coef`` while it reads it (no extra sweep over the gradients, no host sync)."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads or len(grads) > 8:
        raise ValueError("clip_grad_norm_: expects 1..8 (flat) parameters with gradients")
    for g in grads:
        if g.dtype != torch.float32 or not g.is_cuda or not g.is_contiguous():
            raise ValueError("clip_grad_norm_: gradients must be contiguous fp32 CUDA tensors")
    dev = grads[0].device
    out = torch.empty(2, dtype=torch.float32, device=dev)
    partial = torch.empty(len(grads) * 1024, dtype=torch.float32, device=dev)
    ptrs = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
    ns = (C.c_size_t * len(grads))(*[g.numel() for g in grads])
    check(lib().cocodr_grad_norm_clip(ptrs, ns, len(grads), float(max_norm), ptr(partial), ptr(out), stream_ptr()), "grad_norm_clip")
    return out


def lamb_plan(offsets: Sequence[int], numel: int, chunk: int = 4096) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Chunk table for ``cocodr_lamb_step``: tensor s covers ``[offsets[s], offsets[s+1])`` (the last one runs to
    ``numel``; alignment padding between tensors rides with the tensor in front of it - it holds zeros and receives
    zero gradients, so it changes no norm).  Returns (chunk_start i64, chunk_len i32, chunk_seg i32, seg_chunk_begin i32)."""
    offs = list(offsets) + [numel]
    if offs[0] != 0 or any(b <= a for a, b in zip(offs, offs[1:])) or any(o % 4 for o in offs):
        raise ValueError("lamb_plan: offsets must start at 0, increase strictly and be multiples of 4")
    start, length, seg, seg_begin = [], [], [], [0]
    for s, (a, b) in enumerate(zip(offs, offs[1:])):
        for c in range(a, b, chunk):
            start.append(c)
            length.append(min(chunk, b - c))
            seg.append(s)
        seg_begin.append(len(start))
    return (np.asarray(start, np.int64), np.asarray(length, np.int32), np.asarray(seg, np.int32), np.asarray(seg_begin, np.int32))


def _shadow_owners(model) -> dict:
    """id(flat_decay) -> the module that keeps a bf16 shadow of it (CocoBertModel, CondenserHead), for ``model`` and all
    of its sub-modules."""
    owners = {}
    for m in model.modules():
        if hasattr(m, "_shadow_target") and hasattr(m, "flat_decay"):
            owners[id(m.flat_decay)] = m
    return owners


def _after_native_update(p: torch.Tensor, owner, shadow_written: bool) -> None:
    """The kernels write the parameter through a raw pointer, which torch's version counter does not see; the bf16 shadows
    decide on that counter whether they are stale.  Bump it, and mark the shadow fresh only if this pass wrote it."""
    torch.autograd.graph.increment_version(p)
    if owner is not None and shadow_written:
        if hasattr(owner, "_shadow_mark_fresh"):
            owner._shadow_mark_fresh()
        else:
            owner._shadow_version = p._version


class FlatAdamW(torch.optim.Optimizer):
    """``FlatAdamW(model.param_groups(weight_decay), lr=...)`` or ``FlatAdamW.for_model(model, ...)``.

    ``for_model`` accepts anything with ``param_groups()`` (CocoBertModel, BertDotNLL.bert, CoCondenserForPretraining with
    its Condenser head) and refreshes every bf16 weight shadow inside the optimizer pass; the plain constructor leaves the
    shadows to the next forward (one extra cast pass), never stale."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._owners = {}

    @classmethod
    def for_model(cls, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        opt = cls(model.param_groups(weight_decay), lr=lr, betas=betas, eps=eps)
        opt._owners = _shadow_owners(model)
        return opt

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, clip: torch.Tensor = None):
        """``clip``: the tensor ``clip_grad_norm_`` returned (its second element scales the gradient on the device)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.numel() % 4:
                    raise ValueError("FlatAdamW handles contiguous fp32 CUDA parameters whose size is a multiple of 4")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                shadow, begin = None, 0
                m = self._owners.get(id(p))
                if m is not None:
                    shadow, begin = m._shadow_target()
                check(lib().cocodr_adamw_step(ptr(p), ptr(p.grad), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), ptr(shadow), begin,
                                              p.numel(), float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                                              st["step"], grad_scale, None if clip is None else clip.data_ptr() + 4,
                                              stream_ptr()), "adamw_step")
                _after_native_update(p, m, shadow is not None)  # the shadow already holds the updated weights
        return loss


class FlatLamb(torch.optim.Optimizer):
    """The reference's ``Lamb`` (ANCE/utils/lamb.py) over flat parameters: ``FlatLamb.for_model(model, lr=..., eps=...)``.

    Trust ratios are per HF-named tensor (the reference's ``nn.Parameter`` granularity), located inside the flats through
    the model's layout.  ``state[p]["weight_norm" / "adam_norm"]`` mirror the statistics the reference logs
    (lamb.py:12-22), as device tensors ``[n_tensors]``."""

    def __init__(self, params, segments: List[Sequence[int]], lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._segments = segments  # per parameter (in param_groups order): sorted tensor start offsets
        self._owners = {}
        self._plans = {}

    @classmethod
    def for_model(cls, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0):
        groups = model.param_groups(weight_decay)
        lo = model.layout
        segs = []
        for g in groups:
            for p in g["params"]:
                which = 0 if p is model.flat_decay else 1
                offs = sorted(off for (w, off, _shape) in lo.names.values() if w == which)
                segs.append(offs)
        opt = cls(groups, segs, lr=lr, betas=betas, eps=eps, weight_decay=0.0)
        for g, src in zip(opt.param_groups, groups):
            g["weight_decay"] = src.get("weight_decay", 0.0) if weight_decay else 0.0
        opt._owners = _shadow_owners(model)
        return opt

    def _plan(self, p, offs):
        key = id(p)
        if key not in self._plans:
            arrs = lamb_plan(offs, p.numel())
            dev = [torch.from_numpy(a).to(p.device) for a in arrs]
            plan = N.LambPlan(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), len(arrs[0]), len(offs))
            ws = torch.empty(2 * len(arrs[0]) + len(offs), dtype=torch.float32, device=p.device)
            stats = torch.empty((len(offs), 2), dtype=torch.float32, device=p.device)
            self._plans[key] = (plan, dev, ws, stats)
        return self._plans[key]

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, clip: torch.Tensor = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        k = 0
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                offs = self._segments[k]
                k += 1
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.numel() % 4:
                    raise ValueError("FlatLamb handles contiguous fp32 CUDA parameters whose size is a multiple of 4")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                plan, _keep, ws, stats = self._plan(p, offs)
                shadow, begin = None, 0
                m = self._owners.get(id(p))
                if m is not None:
                    shadow, begin = m._shadow_target()
                check(lib().cocodr_lamb_step(ptr(p), ptr(p.grad), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), ptr(shadow), begin,
                                             p.numel(), C.byref(plan), float(group["lr"]), b1, b2, group["eps"],
                                             group["weight_decay"], grad_scale, None if clip is None else clip.data_ptr() + 4,
                                             ptr(ws), ptr(stats), stream_ptr()), "lamb_step")
                st["weight_norm"], st["adam_norm"] = stats[:, 0], stats[:, 1]
                st["trust_ratio"] = ws[2 * plan.nchunk:]
                _after_native_update(p, m, shadow is not None)
        return loss
