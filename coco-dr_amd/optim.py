"""Optimizers for the two flat parameters of CocoBertModel: AdamW (torch.optim.AdamW semantics; COCO steps it through
the HF Trainer, COCO/trainer.py:66-70) and LAMB as the reference implements it (ANCE/utils/lamb.py; ANCE's default,
ANCE/drivers/run_ann.py:128-133), plus ``clip_grad_norm_`` (run_ann.py:347-352) whose coefficient never leaves the
device.  The pass over ``flat_decay`` also refreshes the bf16 weight shadow the GEMMs read, so no separate cast runs."""
from __future__ import annotations

import ctypes as C
import os
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


def lamb_plan(offsets: Sequence[int], numel: int, chunk: int = 4096, skip: Sequence[int] = ()) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Chunk table for ``cocodr_lamb_step``: tensor s covers ``[offsets[s], offsets[s+1])`` (the last one runs to
    ``numel``; alignment padding between tensors rides with the tensor in front of it - it holds zeros and receives
    zero gradients, so it changes no norm).  Tensors listed in ``skip`` get no chunks (the one-pass kernel updates them,
    ``lamb_fused_plan``).  Returns (chunk_start i64, chunk_len i32, chunk_seg i32, seg_chunk_begin i32)."""
    offs = list(offsets) + [numel]
    if offs[0] != 0 or any(b <= a for a, b in zip(offs, offs[1:])) or any(o % 4 for o in offs):
        raise ValueError("lamb_plan: offsets must start at 0, increase strictly and be multiples of 4")
    skip = set(skip)
    start, length, seg, seg_begin = [], [], [], [0]
    for s, (a, b) in enumerate(zip(offs, offs[1:])):
        if s not in skip:
            for c in range(a, b, chunk):
                start.append(c)
                length.append(min(chunk, b - c))
                seg.append(s)
        seg_begin.append(len(start))
    return (np.asarray(start, np.int64), np.asarray(length, np.int32), np.asarray(seg, np.int32), np.asarray(seg_begin, np.int32))


#: tensors below this size stay on the two-pass kernels: the one-pass kernel pays one chip-wide rendezvous per tensor
LAMB_FUSED_MIN = 1 << 18


def lamb_fused_plan(offsets: Sequence[int], numel: int, capacity: int, min_len: int = LAMB_FUSED_MIN, workgroups: int = 0,
                    wg_elements: int = 16384):
    """The tensors of a flat parameter that ``cocodr_lamb_step_fused`` updates in one pass: at least ``min_len`` elements and at
    most ``capacity`` (what the persistent grid keeps on chip: ``cocodr_lamb_fused_capacity()`` = ``workgroups`` x ``wg_elements``),
    packed in order into ROUNDS of at most ``workgroups`` workgroups: a tensor needs ceil(len / wg_elements) of them, a round's spare
    workgroups are handed out in proportion (every CU streams), so BERT's 1 M-element Wq / Wk / Wv / Wo share one round and a 4 M-element
    FFN matrix has its own.  Returns (indices into ``offsets``, seg_start i64, seg_len i32, seg_index i32, wg_begin i32, wg_count i32,
    round_first i32 [rounds + 1]) - empty arrays when nothing qualifies."""
    offs = list(offsets) + [numel]
    G = int(workgroups) if workgroups else max(1, capacity // wg_elements)
    idx = [s for s, (a, b) in enumerate(zip(offs, offs[1:])) if min_len <= b - a <= capacity] if capacity > 0 else []
    lens = [offs[s + 1] - offs[s] for s in idx]
    need = [-(-n // wg_elements) for n in lens]
    rounds, cur, used = [], [], 0
    for k, nd in enumerate(need):
        if cur and used + nd > G:
            rounds.append(cur)
            cur, used = [], 0
        cur.append(k)
        used += nd
    if cur:
        rounds.append(cur)
    wg_begin, wg_count, round_first = [0] * len(idx), [0] * len(idx), [0]
    for members in rounds:
        total = sum(need[k] for k in members)
        spare, b0 = G - total, 0
        for i, k in enumerate(members):  # spare workgroups in proportion to the need (the last member takes the rounding remainder)
            extra = spare * need[k] // total if i + 1 < len(members) else G - b0 - need[k]
            wg_begin[k], wg_count[k] = b0, need[k] + extra
            b0 += wg_count[k]
        round_first.append(round_first[-1] + len(members))
    return (idx, np.asarray([offs[s] for s in idx], np.int64), np.asarray(lens, np.int32), np.asarray(idx, np.int32),
            np.asarray(wg_begin, np.int32), np.asarray(wg_count, np.int32), np.asarray(round_first, np.int32))


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

    #: False: every tensor through the two-pass kernels (A/B switch - also COCODR_LAMB_TWO_PASS=1; the tests run both)
    one_pass = os.environ.get("COCODR_LAMB_TWO_PASS") is None

    def _plan(self, p, offs):
        """(two-pass plan or None, one-pass plan or None, keep-alive, workspace, stats, one-pass workspace): the weight matrices
        (>= 2^18 elements, <= what the persistent grid holds in registers) go through ``cocodr_lamb_step_fused`` - 30 instead of
        42 B / parameter -, everything else (vectors, the embedding tables) through ``cocodr_lamb_step``."""
        key = (id(p), bool(self.one_pass))
        if key not in self._plans:
            fused_idx, fplan, fws = [], None, None
            keep = []
            if self.one_pass:
                cap = int(lib().cocodr_lamb_fused_capacity())
                fused_idx, *arrs_f = lamb_fused_plan(offs, p.numel(), cap, workgroups=int(lib().cocodr_lamb_fused_workgroups()),
                                                     wg_elements=int(lib().cocodr_lamb_fused_workgroup_elements())) if cap > 0 else ([],)
                if fused_idx:
                    fdev = [torch.from_numpy(a).to(p.device) for a in arrs_f]
                    keep += fdev
                    fplan = N.LambFusedPlan(*(t.data_ptr() for t in fdev), len(fused_idx), len(arrs_f[-1]) - 1)
                    fws = torch.zeros(int(lib().cocodr_lamb_fused_workspace_floats(len(fused_idx))), dtype=torch.float32, device=p.device)
            arrs = lamb_plan(offs, p.numel(), skip=fused_idx)
            plan = None
            if len(arrs[0]):
                dev = [torch.from_numpy(a).to(p.device) for a in arrs]
                keep += dev
                plan = N.LambPlan(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), len(arrs[0]), len(offs))
            ws = torch.empty(2 * len(arrs[0]) + len(offs), dtype=torch.float32, device=p.device)
            stats = torch.empty((len(offs), 2), dtype=torch.float32, device=p.device)
            self._plans[key] = (plan, fplan, keep, ws, stats, fws, 2 * len(arrs[0]))
        return self._plans[key]

    #: steps between two read-backs of the one-pass kernel's error flag (the first step always checks)
    one_pass_check_every = 64

    def one_pass_error(self) -> bool:
        """True if a workgroup of the one-pass kernel ever gave up waiting for the others (a device that cannot hold its grid -
        CU mask, partition mode, a GPU shared with another process: the tensors concerned skipped that step).  Reads the flags
        back (a device synchronisation): ``step`` calls it on its first step and every ``one_pass_check_every`` steps."""
        bad = False
        for (plan, fplan, _keep, _ws, _stats, fws, _t0) in self._plans.values():
            if fplan is not None:
                i = int(lib().cocodr_lamb_fused_error_index(fplan.nfused))
                bad |= bool(fws[i:i + 1].view(torch.int32).item())
        return bad

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
                plan, fplan, _keep, ws, stats, fws, trust0 = self._plan(p, offs)
                shadow, begin = None, 0
                m = self._owners.get(id(p))
                if m is not None:
                    shadow, begin = m._shadow_target()
                gsd = None if clip is None else clip.data_ptr() + 4
                if plan is not None:
                    check(lib().cocodr_lamb_step(ptr(p), ptr(p.grad), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), ptr(shadow), begin,
                                                 p.numel(), C.byref(plan), float(group["lr"]), b1, b2, group["eps"],
                                                 group["weight_decay"], grad_scale, gsd, ptr(ws), ptr(stats), stream_ptr()), "lamb_step")
                if fplan is not None:
                    check(lib().cocodr_lamb_step_fused(ptr(p), ptr(p.grad), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), ptr(shadow), begin,
                                                       C.byref(fplan), float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                                                       grad_scale, gsd, ptr(fws), ptr(ws[trust0:]), ptr(stats), stream_ptr()),
                          "lamb_step_fused")
                st["weight_norm"], st["adam_norm"] = stats[:, 0], stats[:, 1]
                st["trust_ratio"] = ws[trust0:]
                _after_native_update(p, m, shadow is not None)
        if self.one_pass:
            self._steps_taken = getattr(self, "_steps_taken", 0) + 1
            if (self._steps_taken == 1 or self._steps_taken % self.one_pass_check_every == 0) and self.one_pass_error():
                import warnings
                self.one_pass = False  # (instance attribute: this optimizer stays on the two-pass kernels)
                warnings.warn("FlatLamb: the one-pass LAMB kernel's workgroups were not co-resident on this device (CU mask, partition "
                              "mode or a shared GPU); the weight matrices skipped their update in at most the last "
                              f"{self.one_pass_check_every} steps. Falling back to the two-pass kernels for the rest of the run.",
                              RuntimeWarning, stacklevel=2)
        return loss
