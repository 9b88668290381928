"""AdamW for the two flat parameters of CocoBertModel, one native pass each (torch.optim.AdamW semantics).
The pass over ``flat_decay`` also refreshes the bf16 weight shadow the GEMMs read, so no separate cast runs."""
from __future__ import annotations

import torch

from ._native import check, lib, ptr, stream_ptr

__all__ = ["FlatAdamW"]


class FlatAdamW(torch.optim.Optimizer):
    """``FlatAdamW(model.param_groups(weight_decay), lr=...)`` or ``FlatAdamW.for_model(model, ...)``.

    The latter also keeps the model's bf16 shadow in sync inside the optimizer pass."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._model = None

    @classmethod
    def for_model(cls, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        opt = cls(model.param_groups(weight_decay), lr=lr, betas=betas, eps=eps)
        opt._model = model
        return opt

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
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
                m = self._model
                if m is not None and p is m.flat_decay:
                    m._ensure_shadow()
                    shadow, begin = m._shadow, m.layout.mat_begin
                check(lib().cocodr_adamw_step(ptr(p), ptr(p.grad), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), ptr(shadow), begin,
                                              p.numel(), float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                                              st["step"], grad_scale, stream_ptr()), "adamw_step")
                if shadow is not None:
                    m._shadow_version = p._version  # the shadow already holds the updated weights
        return loss
