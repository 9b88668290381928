"""CPU restatement of the optimizer half of the ANCE step (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

  * ``clip_grad_norm``  - torch.nn.utils.clip_grad_norm_ as called at ANCE/drivers/run_ann.py:347-352
                          (third-party torch; published definition: total 2-norm over all gradients,
                          coefficient max_norm / (norm + 1e-6) clamped to 1)
  * ``lamb_step``       - ANCE/utils/lamb.py:61-121, statement by statement

Pinned by tests/golden/lamb_steps.npz, produced by the reference's own ``Lamb`` class behind torch's
``clip_grad_norm_`` (tests/golden/make_golden.py::golden_lamb).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

__all__ = ["clip_grad_norm", "lamb_step"]


def clip_grad_norm(grads: Sequence[np.ndarray], max_norm: float) -> Tuple[float, float]:
    """(total_norm, clip_coef); the caller multiplies every gradient by clip_coef."""
    total = float(np.sqrt(sum(float(np.sum(np.asarray(g, np.float64) ** 2)) for g in grads)))
    return total, min(1.0, max_norm / (total + 1e-6))


def lamb_step(params: List[np.ndarray], grads: Sequence[np.ndarray], exp_avg: List[np.ndarray], exp_avg_sq: List[np.ndarray],
              lr: float, betas=(0.9, 0.999), eps: float = 1e-6, weight_decay: float = 0.0) -> np.ndarray:
    """One ``Lamb.step`` over a list of tensors, in place (float64 arithmetic); returns the trust ratios."""
    b1, b2 = betas
    trust = np.ones(len(params))
    for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avg, exp_avg_sq)):
        m *= b1; m += (1 - b1) * g                      # lamb.py:95  (no bias correction, :99-101)
        v *= b2; v += (1 - b2) * g * g                  # lamb.py:97
        weight_norm = min(max(float(np.sqrt(np.sum(p * p))), 0.0), 10.0)  # lamb.py:103  clamp(0, 10)
        adam_step = m / (np.sqrt(v) + eps)              # lamb.py:105
        if weight_decay != 0:
            adam_step = adam_step + weight_decay * p    # lamb.py:106-107
        adam_norm = float(np.sqrt(np.sum(adam_step * adam_step)))  # lamb.py:109
        trust[i] = 1.0 if (weight_norm == 0 or adam_norm == 0) else weight_norm / adam_norm  # lamb.py:110-113
        p -= lr * trust[i] * adam_step                  # lamb.py:120
    return trust
