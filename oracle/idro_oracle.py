"""CPU restatement of the iDRO re-weighted ANCE step (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py):
``iDROLoss.forward`` (ANCE/model/dro_loss.py:160-254) driven by ``BertDot_NLL_LN.forward(group_ids=...)``
(ANCE/model/models.py:234-273).  Pinned by tests/golden/idro_steps.npz, two steps of the reference's own classes.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .bert_oracle import OracleConfig, cls_embedding, encoder_bwd, encoder_fwd, layer_names, layers_bwd, triplet_nll

__all__ = ["idro_selected_layers", "idro_weights_update", "idro_step"]


def idro_selected_layers(cfg: OracleConfig, model_size: str = "base") -> List[int]:
    """dro_loss.py:177-181: the last 3 layers of a 12-layer model (names layer.9/10/11), the last 2 of a 24-layer one."""
    return [22, 23] if model_size == "large" else [9, 10, 11]


def idro_weights_update(h_fun: np.ndarray, group_losses: np.ndarray, counts: np.ndarray, all_grads: np.ndarray,
                        alpha: float, eps: float, ema: float, rho: float) -> np.ndarray:
    """dro_loss.py:236-252 - cosine gram of the per-group gradients, scaled by loss^alpha outer products, averaged over
    groups, exponentiated (multiplicative weights), EMA in the log domain, renormalised, floored at eps."""
    mask = (counts > 0).astype(np.float64)
    A = all_grads / (1e-12 + np.linalg.norm(all_grads, axis=-1, keepdims=True))
    RTG = A @ A.T
    gl = np.power(group_losses[:, None], alpha)
    RTG = (gl @ gl.T) * RTG
    ex = rho * RTG.mean(0)
    ex = ex * mask
    ex = ex - ex.max()
    h = np.power(h_fun, ema) * np.exp(ex) * (counts != 0)
    h = h / h.sum()
    return np.maximum(h, eps)


def _sel_vector(G: Dict[str, np.ndarray], layers) -> np.ndarray:
    out = []
    for i in layers:
        for key in layer_names(i).values():
            out.append(np.asarray(G[key], np.float64).ravel())
    return np.concatenate(out)


def idro_step(P, cfg: OracleConfig, batch, groups: np.ndarray, h_fun: np.ndarray, n_groups: int, alpha: float, eps: float,
              ema: float, rho: float, model_size: str = "base") -> Tuple[float, np.ndarray, np.ndarray, np.ndarray, Dict[str, np.ndarray]]:
    """One step: returns (robust_loss, group_losses, counts, new h_fun, gradients of robust_loss w.r.t. every
    parameter).  batch = (q_ids, q_mask, a_ids, a_mask, b_ids, b_mask); three encoder passes (models.py:80-96)."""
    q_ids, q_mask, a_ids, a_mask, b_ids, b_mask = batch
    enc = []
    for ids, mask in ((q_ids, q_mask), (a_ids, a_mask), (b_ids, b_mask)):
        hs, cache = encoder_fwd(P, cfg, ids, mask, keep_cache=True)
        enc.append((hs[-1], cache))
    q, a, b = (cls_embedding(e[0]).astype(np.float64) for e in enc)
    rows, logits = triplet_nll(q, a, b)                       # per-row losses, models.py:97-106
    B = rows.shape[0]
    counts = np.zeros(n_groups); np.add.at(counts, groups, 1.0)                       # dro_loss.py:224-226
    sums = np.zeros(n_groups); np.add.at(sums, groups, rows)
    group_losses = sums / (counts + (counts == 0))                                    # :227
    robust = float((group_losses * h_fun).sum())                                      # :229 (weights BEFORE the update)
    # d(row loss)/d(q, a, b)
    z = logits - logits.max(1, keepdims=True)
    p = np.exp(z) / np.exp(z).sum(1, keepdims=True)
    dl = p.copy(); dl[:, 0] -= 1.0
    dq, da, db = dl[:, :1] * a + dl[:, 1:] * b, dl[:, :1] * q, dl[:, 1:] * q
    sel = idro_selected_layers(cfg, model_size)

    def backward(row_w, only_selected):
        total: Dict[str, np.ndarray] = {}
        for (last, cache), dE in zip(enc, (dq, da, db)):
            d_last = np.zeros_like(last, dtype=np.float64)
            d_last[:, 0] = dE * row_w[:, None]
            if only_selected:  # gradient w.r.t. the selected (top) layers only, :192-205
                G: Dict[str, np.ndarray] = {}
                layers_bwd(P, cfg.num_attention_heads, cache, sel, d_last, G)
            else:
                G = encoder_bwd(P, cfg, cache, d_last)
            for k, v in G.items():
                total[k] = total[k] + v if k in total else np.asarray(v, np.float64).copy()
        return total

    all_grads = []
    for gi in range(n_groups):                                                        # :194-205
        if counts[gi] > 0:
            w = (groups == gi) / counts[gi]
            all_grads.append(_sel_vector(backward(w, True), sel))
        else:
            all_grads.append(None)
    dim = next(g.shape[0] for g in all_grads if g is not None)
    all_grads = np.stack([g if g is not None else np.zeros(dim) for g in all_grads])
    new_h = idro_weights_update(h_fun, group_losses, counts, all_grads, alpha, eps, ema, rho)
    grads = backward(h_fun[groups] / counts[groups], False)                           # d robust / d theta
    return robust, group_losses, counts, new_h, grads
