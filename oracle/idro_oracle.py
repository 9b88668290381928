"""CPU restatement of the iDRO re-weighted ANCE step (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py):
``iDROLoss.forward`` (ANCE/model/dro_loss.py:160-254) driven by ``BertDot_NLL_LN.forward(group_ids=...)``
(ANCE/model/models.py:234-273), and of ``DROGreedyLoss`` (:11-126, the driver's default strategy).  Pinned by
tests/golden/idro_steps.npz and dro_greedy_steps.npz, steps of the reference's own classes.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .bert_oracle import OracleConfig, cls_embedding, encoder_bwd, encoder_fwd, layer_names, layers_bwd, triplet_nll

__all__ = ["idro_selected_layers", "idro_weights_update", "idro_step", "DROGreedyState", "dro_greedy_update_mw",
           "dro_greedy_forward"]


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


# ------------------------------------------------------------------------------------------------ DRO-greedy
class DROGreedyState:
    """Buffers of ``DROGreedyLoss`` (ANCE/model/dro_loss.py:27-33): h_fun, EMA group losses, EMA group counts."""

    def __init__(self, n_groups: int):
        self.h_fun = np.ones(n_groups)
        self.sum_losses = np.zeros(n_groups)
        self.count_cat = np.ones(n_groups)


def dro_greedy_update_mw(st: DROGreedyState, alpha: float, eps: float, ema: float, weight_ema: bool) -> None:
    """``update_mw`` (dro_loss.py:93-126): groups sorted by EMA loss (descending); the worst ones whose cumulative
    EMA fraction stays below alpha get weight 1/alpha, the next one the left-over mass, the rest eps."""
    past_frac = st.count_cat / st.count_cat.sum()
    sort_id = np.argsort(-st.sum_losses, kind="stable")
    sorted_frac = past_frac[sort_id]
    cutoff = int(np.sum(np.cumsum(sorted_frac) < alpha))
    if cutoff == len(sorted_frac):
        cutoff = len(sorted_frac) - 1
    h = np.full_like(st.h_fun, eps)
    h[sort_id[:cutoff]] = 1.0 / alpha
    leftover = 1.0 - sorted_frac[:cutoff].sum() / alpha
    h[sort_id[cutoff]] = max(leftover / sorted_frac[cutoff], eps)
    if weight_ema:
        h = np.maximum(h, eps)                       # weight_cutoff clamp, :110-112
        st.h_fun = st.h_fun * (1 - ema) + h * ema    # :113-114
    else:
        st.h_fun = h                                 # :117-121


def dro_greedy_forward(st: DROGreedyState, losses: np.ndarray, groups: np.ndarray, weights, n_groups: int, alpha: float,
                       eps: float, ema: float, weight_ema: bool, all_losses=None, all_groups=None):
    """``DROGreedyLoss.forward`` (dro_loss.py:50-90), training mode.  ``all_*`` are the cross-rank gathers (default:
    this rank only).  Returns (robust_loss, row weights the loss gradient carries, group mean losses, group counts)."""
    if weights is not None:
        losses = losses * weights                                            # :51-52
    B = losses.shape[0]
    gsum = np.zeros(n_groups); np.add.at(gsum, groups, losses)
    robust = float((gsum * st.h_fun).sum() / B)                              # :59-60, weights of the PREVIOUS step
    row_w = st.h_fun[groups] * (1.0 if weights is None else weights) / B     # d robust / d (unweighted row loss)
    la = losses if all_losses is None else all_losses
    ga = groups if all_groups is None else all_groups
    cnt_agg = np.zeros(n_groups); np.add.at(cnt_agg, ga, 1.0)                 # :67-69
    sum_agg = np.zeros(n_groups); np.add.at(sum_agg, ga, la)
    mean_agg = sum_agg / (cnt_agg + (cnt_agg == 0))                          # :71
    valid = cnt_agg > 0
    st.sum_losses[valid] = st.sum_losses[valid] * (1 - ema) + ema * mean_agg[valid]  # :75
    st.count_cat = st.count_cat * (1 - ema) + ema * cnt_agg                  # :78-79
    dro_greedy_update_mw(st, alpha, eps, ema, weight_ema)                    # :80
    cnt = np.zeros(n_groups); np.add.at(cnt, groups, 1.0)                    # :82-86 (local statistics returned)
    gl = gsum / (cnt + (cnt == 0))
    return robust, row_w, gl, cnt
