"""numpy restatement of the full coCondenser pre-training step: Condenser head + the two masked-LM losses on top of
the encoder / contrastive path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows COCO/modeling.py:192-235 (forward), :87-93 (mlm_loss), :43-46 (c_head = n_head_layers BertLayers) and hf
``BertOnlyMLMHead`` = BertPredictionHeadTransform (dense + erf-GELU + LayerNorm) + decoder tied to the word embeddings
(+ bias).  Parameter names are the reference's state-dict names without the ``lm.`` / ``lm.bert.`` prefixes:
encoder ``encoder.layer.{i}...``, head ``c_head.{i}...``, MLM ``cls.predictions.transform.dense.weight`` etc.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import bert_oracle as B

__all__ = ["make_head_params", "mlm_head_fwd", "mlm_head_bwd", "condenser_step"]


def make_head_params(cfg: "B.OracleConfig", n_head_layers: int, seed: int, dtype=np.float32, std: float = 0.02):
    """Seeded parameters of the Condenser head layers and the MLM head (same init rules as make_params)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    P: Dict[str, np.ndarray] = {}

    def nrm(*shape):
        return (rng.standard_normal(shape) * std).astype(dtype)

    def vec(n, scale):
        return (scale * rng.standard_normal(n)).astype(dtype)

    for i in range(n_head_layers):
        n = B.layer_names(i, "c_head.")
        for w, b in (("wq", "bq"), ("wk", "bk"), ("wv", "bv"), ("wo", "bo")):
            P[n[w]] = nrm(H, H)
            P[n[b]] = vec(H, 0.02)
        P[n["g1"]] = (1.0 + vec(H, 0.1)).astype(dtype)
        P[n["b1"]] = vec(H, 0.05)
        P[n["w1"]] = nrm(I, H)
        P[n["bi"]] = vec(I, 0.02)
        P[n["w2"]] = nrm(H, I)
        P[n["b2"]] = vec(H, 0.02)
        P[n["g2"]] = (1.0 + vec(H, 0.1)).astype(dtype)
        P[n["be2"]] = vec(H, 0.05)
    P["cls.predictions.transform.dense.weight"] = nrm(H, H)
    P["cls.predictions.transform.dense.bias"] = vec(H, 0.02)
    P["cls.predictions.transform.LayerNorm.weight"] = (1.0 + vec(H, 0.1)).astype(dtype)
    P["cls.predictions.transform.LayerNorm.bias"] = vec(H, 0.05)
    P["cls.predictions.bias"] = vec(V, 0.02)
    return P


def mlm_head_fwd(Pm, word: np.ndarray, x: np.ndarray):
    """hf BertOnlyMLMHead on rows x [n,H] -> logits [n,V]."""
    a = x @ Pm["cls.predictions.transform.dense.weight"].T + Pm["cls.predictions.transform.dense.bias"]
    g = B.gelu(a)
    t, xhat, rstd = B.layer_norm_fwd(g, Pm["cls.predictions.transform.LayerNorm.weight"],
                                     Pm["cls.predictions.transform.LayerNorm.bias"])
    logits = t @ word.T + Pm["cls.predictions.bias"]
    return logits, dict(x=x, a=a, xhat=xhat, rstd=rstd, t=t)


def mlm_head_bwd(Pm, word, cache, dlogits, G):
    t = cache["t"]
    G["cls.predictions.bias"] = G.get("cls.predictions.bias", 0) + dlogits.sum(0)
    G["embeddings.word_embeddings.weight"] = G.get("embeddings.word_embeddings.weight", 0) + dlogits.T @ t
    dt = dlogits @ word
    dg, dgam, dbet = B.layer_norm_bwd(dt, cache["xhat"], cache["rstd"], Pm["cls.predictions.transform.LayerNorm.weight"])
    G["cls.predictions.transform.LayerNorm.weight"] = G.get("cls.predictions.transform.LayerNorm.weight", 0) + dgam
    G["cls.predictions.transform.LayerNorm.bias"] = G.get("cls.predictions.transform.LayerNorm.bias", 0) + dbet
    da = dg * B.gelu_grad(cache["a"])
    G["cls.predictions.transform.dense.weight"] = G.get("cls.predictions.transform.dense.weight", 0) + da.T @ cache["x"]
    G["cls.predictions.transform.dense.bias"] = G.get("cls.predictions.transform.dense.bias", 0) + da.sum(0)
    return da @ Pm["cls.predictions.transform.dense.weight"]


def _ce_mean(logits, labels):
    mx = logits.max(-1, keepdims=True)
    lse = mx[:, 0] + np.log(np.exp(logits - mx).sum(-1))
    n = logits.shape[0]
    loss = float((lse - logits[np.arange(n), labels]).mean())
    d = np.exp(logits - lse[:, None])
    d[np.arange(n), labels] -= 1.0
    return loss, (d / n).astype(logits.dtype)


def condenser_step(P, Ph, cfg: "B.OracleConfig", input_ids, attention_mask, labels, n_head_layers: int, skip_from: int,
                   late_mlm: bool, world_size: int = 1, head_dropout=None):
    """Loss and all gradients of CoCondenserForPretraining.forward (single process): returns
    (total, parts dict(mlm_head, mlm_late, co), G encoder grads, Gh head+MLM grads).
    ``head_dropout`` (see bert_oracle._drop_mult): the c_head BertLayers in train() mode - the backbone never drops
    (COCO/modeling.py:198 ``self.lm.eval()``), the head layers belong to the module the trainer puts in train()."""
    nh = cfg.num_attention_heads
    hs, cache = B.encoder_fwd(P, cfg, input_ids, attention_mask, keep_cache=True)
    last = hs[-1]
    word = P["embeddings.word_embeddings.weight"]
    # --- Condenser head: cat(cls of the last layer, skip_from hidden states without their first token)
    x = np.concatenate([last[:, :1], hs[skip_from][:, 1:]], axis=1)
    hcache = {}
    for i in range(n_head_layers):
        x = B._layer_fwd(Ph, i, x, attention_mask, nh, hcache, stack="c_head.", dropout=head_dropout)
    lab_mask = labels != -100
    rows = np.nonzero(lab_mask.reshape(-1))[0]
    lab = labels.reshape(-1)[rows]
    H = cfg.hidden_size
    Gh: Dict[str, np.ndarray] = {}
    logits_h, ch = mlm_head_fwd(Ph, word, x.reshape(-1, H)[rows])
    loss_h, dlog_h = _ce_mean(logits_h, lab)
    parts = dict(mlm_head=loss_h, mlm_late=0.0)
    d_last = np.zeros_like(last)
    dxg = mlm_head_bwd(Ph, word, ch, dlog_h, Gh)
    d_head_out = np.zeros_like(x).reshape(-1, H)
    d_head_out[rows] = dxg
    if late_mlm:
        logits_l, cl = mlm_head_fwd(Ph, word, last.reshape(-1, H)[rows])
        loss_l, dlog_l = _ce_mean(logits_l, lab)
        parts["mlm_late"] = loss_l
        dxl = mlm_head_bwd(Ph, word, cl, dlog_l, Gh)
        d_last.reshape(-1, H)[rows] += dxl
    d_in = B.layers_bwd(Ph, nh, hcache, range(n_head_layers), d_head_out.reshape(x.shape), Gh, stack="c_head.")
    d_last[:, :1] += d_in[:, :1]
    d_skip = d_in.copy()
    d_skip[:, :1] = 0
    # --- contrastive part
    E = B.cls_embedding(last)
    co, dE = B.contrastive_loss_grad(E.copy(), world_size)
    parts["co"] = co
    d_last[:, 0] += dE
    G = B.encoder_bwd(P, cfg, cache, d_last, extra={skip_from: d_skip})
    G["embeddings.word_embeddings.weight"] = G["embeddings.word_embeddings.weight"] + Gh.pop("embeddings.word_embeddings.weight")
    total = parts["mlm_head"] + parts["mlm_late"] + parts["co"]
    return total, parts, G, Gh
