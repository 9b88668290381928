"""numpy restatement of the BERT bi-encoder forward/backward and the two
training losses on the COCO-DR hot path.  TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).

Every function cites the reference line it follows (paths relative to
/root/reference; ``hf:`` = transformers/models/bert/modeling_bert.py of the
installed transformers 5.15.0, the third-party package the reference calls).

All arithmetic is done in the dtype of the parameters handed in (float32 to
mirror the reference CPU path, float64 for a tighter checker).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
from scipy.special import erf

__all__ = [
    "OracleConfig", "make_params", "layer_names", "embeddings_fwd", "encoder_fwd",
    "encoder_bwd", "cls_embedding", "co_target", "contrastive_loss",
    "contrastive_loss_grad", "contrastive_local_grad", "triplet_nll",
    "triplet_nll_grad", "gelu", "gelu_grad", "layer_norm_fwd", "layer_norm_bwd", "layers_bwd", "_layer_fwd",
    "bf16_storage", "bf16_weights", "round_bf16",
]

LN_EPS = 1e-12  # BertConfig.layer_norm_eps default (hf: BertEmbeddings / BertSelfOutput)


@dataclass
class OracleConfig:
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def layer_names(i: int, stack: str = "encoder.layer.") -> Dict[str, str]:
    p = f"{stack}{i}."
    return dict(
        wq=p + "attention.self.query.weight", bq=p + "attention.self.query.bias",
        wk=p + "attention.self.key.weight", bk=p + "attention.self.key.bias",
        wv=p + "attention.self.value.weight", bv=p + "attention.self.value.bias",
        wo=p + "attention.output.dense.weight", bo=p + "attention.output.dense.bias",
        g1=p + "attention.output.LayerNorm.weight", b1=p + "attention.output.LayerNorm.bias",
        w1=p + "intermediate.dense.weight", bi=p + "intermediate.dense.bias",
        w2=p + "output.dense.weight", b2=p + "output.dense.bias",
        g2=p + "output.LayerNorm.weight", be2=p + "output.LayerNorm.bias",
    )


def make_params(cfg: OracleConfig, seed: int = 0, dtype=np.float32, std: float = 0.02,
                perturb_ln: bool = True) -> Dict[str, np.ndarray]:
    """Seeded random-init parameters under HF BertModel state-dict names.

    HF ``_init_weights`` is normal(0, 0.02) for Linear/Embedding, ones/zeros for
    LayerNorm (ANCE/model/models.py:54-60 restates the same rule).  With
    ``perturb_ln`` the LayerNorm gains/biases and all Linear biases get small
    random values so that parity tests exercise them.
    numpy's PCG64 stream is stable across versions, so fixtures only store the seed.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    H, I = cfg.hidden_size, cfg.intermediate_size
    P: Dict[str, np.ndarray] = {}

    def nrm(*shape):
        return (rng.standard_normal(shape) * std).astype(dtype)

    def ln(name):
        if perturb_ln:
            P[name + ".weight"] = (1.0 + 0.1 * rng.standard_normal(H)).astype(dtype)
            P[name + ".bias"] = (0.05 * rng.standard_normal(H)).astype(dtype)
        else:
            P[name + ".weight"] = np.ones(H, dtype)
            P[name + ".bias"] = np.zeros(H, dtype)

    def bias(n):
        return (0.02 * rng.standard_normal(n)).astype(dtype) if perturb_ln else np.zeros(n, dtype)

    P["embeddings.word_embeddings.weight"] = nrm(cfg.vocab_size, H)
    P["embeddings.position_embeddings.weight"] = nrm(cfg.max_position_embeddings, H)
    P["embeddings.token_type_embeddings.weight"] = nrm(cfg.type_vocab_size, H)
    ln("embeddings.LayerNorm")
    for i in range(cfg.num_hidden_layers):
        n = layer_names(i)
        for w, b, shape in (("wq", "bq", (H, H)), ("wk", "bk", (H, H)), ("wv", "bv", (H, H)),
                            ("wo", "bo", (H, H))):
            P[n[w]] = nrm(*shape)
            P[n[b]] = bias(shape[0])
        ln(n["g1"][: -len(".weight")])
        P[n["w1"]] = nrm(I, H)
        P[n["bi"]] = bias(I)
        P[n["w2"]] = nrm(H, I)
        P[n["b2"]] = bias(H)
        ln(n["g2"][: -len(".weight")])
    return P


# --------------------------------------------------------------------------- primitives
def gelu(x: np.ndarray) -> np.ndarray:
    """Exact erf GELU - ``hidden_act="gelu"`` (hf: BertIntermediate)."""
    return (0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))).astype(x.dtype)


def gelu_grad(x: np.ndarray) -> np.ndarray:
    cdf = 0.5 * (1.0 + erf(x / np.sqrt(2.0)))
    pdf = np.exp(-0.5 * x * x) / np.sqrt(2.0 * np.pi)
    return (cdf + x * pdf).astype(x.dtype)


# ---- optional "bf16 storage" mode (tests only): the device path keeps every activation it STORES between kernels - and the
# probability / score-gradient operands of the attention MFMAs - in bfloat16 (fp32 accumulation inside the kernels, fp32 LayerNorm
# statistics, fp32 master weights with a bf16 copy for the GEMMs).  Inside ``bf16_storage()`` the oracle rounds exactly those
# tensors (round to nearest even), so that what remains between the two is summation order and the kernels' internal roundings -
# the end-to-end comparison can then be held to ~1e-2 instead of the 6-8e-2 that fp32-vs-bf16 needs.
_ROUND = None


def round_bf16(x: np.ndarray) -> np.ndarray:
    """round-to-nearest-even to bfloat16 precision, returned in x's dtype"""
    f = np.ascontiguousarray(x, np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).reshape(f.shape).astype(x.dtype)


class bf16_storage:
    def __enter__(self):
        global _ROUND
        self._old, _ROUND = _ROUND, round_bf16
        return self

    def __exit__(self, *exc):
        global _ROUND
        _ROUND = self._old
        return False


def _r(x):
    return x if _ROUND is None else _ROUND(x)


def bf16_weights(P: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """the parameters as the device's GEMMs see them: the weight MATRICES of the layers rounded to bf16 (the bf16 shadow of the
    fp32 master copy); embedding tables, biases and LayerNorm parameters stay fp32"""
    return {k: (round_bf16(v) if (k.endswith(".weight") and v.ndim == 2 and "embeddings" not in k) else v) for k, v in P.items()}


def layer_norm_fwd(y: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float = LN_EPS):
    """torch.nn.LayerNorm over the last axis, biased variance (hf: BertSelfOutput.LayerNorm)."""
    mu = y.mean(-1, keepdims=True)
    var = ((y - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = (y - mu) * rstd
    return (xhat * g + b).astype(y.dtype), xhat.astype(y.dtype), rstd.astype(y.dtype)


def layer_norm_bwd(dout: np.ndarray, xhat: np.ndarray, rstd: np.ndarray, g: np.ndarray):
    dg = (dout * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    db = dout.reshape(-1, xhat.shape[-1]).sum(0)
    dxhat = dout * g
    dy = rstd * (dxhat - dxhat.mean(-1, keepdims=True) - xhat * (dxhat * xhat).mean(-1, keepdims=True))
    return dy, dg, db


# --------------------------------------------------------------------------- encoder
def _drop_mult(dropout, shape, layer: int, kind: int, dtype):
    """Multiplier (0 or 1 / (1 - p)) of one dropout site, or None.  ``dropout`` = dict(p_hidden, p_attn, seed, call) - the
    hf nn.Dropout sites under model.train() (see oracle/dropout_oracle.py) - or a callable (shape, layer, kind) -> array."""
    if dropout is None:
        return None
    if callable(dropout):
        return dropout(shape, layer, kind)
    from . import dropout_oracle as D
    p = dropout["p_attn"] if kind == D.KIND_ATTN_PROBS else dropout["p_hidden"]
    if p <= 0:
        return None
    return D.multiplier(shape, p, dropout["seed"], dropout["call"], layer, kind, dtype)


def embeddings_fwd(P, input_ids: np.ndarray, cache: Optional[dict] = None, dropout=None) -> np.ndarray:
    """hf: BertEmbeddings.forward - LN(word[ids] + type[0] + pos[0..L-1]).

    The reference never passes token_type_ids or position_ids
    (COCO/data.py:140 ``return_token_type_ids=False``; ANCE/model/models.py:226-227
    passes only ids+mask) so segment row 0 and positions 0..L-1 are used always.
    """
    B, L = input_ids.shape
    we = P["embeddings.word_embeddings.weight"]
    pe = P["embeddings.position_embeddings.weight"]
    te = P["embeddings.token_type_embeddings.weight"]
    y = we[input_ids] + pe[None, :L] + te[0][None, None]
    out, xhat, rstd = layer_norm_fwd(y, P["embeddings.LayerNorm.weight"], P["embeddings.LayerNorm.bias"])
    out = _r(out)
    m = _drop_mult(dropout, out.shape, 0, 3, out.dtype.type)  # hf BertEmbeddings: dropout(LayerNorm(..))
    if m is not None:
        out = out * m
    if cache is not None:
        cache["emb"] = dict(xhat=xhat, rstd=rstd, ids=input_ids, m=m)
    return out


def _attn_ctx_blocked(s: np.ndarray, vh: np.ndarray, block: int = 32) -> np.ndarray:
    """softmax(s) @ vh the way a one-pass (online-softmax) kernel accumulates it; ``s`` [B, nh, L, L] are the masked scores MINUS
    their row maximum (masked keys: -inf), so exp(s) <= 1.  Only used in bf16-storage mode (the rounding of the un-normalised
    block probabilities is the point)."""
    B, nh, L, _ = s.shape
    m = np.full((B, nh, L), -np.inf)
    l = np.zeros((B, nh, L))
    o = np.zeros((B, nh, L, vh.shape[-1]))
    for k0 in range(0, L, block):
        sb = s[..., k0:k0 + block].astype(np.float64)
        mnew = np.maximum(m, sb.max(-1))
        safe = np.where(np.isfinite(mnew), mnew, 0.0)
        alpha = np.where(np.isfinite(m), np.exp(m - safe), 0.0)
        pb = np.exp(sb - safe[..., None])          # masked keys: exp(-inf) = 0
        l = l * alpha + pb.sum(-1)
        o = o * alpha[..., None] + round_bf16(pb.astype(np.float32)).astype(np.float64) @ vh[..., k0:k0 + block, :].astype(np.float64)
        m = mnew
    return (o / l[..., None]).astype(vh.dtype)


def _layer_fwd(P, i: int, x: np.ndarray, mask: np.ndarray, nh: int, cache: Optional[dict], stack: str = "encoder.layer.",
               dropout=None):
    """One BertLayer: hf BertSelfAttention (eager) + BertSelfOutput + BertIntermediate + BertOutput."""
    n = layer_names(i, stack)
    B, L, H = x.shape
    d = H // nh
    q = _r(x @ P[n["wq"]].T + P[n["bq"]])
    k = _r(x @ P[n["wk"]].T + P[n["bk"]])
    v = _r(x @ P[n["wv"]].T + P[n["bv"]])

    def heads(t):
        return t.reshape(B, L, nh, d).transpose(0, 2, 1, 3)  # [B,nh,L,d]

    qh, kh, vh = heads(q), heads(k), heads(v)
    s = (qh @ kh.transpose(0, 1, 3, 2)) * x.dtype.type(1.0 / np.sqrt(d))
    # key-padding mask: additive 0 / finfo.min in hf 5.x (-10000 in <=4.x); both give exactly 0
    # probability after the fp32 softmax whenever >=1 key is unmasked.
    s = np.where(mask[:, None, None, :] != 0, s, -np.inf)
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    p = (e / e.sum(-1, keepdims=True)).astype(x.dtype)
    t = x.dtype.type
    mp = _drop_mult(dropout, p.shape, i, 0, t)   # hf eager_attention_forward: dropout(softmax(..))
    pd = p if mp is None else p * mp
    if _ROUND is not None and mp is None:
        # bf16-storage mode: the output as the device's one-pass kernel forms it (csrc/attention.hip attn_fwd_rows) - keys in blocks
        # of 32, a running row maximum, the UN-normalised exp(s - max so far) rounded to bf16 for the product with V, fp32 running
        # sum of the unrounded values, one division at the end.  Rounding the normalised probabilities instead (the line below) is
        # the same noise in another realisation: ~20 % of the output's bf16 values come out one ulp apart
        ctx = _r(_attn_ctx_blocked(s, vh).transpose(0, 2, 1, 3).reshape(B, L, H))
    else:
        ctx = _r((_r(pd) @ vh).transpose(0, 2, 1, 3).reshape(B, L, H))
    a = ctx @ P[n["wo"]].T + P[n["bo"]]
    ma = _drop_mult(dropout, a.shape, i, 1, t)   # hf BertSelfOutput: LayerNorm(dropout(dense(..)) + input)
    if ma is not None:
        a = a * ma
    x1, xhat1, rstd1 = layer_norm_fwd(_r(a + x), P[n["g1"]], P[n["b1"]])
    x1 = _r(x1)
    u = x1 @ P[n["w1"]].T + P[n["bi"]]
    h = _r(gelu(u))
    f = h @ P[n["w2"]].T + P[n["b2"]]
    mf = _drop_mult(dropout, f.shape, i, 2, t)   # hf BertOutput: LayerNorm(dropout(dense(..)) + input)
    if mf is not None:
        f = f * mf
    x2, xhat2, rstd2 = layer_norm_fwd(_r(f + x1), P[n["g2"]], P[n["be2"]])
    x2 = _r(x2)
    if cache is not None:
        cache[i] = dict(x=x, qh=qh, kh=kh, vh=vh, p=p, ctx=ctx, xhat1=xhat1, rstd1=rstd1, x1=x1,
                        u=u, h=h, xhat2=xhat2, rstd2=rstd2, mp=mp, ma=ma, mf=mf)
    return x2


def encoder_fwd(P, cfg: OracleConfig, input_ids: np.ndarray, attention_mask: np.ndarray,
                keep_cache: bool = False, dropout=None):
    """BertModel forward -> list of N+1 hidden states (``output_hidden_states=True``,
    COCO/modeling.py:199-204) and the cache the backward needs.  ``dropout``: the train()-mode forward (see _drop_mult)."""
    cache = {} if keep_cache else None
    x = embeddings_fwd(P, input_ids, cache, dropout)
    hs = [x]
    for i in range(cfg.num_hidden_layers):
        x = _layer_fwd(P, i, x, attention_mask, cfg.num_attention_heads, cache, dropout=dropout)
        hs.append(x)
    if keep_cache:
        cache["mask"] = attention_mask
    return hs, cache


def cls_embedding(last_hidden: np.ndarray) -> np.ndarray:
    """Raw last-layer [CLS], no pooler/projection/norm.
    ANCE/model/models.py:225-229 ``outputs1[0][:, 0]``; COCO/modeling.py:206 ``hidden_states[-1][:, :1]``."""
    return last_hidden[:, 0]


def layers_bwd(P, nh: int, cache: dict, layer_ids, dx: np.ndarray, G: Dict[str, np.ndarray], stack: str = "encoder.layer.",
               extra: Optional[Dict[int, np.ndarray]] = None) -> np.ndarray:
    """Backward through the BertLayers ``layer_ids`` (walked in reverse); returns dL/d(input of the first one).
    ``extra[i]`` is added to the gradient of hidden state i (the INPUT of layer i) - how the Condenser head's
    gradient re-enters the backbone at ``skip_from`` (COCO/modeling.py:212-213)."""
    for i in reversed(list(layer_ids)):
        n = layer_names(i, stack)
        c = cache[i]
        B, L, H = c["x"].shape
        d = H // nh
        dres2, G[n["g2"]], G[n["be2"]] = layer_norm_bwd(dx, c["xhat2"], c["rstd2"], P[n["g2"]])
        dres2 = _r(dres2)
        dy2 = dres2 if c.get("mf") is None else dres2 * c["mf"]   # gradient of the dense output behind its dropout
        G[n["w2"]] = dy2.reshape(-1, H).T @ c["h"].reshape(-1, c["h"].shape[-1])
        G[n["b2"]] = dy2.reshape(-1, H).sum(0)
        dh = dy2 @ P[n["w2"]]
        du = _r(dh * _r(gelu_grad(c["u"])))
        G[n["w1"]] = du.reshape(-1, du.shape[-1]).T @ c["x1"].reshape(-1, H)
        G[n["bi"]] = du.reshape(-1, du.shape[-1]).sum(0)
        dx1 = _r(du @ P[n["w1"]] + dres2)
        dres1, G[n["g1"]], G[n["b1"]] = layer_norm_bwd(dx1, c["xhat1"], c["rstd1"], P[n["g1"]])
        dres1 = _r(dres1)
        dy1 = dres1 if c.get("ma") is None else dres1 * c["ma"]
        G[n["wo"]] = dy1.reshape(-1, H).T @ c["ctx"].reshape(-1, H)
        G[n["bo"]] = dy1.reshape(-1, H).sum(0)
        dctx = _r(dy1 @ P[n["wo"]]).reshape(B, L, nh, d).transpose(0, 2, 1, 3)
        p = c["p"]
        pd = p if c.get("mp") is None else p * c["mp"]
        dv = _r(_r(pd).transpose(0, 1, 3, 2) @ dctx)
        dp = dctx @ c["vh"].transpose(0, 1, 3, 2)
        if c.get("mp") is not None:
            dp = dp * c["mp"]
        if _ROUND is None:
            delta = (dp * p).sum(-1, keepdims=True)
        else:
            # bf16-storage mode: the row term as the device forms it - delta = rowsum(dO * O) from the STORED (bf16) attention output
            # (csrc/attention.hip: "delta = rowsum(dO * O)"), not from the probabilities.  Mathematically the same number; but the
            # query / key gradients of a near-uniform attention are the small difference dP - delta, and the rounding of O moves delta
            # by 2^-9 of itself - as large as that difference.  Formed the same way, the two sides make the same error.
            oh = c["ctx"].reshape(B, L, nh, d).transpose(0, 2, 1, 3)
            delta = (dctx * oh).sum(-1, keepdims=True)
        ds = _r(p * (dp - delta))
        scale = p.dtype.type(1.0 / np.sqrt(d))
        dq = _r((ds @ c["kh"]) * scale)
        dk = _r((ds.transpose(0, 1, 3, 2) @ c["qh"]) * scale)

        def merge(t):
            return t.transpose(0, 2, 1, 3).reshape(B, L, H)

        dq, dk, dv = merge(dq), merge(dk), merge(dv)
        xf = c["x"].reshape(-1, H)
        G[n["wq"]] = dq.reshape(-1, H).T @ xf
        G[n["wk"]] = dk.reshape(-1, H).T @ xf
        G[n["wv"]] = dv.reshape(-1, H).T @ xf
        G[n["bq"]] = dq.reshape(-1, H).sum(0)
        G[n["bk"]] = dk.reshape(-1, H).sum(0)
        G[n["bv"]] = dv.reshape(-1, H).sum(0)
        dx = _r(dq @ P[n["wq"]] + dk @ P[n["wk"]] + dv @ P[n["wv"]] + dres1)
        if extra is not None and i in extra:
            dx = dx + extra[i]
    return dx


def encoder_bwd(P, cfg: OracleConfig, cache: dict, d_last: np.ndarray,
                extra: Optional[Dict[int, np.ndarray]] = None) -> Dict[str, np.ndarray]:
    """Reverse-mode gradient of ``encoder_fwd`` w.r.t. every parameter given dL/d(hidden_states[-1]) (plus optional
    gradients ``extra[i]`` w.r.t. intermediate hidden_states[i])."""
    G: Dict[str, np.ndarray] = {}
    dx = layers_bwd(P, cfg.num_attention_heads, cache, range(cfg.num_hidden_layers), d_last, G, extra=extra)
    e = cache["emb"]
    if e.get("m") is not None:
        dx = dx * e["m"]
    dy, G["embeddings.LayerNorm.weight"], G["embeddings.LayerNorm.bias"] = layer_norm_bwd(
        dx, e["xhat"], e["rstd"], P["embeddings.LayerNorm.weight"])
    B, L, H = dy.shape
    gw = np.zeros_like(P["embeddings.word_embeddings.weight"])
    np.add.at(gw, e["ids"].reshape(-1), dy.reshape(-1, H))
    G["embeddings.word_embeddings.weight"] = gw
    gp = np.zeros_like(P["embeddings.position_embeddings.weight"])
    gp[:L] = dy.sum(0)
    G["embeddings.position_embeddings.weight"] = gp
    gt = np.zeros_like(P["embeddings.token_type_embeddings.weight"])
    gt[0] = dy.sum((0, 1))
    G["embeddings.token_type_embeddings.weight"] = gt
    return G


# --------------------------------------------------------------------------- losses
def co_target(m: int) -> np.ndarray:
    """COCO/modeling.py:172-177 - ``arange(m).view(-1,2).flip([1]).flatten()`` = [1,0,3,2,...]."""
    return np.arange(m, dtype=np.int64).reshape(-1, 2)[:, ::-1].reshape(-1).copy()


def _log_softmax_rows(S: np.ndarray) -> np.ndarray:
    mx = S.max(-1, keepdims=True)
    return S - mx - np.log(np.exp(S - mx).sum(-1, keepdims=True))


def contrastive_loss(E: np.ndarray, world_size: int = 1) -> np.ndarray:
    """COCO/modeling.py:244-248 - S = E.E^T, diagonal = -inf, CE(S, co_target, 'none') * world.
    Returns the per-row loss [M]; the caller takes ``.mean()`` (COCO/modeling.py:229)."""
    M = E.shape[0]
    S = E @ E.T
    S[np.arange(M), np.arange(M)] = -np.inf
    ls = _log_softmax_rows(S)
    return (-ls[np.arange(M), co_target(M)] * world_size).astype(E.dtype)


def contrastive_loss_grad(E: np.ndarray, world_size: int = 1) -> Tuple[float, np.ndarray]:
    """loss = mean(contrastive_loss(E)) and dloss/dE for ALL rows (single-process autograd result)."""
    M = E.shape[0]
    S = E @ E.T
    S[np.arange(M), np.arange(M)] = -np.inf
    ls = _log_softmax_rows(S)
    t = co_target(M)
    loss = float((-ls[np.arange(M), t]).mean() * world_size)
    G = np.exp(ls)
    G[np.arange(M), t] -= 1.0
    G *= world_size / M
    dE = G @ E + G.T @ E
    return loss, dE.astype(E.dtype)


def contrastive_local_grad(E: np.ndarray, world_size: int, rank: int) -> np.ndarray:
    """Gradient that reaches rank ``rank``'s own rows under the reference's gather
    (COCO/modeling.py:182-186: only slot ``local_rank`` carries autograd history).
    Equals rows [rank*m:(rank+1)*m] of (W/M)(G E + G^T E) - SURVEY 8(e)."""
    _, dE = contrastive_loss_grad(E, world_size)
    m = E.shape[0] // world_size
    return dE[rank * m:(rank + 1) * m]


def triplet_nll(q: np.ndarray, a: np.ndarray, b: np.ndarray):
    """ANCE/model/models.py:97-106 - logits = [sum(q*a), sum(q*b)]; loss = -log_softmax(logits)[:,0]."""
    logits = np.stack([(q * a).sum(-1), (q * b).sum(-1)], 1)
    ls = _log_softmax_rows(logits)
    return (-ls[:, 0]).astype(q.dtype), logits.astype(q.dtype)


def triplet_nll_grad(q, a, b, weights: Optional[np.ndarray] = None):
    """``(loss*weights).mean()`` (ANCE/model/models.py:260-261) and its gradient w.r.t. q, a, b."""
    B = q.shape[0]
    loss, logits = triplet_nll(q, a, b)
    w = np.ones(B, q.dtype) if weights is None else weights.astype(q.dtype)
    p = np.exp(_log_softmax_rows(logits))
    dl = p.copy()
    dl[:, 0] -= 1.0
    dl *= (w / B)[:, None]
    dq = dl[:, :1] * a + dl[:, 1:] * b
    da = dl[:, :1] * q
    db = dl[:, 1:] * q
    return float((loss * w).mean()), dq.astype(q.dtype), da.astype(q.dtype), db.astype(q.dtype)
