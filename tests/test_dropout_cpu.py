"""Dropout: the oracle's restatement of WHERE the reference drops is pinned against the reference's own
BertDot_NLL_LN in train() mode on transformers' BertModel (tests/golden/dropout_sites.npz), its backward against finite
differences, the mask hash against its specification's statistics, and the library's host-side key derivation against the
oracle's.  CPU only."""
import ctypes as C

import numpy as np

from conftest import cfg_from_golden, load_golden, params_from_golden
import oracle as O
from oracle import dropout_oracle as D


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_oracle_dropout_placement_matches_reference_train_mode():
    g = load_golden("dropout_sites.npz")
    cfg = cfg_from_golden(g)
    P = params_from_golden(g)
    embs, caches = [], []
    for call, (ids, mask) in enumerate(((g["q_ids"], g["q_mask"]), (g["a_ids"], g["a_mask"]), (g["b_ids"], g["b_mask"])), start=1):
        drop = dict(p_hidden=float(g["p_hidden"]), p_attn=float(g["p_attn"]), seed=int(g["drop_seed"]), call=call)
        hs, cache = O.encoder_fwd(P, cfg, ids, mask, keep_cache=True, dropout=drop)
        embs.append(O.cls_embedding(hs[-1]))
        caches.append((cache, hs[-1].shape))
    _rows, logits = O.triplet_nll(*embs)
    np.testing.assert_allclose(logits, g["logits"], rtol=2e-5)
    loss, dq, da, db = O.triplet_nll_grad(*embs, weights=g["weights"])
    assert abs(loss - float(g["loss"])) < 1e-5
    G = {}
    for (cache, shape), de in zip(caches, (dq, da, db)):
        d_last = np.zeros(shape, np.float32)
        d_last[:, 0] = de
        for k, v in O.encoder_bwd(P, cfg, cache, d_last).items():
            G[k] = G.get(k, 0) + v
    checked = 0
    for key in g.files:
        if key.startswith("grad:"):
            name = key[5:]
            if name.endswith("key.bias"):  # identically zero (softmax shift invariance), round-off on both sides
                assert np.abs(G[name]).max() < 1e-6 and np.abs(g[key]).max() < 1e-6
            else:
                assert _rel(G[name], g[key]) < 2e-3, (name, _rel(G[name], g[key]))
            checked += 1
    assert checked >= 10
    # and the masks matter: without them the same weights give a different loss
    embs0 = [O.cls_embedding(O.encoder_fwd(P, cfg, g[a], g[b])[0][-1]) for a, b in (("q_ids", "q_mask"), ("a_ids", "a_mask"), ("b_ids", "b_mask"))]
    assert abs(O.triplet_nll_grad(*embs0, weights=g["weights"])[0] - float(g["loss"])) > 1e-3


def test_oracle_dropout_backward_matches_finite_differences():
    cfg = O.OracleConfig(vocab_size=50, hidden_size=16, num_hidden_layers=2, num_attention_heads=2, intermediate_size=32,
                         max_position_embeddings=16)
    P = O.make_params(cfg, 5, dtype=np.float64, std=0.3)
    rng = np.random.Generator(np.random.PCG64(1))
    ids = rng.integers(1, 50, (3, 8))
    mask = np.ones((3, 8), np.int64)
    mask[1, 5:] = 0
    drop = dict(p_hidden=0.3, p_attn=0.25, seed=9, call=4)
    w = rng.standard_normal((3, 8, 16))

    def f(Pq):
        hs, _ = O.encoder_fwd(Pq, cfg, ids, mask, dropout=drop)
        return float((hs[-1] * w).sum())

    hs, cache = O.encoder_fwd(P, cfg, ids, mask, keep_cache=True, dropout=drop)
    G = O.encoder_bwd(P, cfg, cache, w.copy())
    names = ["encoder.layer.0.attention.self.query.weight", "encoder.layer.0.attention.self.value.bias",
             "encoder.layer.0.attention.output.dense.weight", "encoder.layer.1.output.dense.bias",
             "encoder.layer.0.intermediate.dense.weight", "embeddings.LayerNorm.weight", "embeddings.position_embeddings.weight",
             "encoder.layer.1.attention.output.LayerNorm.bias"]
    for name in names:
        flat = P[name].reshape(-1)
        for idx in rng.choice(flat.size, 3, replace=False):
            old = flat[idx]
            eps = 1e-6
            flat[idx] = old + eps
            fp = f(P)
            flat[idx] = old - eps
            fm = f(P)
            flat[idx] = old
            num = (fp - fm) / (2 * eps)
            ana = G[name].reshape(-1)[idx]
            assert abs(num - ana) <= 1e-5 * max(1.0, abs(num)), (name, idx, num, ana)


def test_mask_statistics_and_independence():
    n = 1 << 20
    for p in (0.1, 0.15, 0.5):
        thr, scale = D.threshold_scale(p)
        keep = D.keep_mask((n,), p, seed=3, call=1, layer=0, kind=D.KIND_ATTN_OUT)
        q = 1.0 - thr / 65536.0
        sigma = np.sqrt(q * (1 - q) / n)
        assert abs(keep.mean() - q) < 5 * sigma, (p, keep.mean(), q)
        assert abs(float(scale) * q - 1.0) < 1e-6  # the scale matches the realised keep probability exactly
        # neighbours (the two halves of a hash word), rows and far elements are uncorrelated
        k = keep.astype(np.float64) - q
        for lag in (1, 2, 3, 128, 1024, 4097):
            corr = float((k[:-lag] * k[lag:]).mean() / (q * (1 - q)))
            assert abs(corr) < 5 / np.sqrt(n), (p, lag, corr)
    # different sites / layers / calls / seeds draw unrelated masks
    base = D.keep_mask((n,), 0.1, 3, 1, 0, D.KIND_ATTN_OUT)
    others = [D.keep_mask((n,), 0.1, 3, 1, 0, D.KIND_FFN_OUT), D.keep_mask((n,), 0.1, 3, 1, 1, D.KIND_ATTN_OUT),
              D.keep_mask((n,), 0.1, 3, 2, 0, D.KIND_ATTN_OUT), D.keep_mask((n,), 0.1, 4, 1, 0, D.KIND_ATTN_OUT)]
    q = 1.0 - D.threshold_scale(0.1)[0] / 65536.0
    for o in others:
        corr = float(((base - q) * (o - q)).mean() / (q * (1 - q)))
        assert abs(corr) < 5 / np.sqrt(n), corr
    assert D.keep_mask((64,), 0.0, 1, 1, 0, 0).all()


def test_library_key_derivation_matches_oracle():
    from cocodr_amd import _native as N
    lib = N.lib()
    for p, seed, call, layer, kind in ((0.1, 0, 0, 0, 3), (0.1, 777, 1, 0, 0), (0.15, 2 ** 63 - 5, 123456789, 23, 2),
                                      (0.5, 42, 7, 11, 1), (0.0, 1, 1, 1, 1)):
        dm = N.DropoutMask()
        assert lib.cocodr_dropout_mask_for(p, seed, call, layer, kind, C.byref(dm)) == 0
        k0, k1 = D.site_keys(seed, call, layer, kind)
        thr, scale = D.threshold_scale(p)
        assert (dm.k0, dm.k1, dm.threshold) == (k0, k1, thr)
        assert np.float32(dm.scale) == np.float32(scale)
    dm = N.DropoutMask()
    assert lib.cocodr_dropout_mask_for(1.0, 0, 0, 0, 0, C.byref(dm)) != 0  # p must stay below 1
    assert lib.cocodr_dropout_mask_for(0.1, 0, 0, 0, 4, C.byref(dm)) != 0  # unknown site kind
