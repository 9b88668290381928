"""The C twins of the C ABI (oracle/cocodr_ref.c: `*_ref`, the signatures of include/cocodr.h on host pointers) against the
numpy oracle - itself pinned to the reference - and, for the contrastive loss, against the reference's own golden vectors
(tests/golden/contrastive_loss.npz: COCO/modeling.py:244-248 run unmodified)."""
import ctypes as C
import os

import numpy as np
import pytest

import cocodr_amd  # noqa: F401
from cocodr_amd import _native as N
import oracle as O
from oracle import ref_twins

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bf16(x):  # round-to-nearest-even bf16 bits of an fp32 array
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def lib():
    ref_twins.build(force=True)
    return ref_twins.lib()


def test_every_twin_binds_with_the_product_signature(lib):
    for name in ref_twins.TWINS:
        assert getattr(lib, name + "_ref").argtypes == N.SIGNATURES[name][1]


@pytest.mark.parametrize("ta,tb,epi", [(0, 0, "none"), (0, 0, "gelu"), (0, 0, "add"), (0, 1, "dgelu"), (0, 1, "add"), (1, 1, "none")])
def test_gemm_twin_matches_numpy(lib, ta, tb, epi):
    rng = np.random.Generator(np.random.PCG64(3))
    M, Nn, K = 24, 16, 40
    a = _bf16(rng.standard_normal((K, M) if ta else (M, K)))
    b = _bf16(rng.standard_normal((K, Nn) if tb else (Nn, K)) * 0.3)
    bias = rng.standard_normal(Nn).astype(np.float32)
    r = _bf16(rng.standard_normal((M, Nn)))
    f32 = ta == 1
    out = np.zeros((M, Nn), np.float32 if f32 else np.uint16)
    c2 = np.zeros((M, Nn), np.uint16)
    g = N.GemmArgs(A=_p(a), B=_p(b), C=_p(out), C2=_p(c2) if epi == "gelu" else None, bias=_p(bias) if epi in ("none", "gelu", "add") else None,
                   R=_p(r) if epi in ("add", "dgelu") else None, M=M, N=Nn, K=K, lda=a.shape[1], ldb=b.shape[1], ldc=Nn, ldr=Nn,
                   trans_a=ta, trans_b=tb, epi={"none": N.EPI_NONE, "gelu": N.EPI_GELU, "add": N.EPI_ADD, "dgelu": N.EPI_DGELU}[epi],
                   out_f32=int(f32), batch=1)
    assert lib.cocodr_gemm_ref(C.byref(g), None) == 0
    A = _f32(a).astype(np.float64)
    B = _f32(b).astype(np.float64)
    acc = (A.T if ta else A) @ (B if tb else B.T)
    if epi in ("none", "gelu", "add"):
        acc = acc + bias
    if epi == "gelu":
        np.testing.assert_array_equal(c2, _bf16(O.gelu_grad(acc).astype(np.float32)))
        acc = O.gelu(acc)
    elif epi == "add":
        acc = acc + _f32(r)
    elif epi == "dgelu":
        acc = acc * _f32(r)
    if f32:
        np.testing.assert_allclose(out, acc, rtol=1e-6, atol=1e-6)
    else:
        want = _bf16(acc.astype(np.float32))
        assert (out != want).mean() < 0.01 and np.abs(_f32(out) - _f32(want)).max() <= np.abs(_f32(want)).max() * 2 ** -7  # ties of the bf16 rounding only


def test_ln_and_attention_twins_match_the_oracle_layer(lib):
    rng = np.random.Generator(np.random.PCG64(5))
    M, H = 12, 128
    y = _bf16(rng.standard_normal((M, H)) * 2 + 0.3)
    gam = rng.standard_normal(H).astype(np.float32)
    bet = rng.standard_normal(H).astype(np.float32)
    out = np.zeros((M, H), np.uint16)
    mean, rstd = np.zeros(M, np.float32), np.zeros(M, np.float32)
    assert lib.cocodr_ln_fwd_ref(_p(y), _p(gam), _p(bet), _p(out), _p(mean), _p(rstd), None, 0, M, H, 1e-12, None) == 0
    want, xhat, rs = O.layer_norm_fwd(_f32(y).astype(np.float64), gam.astype(np.float64), bet.astype(np.float64))
    assert np.abs(_f32(out) - want).max() <= np.abs(want).max() * 2 ** -7 and np.allclose(rstd, rs.ravel(), rtol=1e-5)
    # attention: against the softmax(QK^T / 8 + mask) V of the oracle's layer forward, recomputed here from the same qkv
    B, L, heads = 2, 32, 2
    Hh = heads * 64
    qkv = _bf16(rng.standard_normal((B * L, 3 * Hh)) * 0.7)
    mask = np.ones((B, L), np.int32)
    mask[1, 20:] = 0
    ctx = np.zeros((B * L, Hh), np.uint16)
    lse = np.zeros((B, heads, L), np.float32)
    assert lib.cocodr_attn_fwd_ref(_p(qkv), _p(mask), _p(ctx), _p(lse), B, L, heads, None) == 0
    x = _f32(qkv).astype(np.float64).reshape(B, L, 3, heads, 64)
    q, k, v = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(0, 1, 3, 2) / 8.0 + np.where(mask[:, None, None, :] != 0, 0.0, -1e30)
    p = np.exp(s - s.max(-1, keepdims=True))
    want_lse = np.log(p.sum(-1)) + s.max(-1)
    p /= p.sum(-1, keepdims=True)
    want = (p @ v).transpose(0, 2, 1, 3).reshape(B * L, Hh)
    assert np.abs(_f32(ctx) - want).max() <= np.abs(want).max() * 2 ** -7 and np.allclose(lse, want_lse, rtol=1e-5, atol=1e-5)


def test_backward_twins_match_the_oracle(lib):
    """cocodr_ln_bwd_ref / cocodr_attn_bwd_ref / cocodr_embed_ln_fwd_ref / cocodr_embed_ln_bwd_ref against the numpy oracle's
    layer_norm_bwd, the attention part of layers_bwd and embeddings_fwd (hf modeling_bert.py:68-108, 111-203, 282-293)."""
    rng = np.random.Generator(np.random.PCG64(15))
    # ---- LayerNorm backward
    M, H = 10, 128
    y = _bf16(rng.standard_normal((M, H)) * 1.5 - 0.2)
    dout = _bf16(rng.standard_normal((M, H)))
    gam = rng.standard_normal(H).astype(np.float32)
    yf = _f32(y).astype(np.float64)
    _, xhat, rs = O.layer_norm_fwd(yf, gam.astype(np.float64), np.zeros(H))
    mean = yf.mean(-1).astype(np.float32)
    rstd = rs.ravel().astype(np.float32)
    dy = np.zeros((M, H), np.uint16)
    dg, db, cs = (np.zeros(H, np.float32) for _ in range(3))
    assert lib.cocodr_ln_bwd_ref(_p(dout), _p(y), _p(gam), _p(mean), _p(rstd), _p(dy), _p(dg), _p(db), _p(cs), None, M, H, None) == 0
    wdy, wdg, wdb = O.layer_norm_bwd(_f32(dout).astype(np.float64), xhat, rs, gam.astype(np.float64))
    assert np.abs(_f32(dy) - wdy).max() <= np.abs(wdy).max() * 2 ** -7
    np.testing.assert_allclose(dg, wdg, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db, wdb, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(cs, wdy.sum(0), rtol=1e-5, atol=1e-5)
    # ---- attention backward
    B, L, heads = 2, 32, 2
    Hh = heads * 64
    qkv = _bf16(rng.standard_normal((B * L, 3 * Hh)) * 0.6)
    dctx = _bf16(rng.standard_normal((B * L, Hh)) * 0.5)
    mask = np.ones((B, L), np.int32)
    mask[0, 25:] = 0
    dqkv = np.zeros((B * L, 3 * Hh), np.uint16)
    qkb = np.zeros((4 * B, 2 * Hh), np.float32)
    assert lib.cocodr_attn_bwd_ref(_p(qkv), _p(mask), None, _p(dctx), None, _p(dqkv), _p(qkb), B, L, heads, None) == 0
    x = _f32(qkv).astype(np.float64).reshape(B, L, 3, heads, 64)
    q, k, v = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(0, 1, 3, 2) / 8.0 + np.where(mask[:, None, None, :] != 0, 0.0, -1e30)
    pr = np.exp(s - s.max(-1, keepdims=True))
    pr /= pr.sum(-1, keepdims=True)
    do = _f32(dctx).astype(np.float64).reshape(B, L, heads, 64).transpose(0, 2, 1, 3)
    dv = pr.transpose(0, 1, 3, 2) @ do
    dp = do @ v.transpose(0, 1, 3, 2)
    ds = pr * (dp - (dp * pr).sum(-1, keepdims=True))
    dq, dk = ds @ k / 8.0, ds.transpose(0, 1, 3, 2) @ q / 8.0
    want = np.stack([t.transpose(0, 2, 1, 3).reshape(B * L, Hh) for t in (dq, dk, dv)], 1).reshape(B * L, 3 * Hh)
    assert np.abs(_f32(dqkv) - want).max() <= np.abs(want).max() * 2 ** -7
    got_q = qkb.reshape(B, 4, 2 * Hh).sum(1)
    np.testing.assert_allclose(got_q[:, :Hh], dq.transpose(0, 2, 1, 3).reshape(B, L, Hh).sum(1), rtol=1e-4, atol=1e-5)
    assert not got_q[:, Hh:].any()   # the key-bias gradient vanishes identically (include/cocodr.h, cocodr_attn_bwd)
    # ---- embeddings forward / backward
    ocfg = O.OracleConfig(vocab_size=50, hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128, max_position_embeddings=16)
    P = O.make_params(ocfg, 3, dtype=np.float64, std=0.5)
    Bq, Lq, He = 3, 8, 64
    ids = rng.integers(0, 50, (Bq, Lq)).astype(np.int32)
    f = lambda k_: np.ascontiguousarray(P[k_], np.float32)  # noqa: E731
    word, pos, typ = f("embeddings.word_embeddings.weight"), f("embeddings.position_embeddings.weight"), f("embeddings.token_type_embeddings.weight")
    g_, b_ = f("embeddings.LayerNorm.weight"), f("embeddings.LayerNorm.bias")
    out = np.zeros((Bq * Lq, He), np.uint16)
    mean, rstd = np.zeros(Bq * Lq, np.float32), np.zeros(Bq * Lq, np.float32)
    assert lib.cocodr_embed_ln_fwd_ref(_p(ids), _p(word), _p(pos), _p(typ[0].copy()), _p(g_), _p(b_), _p(out), _p(mean), _p(rstd), Bq, Lq, He, 50,
                                       1e-12, None) == 0
    P32 = {k_: v_.astype(np.float32).astype(np.float64) for k_, v_ in P.items()}
    cache = {}
    want = O.embeddings_fwd(P32, ids.astype(np.int64), cache)
    assert np.abs(_f32(out) - want.reshape(-1, He)).max() <= np.abs(want).max() * 2 ** -7
    dout = _bf16(rng.standard_normal((Bq * Lq, He)))
    dword = np.zeros_like(word)
    dpos, dtyp = np.zeros((Lq, He), np.float32), np.zeros(He, np.float32)
    dg, db = np.zeros(He, np.float32), np.zeros(He, np.float32)
    assert lib.cocodr_embed_ln_bwd_ref(_p(dout), _p(ids), _p(word), _p(pos), _p(typ[0].copy()), _p(g_), _p(mean), _p(rstd), _p(dword), _p(dpos),
                                       _p(dtyp), _p(dg), _p(db), None, Bq, Lq, He, 50, None) == 0
    e = cache["emb"]
    dx, wdg, wdb = O.layer_norm_bwd(_f32(dout).astype(np.float64).reshape(Bq, Lq, He), e["xhat"], e["rstd"], P32["embeddings.LayerNorm.weight"])
    wword = np.zeros_like(word, dtype=np.float64)
    np.add.at(wword, ids.astype(np.int64).ravel(), dx.reshape(-1, He))
    np.testing.assert_allclose(dword, wword, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dpos, dx.sum(0), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dtyp, dx.sum((0, 1)), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dg, wdg, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(db, wdb, rtol=1e-5, atol=1e-5)


def test_simce_twin_matches_the_reference_golden_and_the_oracle_gradient(lib):
    g = np.load(os.path.join(GOLD, "contrastive_loss.npz"))  # COCO/modeling.py:244-248 itself, at world sizes 1 / 2 / 8
    sizes = sorted(int(k[2:]) for k in g.files if k.startswith("E_"))
    assert sizes == [8, 16, 64]
    for M in sizes:
        E = np.ascontiguousarray(g[f"E_{M}"], np.float32)
        H, world = E.shape[1], int(g[f"W_{M}"])
        loss_rows, loss = np.zeros(M, np.float32), np.zeros(1, np.float32)
        dE = np.zeros((M, H), np.float32)
        ws = np.zeros(M * M, np.float32)
        assert lib.cocodr_simce_fwd_bwd_ref(_p(E), M, H, world, 0, M, _p(loss_rows), _p(loss), _p(dE), _p(ws), None) == 0
        np.testing.assert_allclose(loss_rows, g[f"rows_{M}"], rtol=2e-5, atol=2e-5)  # the reference's own per-row losses (x world)
        np.testing.assert_allclose(dE, g[f"dE_{M}"], rtol=1e-4, atol=1e-6)           # and its autograd gradient of rows.mean()
        assert abs(float(loss[0]) - float(g[f"rows_{M}"].mean())) < 1e-5 * abs(float(g[f"rows_{M}"].mean()))
        # a rank's share: rows [row0, row0 + m_local) of the same matrix
        m_local = M // world
        for rank in range(world):
            part = np.zeros((m_local, H), np.float32)
            assert lib.cocodr_simce_fwd_bwd_ref(_p(E), M, H, world, rank * m_local, m_local, _p(loss_rows), _p(loss), _p(part), _p(ws), None) == 0
            np.testing.assert_allclose(part, g[f"dE_{M}"][rank * m_local:(rank + 1) * m_local], rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(part, O.contrastive_local_grad(E.astype(np.float64), world, rank), rtol=1e-4, atol=1e-6)


def test_triplet_search_and_merge_twins_match_the_oracle(lib):
    rng = np.random.Generator(np.random.PCG64(9))
    B, H = 6, 32
    q, a, b = (rng.standard_normal((B, H)).astype(np.float32) * 0.4 for _ in range(3))
    w = rng.random(B).astype(np.float32)
    outs = [np.zeros(B, np.float32), np.zeros((B, 2), np.float32), np.zeros(1, np.float32)] + [np.zeros((B, H), np.float32) for _ in range(3)]
    assert lib.cocodr_triplet_nll_fwd_bwd_ref(_p(q), _p(a), _p(b), _p(w), B, H, *[_p(o) for o in outs], None) == 0
    rl, rq, ra, rb = O.triplet_nll_grad(q.astype(np.float64), a.astype(np.float64), b.astype(np.float64), w.astype(np.float64))
    assert abs(float(outs[2][0]) - rl) < 1e-6
    for got, ref in zip(outs[3:], (rq, ra, rb)):
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-7)
    # search: integer-valued vectors make every score exact, so ids (ties by position) must agree exactly
    Q = rng.integers(-3, 4, (5, 16)).astype(np.float32)
    P = rng.integers(-3, 4, (60, 16)).astype(np.float32)
    k = 70
    D, I = np.zeros((5, k), np.float32), np.zeros((5, k), np.int64)
    ws = np.zeros(60 * 16, np.uint8)
    assert lib.cocodr_score_topk_ref(_p(Q), _p(P), 5, 60, 16, k, 100, _p(D), _p(I), _p(ws), ws.nbytes, None) == 0
    Dr, Ir = O.score_topk(Q, P, k)
    assert np.array_equal(I[:, :60], Ir[:, :60] + 100) and np.array_equal(D[:, :60], Dr[:, :60]) and (I[:, 60:] == -1).all()
    # merge: three shards of sorted lists
    W, Nq, kk = 3, 4, 8
    Dw = np.full((W, Nq, kk), -np.inf, np.float32)
    Iw = np.full((W, Nq, kk), -1, np.int32)
    sizes = [20, 5, 11]
    for wv in range(W):
        for qi in range(Nq):
            m = min(kk, sizes[wv])
            pos = rng.permutation(sizes[wv])[:m]
            sc = rng.integers(0, 4, m).astype(np.float32)
            o = np.lexsort((pos, -sc))
            Dw[wv, qi, :m], Iw[wv, qi, :m] = sc[o], pos[o]
    offs = np.array([0, 20, 25], np.int64)
    oD, oI = np.zeros((Nq, 10), np.float32), np.zeros((Nq, 10), np.int64)
    assert lib.cocodr_topk_merge_ref(_p(Dw), _p(Iw), _p(offs), W, Nq, kk, Nq * kk, _p(oD), _p(oI), 10, None) == 0
    Ig = [np.where(Iw[wv] >= 0, Iw[wv].astype(np.int64) + offs[wv], -1) for wv in range(W)]
    Dr, Ir = O.merge_topk([Dw[wv] for wv in range(W)], Ig, 10)
    assert np.array_equal(oI, Ir) and np.array_equal(oD, np.where(Ir >= 0, Dr, -np.inf))
