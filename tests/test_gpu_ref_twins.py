"""Device entry point vs its C twin (oracle/cocodr_ref.c, `*_ref`: the SAME signature on host pointers), called with the same
argument list - the form SURVEY 8(b) names for parity tests.  The twins themselves are pinned on CPU (tests/test_ref_twins_cpu.py:
numpy oracle + the reference's contrastive-loss golden).  Tolerances: bf16 outputs one bf16 ulp of the largest value (the device
accumulates in fp32 on MFMA, the twin in fp64), fp32 outputs 1e-5 relative, positions exact."""
import ctypes as C

import numpy as np
import pytest
import torch

import cocodr_amd  # noqa: F401
from cocodr_amd import _native as N
from cocodr_amd import ops
from cocodr_amd._native import lib as dev_lib, stream_ptr
from oracle import ref_twins  # checker

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


def _bf(x):
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(torch.bfloat16)


def _bits(t):  # bf16 tensor -> uint16 numpy
    return t.cpu().view(torch.int16).numpy().view(np.uint16)


def _f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


@pytest.mark.parametrize("epi,tb,drop", [("none", 0, False), ("gelu", 0, False), ("add", 0, True), ("dgelu", 1, False), ("add", 1, False)])
def test_gemm_device_vs_twin(epi, tb, drop):
    rng = np.random.Generator(np.random.PCG64(11))
    M, Nn, K = 300, 256, 192
    a = _bf(rng.standard_normal((M, K)))
    b = _bf(rng.standard_normal((K, Nn) if tb else (Nn, K)) * 0.2)
    bias = torch.from_numpy(rng.standard_normal(Nn).astype(np.float32))
    r = _bf(rng.standard_normal((M, Nn)))
    code = {"none": N.EPI_NONE, "gelu": N.EPI_GELU, "add": N.EPI_ADD, "dgelu": N.EPI_DGELU}[epi]
    dm = ops.dropout_mask(0.1, 7, 3, 2, ops.KIND_FFN_OUT) if drop else None

    def args(A, B, Cc, C2, bias_, R):
        g = N.GemmArgs(A=A, B=B, C=Cc, C2=C2 if epi == "gelu" else None, bias=bias_ if epi in ("none", "gelu", "add") else None,
                       R=R if epi in ("add", "dgelu") else None, M=M, N=Nn, K=K, lda=K, ldb=Nn if tb else K, ldc=Nn, ldr=Nn, trans_a=0, trans_b=tb,
                       epi=code, out_f32=0, batch=1)
        if dm is not None:
            g.drop = dm
        return g

    da, db_, dbias, dr = a.to(DEV), b.to(DEV), bias.to(DEV), r.to(DEV)
    dout = torch.empty((M, Nn), dtype=torch.bfloat16, device=DEV)
    dc2 = torch.empty_like(dout)
    g = args(da.data_ptr(), db_.data_ptr(), dout.data_ptr(), dc2.data_ptr(), dbias.data_ptr(), dr.data_ptr())
    assert dev_lib().cocodr_gemm(C.byref(g), stream_ptr()) == 0
    ha, hb, hr = _bits(a), _bits(b), _bits(r)
    hbias = bias.numpy()
    hout, hc2 = np.zeros((M, Nn), np.uint16), np.zeros((M, Nn), np.uint16)
    g2 = args(_hp(ha), _hp(hb), _hp(hout), _hp(hc2), _hp(hbias), _hp(hr))
    assert ref_twins.lib().cocodr_gemm_ref(C.byref(g2), None) == 0
    got, want = _f32(_bits(dout)), _f32(hout)
    assert np.abs(got - want).max() <= np.abs(want).max() * 2 ** -7
    if drop:  # dropped elements are exact: out == residual there, on the same positions
        rr = _f32(hr)
        assert np.array_equal(got == rr, want == rr) and 0.05 < (want == rr).mean() < 0.15
    if epi == "gelu":
        assert np.abs(_f32(_bits(dc2)) - _f32(hc2)).max() <= 2 ** -7 * 1.2


def test_ln_attention_losses_device_vs_twin():
    rng = np.random.Generator(np.random.PCG64(12))
    tw = ref_twins.lib()
    # LayerNorm
    M, H = 96, 256
    y = _bf(rng.standard_normal((M, H)) * 1.5 + 0.2)
    gam, bet = (torch.from_numpy(rng.standard_normal(H).astype(np.float32)) for _ in range(2))
    out, mean, rstd = ops.ln_fwd(y.to(DEV), gam.to(DEV), bet.to(DEV), 1e-12)
    hout, hmean, hrstd = np.zeros((M, H), np.uint16), np.zeros(M, np.float32), np.zeros(M, np.float32)
    assert tw.cocodr_ln_fwd_ref(_hp(_bits(y)), _hp(gam.numpy()), _hp(bet.numpy()), _hp(hout), _hp(hmean), _hp(hrstd), None, 0, M, H, 1e-12, None) == 0
    assert np.abs(_f32(_bits(out)) - _f32(hout)).max() <= np.abs(_f32(hout)).max() * 2 ** -7
    assert np.allclose(mean.cpu().numpy(), hmean, atol=1e-5) and np.allclose(rstd.cpu().numpy(), hrstd, rtol=1e-5)
    # attention forward
    B, L, heads = 3, 64, 2
    Hh = heads * 64
    qkv = _bf(rng.standard_normal((B * L, 3 * Hh)) * 0.8)
    mask = np.ones((B, L), np.int32)
    mask[1, 40:] = 0
    mask[2, 9:] = 0
    ctx, lse = ops.attn_fwd(qkv.to(DEV), torch.from_numpy(mask).to(DEV), B, L, heads)
    hctx, hlse = np.zeros((B * L, Hh), np.uint16), np.zeros((B, heads, L), np.float32)
    assert tw.cocodr_attn_fwd_ref(_hp(_bits(qkv)), _hp(mask), _hp(hctx), _hp(hlse), B, L, heads, None) == 0
    valid = np.repeat(mask.reshape(-1) != 0, Hh).reshape(B * L, Hh)
    assert np.abs(_f32(_bits(ctx)) - _f32(hctx))[valid].max() <= np.abs(_f32(hctx)).max() * 2 ** -6
    vl = np.broadcast_to((mask != 0)[:, None, :], (B, heads, L))
    assert np.allclose(lse.cpu().numpy()[vl], hlse[vl], rtol=1e-3, atol=2e-3)
    # contrastive loss + local gradient (2 ranks' shares of one gathered matrix)
    Mm, He, world = 32, 128, 2
    E = (rng.standard_normal((Mm, He)) * 0.3).astype(np.float32)
    for rank in range(world):
        m_local = Mm // world
        loss, rows, dE = ops.simce_fwd_bwd(torch.from_numpy(E).to(DEV), world, rank * m_local, m_local)
        hrows, hloss, hdE, ws = np.zeros(Mm, np.float32), np.zeros(1, np.float32), np.zeros((m_local, He), np.float32), np.zeros(Mm * Mm, np.float32)
        assert tw.cocodr_simce_fwd_bwd_ref(_hp(E), Mm, He, world, rank * m_local, m_local, _hp(hrows), _hp(hloss), _hp(hdE), _hp(ws), None) == 0
        assert abs(float(loss) - float(hloss[0])) < 1e-5 * abs(float(hloss[0]))
        np.testing.assert_allclose(rows.cpu().numpy(), hrows, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(dE.cpu().numpy(), hdE, rtol=1e-4, atol=1e-6)
    # triplet NLL
    Bt = 16
    q, a, b = ((rng.standard_normal((Bt, He)) * 0.2).astype(np.float32) for _ in range(3))
    w = rng.random(Bt).astype(np.float32)
    loss, rows, logits, dq, da, db = ops.triplet_nll_fwd_bwd(*(torch.from_numpy(x).to(DEV) for x in (q, a, b, w)))
    h = [np.zeros(Bt, np.float32), np.zeros((Bt, 2), np.float32), np.zeros(1, np.float32)] + [np.zeros((Bt, He), np.float32) for _ in range(3)]
    assert tw.cocodr_triplet_nll_fwd_bwd_ref(_hp(q), _hp(a), _hp(b), _hp(w), Bt, He, *[_hp(x) for x in h], None) == 0
    assert abs(float(loss) - float(h[2][0])) < 1e-5
    for got, ref in zip((rows, logits, dq, da, db), (h[0], h[1], h[3], h[4], h[5])):
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-6)


def test_search_and_merge_device_vs_twin():
    rng = np.random.Generator(np.random.PCG64(13))
    tw = ref_twins.lib()
    Nq, Np, H, k = 9, 5000, 64, 50
    Q = (rng.standard_normal((Nq, H)) / 8).astype(np.float32)
    P = (rng.standard_normal((Np, H)) / 8).astype(np.float32)
    P[77] = P[4000]  # an exact tie: bit-identical scores on both sides of the exact pipeline
    ops.score_set_mode(1)  # exact fp32 MFMA = the fmaf chain of the twin, bit for bit
    try:
        D, I = ops.score_topk(torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV), k, id_offset=5)
    finally:
        ops.score_set_mode(0)
    hD, hI = np.zeros((Nq, k), np.float32), np.zeros((Nq, k), np.int64)
    ws = np.zeros(Np * 16, np.uint8)
    assert tw.cocodr_score_topk_ref(_hp(Q), _hp(P), Nq, Np, H, k, 5, _hp(hD), _hp(hI), _hp(ws), ws.nbytes, None) == 0
    assert np.array_equal(I.cpu().numpy(), hI) and np.array_equal(D.cpu().numpy(), hD)
    # merge of four shards' lists
    cuts = [0, 900, 1000, 3500, 5000]
    Ds, Is = [], []
    for a, b in zip(cuts, cuts[1:]):
        d, i = ops.score_topk(torch.from_numpy(Q).to(DEV), torch.from_numpy(P[a:b]).to(DEV), k, 0)
        Ds.append(d)
        Is.append(i.to(torch.int32))
    Dw, Iw = torch.stack(Ds), torch.stack(Is)
    offs = torch.tensor(cuts[:-1], dtype=torch.int64, device=DEV)
    mD, mI = ops.topk_merge(Dw, Iw, offs, k)
    hDw, hIw, hoffs = Dw.cpu().numpy(), Iw.cpu().numpy(), offs.cpu().numpy()
    oD, oI = np.zeros((Nq, k), np.float32), np.zeros((Nq, k), np.int64)
    assert tw.cocodr_topk_merge_ref(_hp(hDw), _hp(hIw), _hp(hoffs), 4, Nq, k, Nq * k, _hp(oD), _hp(oI), k, None) == 0
    assert np.array_equal(mI.cpu().numpy(), oI) and np.array_equal(mD.cpu().numpy(), oD)


def test_backward_kernels_device_vs_twin():
    """cocodr_ln_bwd / cocodr_attn_bwd / cocodr_embed_ln_fwd / cocodr_embed_ln_bwd against their C twins, one argument tuple per pair
    (the backward kernels are the ones a maintainer most needs a host twin for)."""
    rng = np.random.Generator(np.random.PCG64(21))
    tw = ref_twins.lib()
    dl = dev_lib()
    sp = stream_ptr()

    def dev(x):
        return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)

    # ---- LayerNorm backward (mean / rstd from the forward twin)
    M, H = 160, 256
    y, dout = _bits(_bf(rng.standard_normal((M, H)) * 1.3 + 0.1)), _bits(_bf(rng.standard_normal((M, H))))
    gam, bet = rng.standard_normal(H).astype(np.float32), np.zeros(H, np.float32)
    o_, mean, rstd = np.zeros((M, H), np.uint16), np.zeros(M, np.float32), np.zeros(M, np.float32)
    assert tw.cocodr_ln_fwd_ref(_hp(y), _hp(gam), _hp(bet), _hp(o_), _hp(mean), _hp(rstd), None, 0, M, H, 1e-12, None) == 0
    host = [np.zeros((M, H), np.uint16)] + [np.zeros(H, np.float32) for _ in range(3)]
    assert tw.cocodr_ln_bwd_ref(_hp(dout), _hp(y), _hp(gam), _hp(mean), _hp(rstd), *[_hp(h) for h in host], None, M, H, None) == 0
    d_in = [dev(x.view(np.int16)).view(torch.bfloat16) for x in (dout, y)] + [dev(gam), dev(mean), dev(rstd)]
    d_out = [torch.empty((M, H), dtype=torch.bfloat16, device=DEV)] + [torch.empty(H, dtype=torch.float32, device=DEV) for _ in range(3)]
    part = torch.empty(int(dl.cocodr_ln_bwd_partial_floats(M, H)), dtype=torch.float32, device=DEV)
    assert dl.cocodr_ln_bwd(*[t.data_ptr() for t in d_in], *[t.data_ptr() for t in d_out], part.data_ptr(), M, H, sp) == 0
    assert np.abs(_f32(_bits(d_out[0])) - _f32(host[0])).max() <= np.abs(_f32(host[0])).max() * 2 ** -7
    for got, want in zip(d_out[1:], host[1:]):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-3, atol=2e-2)   # fp32 sums of M bf16-derived terms, another order
    # ---- attention backward
    B, L, heads = 2, 64, 2
    Hh = heads * 64
    qkv, dctx = _bits(_bf(rng.standard_normal((B * L, 3 * Hh)) * 0.7)), _bits(_bf(rng.standard_normal((B * L, Hh)) * 0.5))
    mask = np.ones((B, L), np.int32)
    mask[1, 37:] = 0
    hctx, hlse = np.zeros((B * L, Hh), np.uint16), np.zeros((B, heads, L), np.float32)
    assert tw.cocodr_attn_fwd_ref(_hp(qkv), _hp(mask), _hp(hctx), _hp(hlse), B, L, heads, None) == 0
    hdqkv, hqk = np.zeros((B * L, 3 * Hh), np.uint16), np.zeros((4 * B, 2 * Hh), np.float32)
    assert tw.cocodr_attn_bwd_ref(_hp(qkv), _hp(mask), _hp(hctx), _hp(dctx), _hp(hlse), _hp(hdqkv), _hp(hqk), B, L, heads, None) == 0
    dq_, dm_, dc_, dd_ = (dev(x.view(np.int16)).view(torch.bfloat16) if x.dtype == np.uint16 else dev(x) for x in (qkv, mask, hctx, dctx))
    ctx_d, lse_d = ops.attn_fwd(dq_, dm_, B, L, heads)
    ddqkv = torch.empty((B * L, 3 * Hh), dtype=torch.bfloat16, device=DEV)
    dqk = torch.empty((4 * B, 2 * Hh), dtype=torch.float32, device=DEV)
    assert dl.cocodr_attn_bwd(dq_.data_ptr(), dm_.data_ptr(), ctx_d.data_ptr(), dd_.data_ptr(), lse_d.data_ptr(), ddqkv.data_ptr(), dqk.data_ptr(),
                              B, L, heads, sp) == 0
    valid = np.repeat(mask.reshape(-1) != 0, 3 * Hh).reshape(B * L, 3 * Hh)
    got, want = _f32(_bits(ddqkv)), _f32(hdqkv)
    assert np.abs(got - want)[valid].max() <= np.abs(want).max() * 2 ** -5   # P recomputed from the bf16-rounded lse path: a few bf16 ulps
    np.testing.assert_allclose(dqk.cpu().numpy().reshape(B, 4, 2 * Hh).sum(1), hqk.reshape(B, 4, 2 * Hh).sum(1), rtol=2e-2, atol=2e-2)
    # ---- embeddings forward / backward
    Bq, Lq, He, V = 4, 32, 128, 300
    ids = rng.integers(0, V, (Bq, Lq)).astype(np.int32)
    word = (rng.standard_normal((V, He)) * 0.5).astype(np.float32)
    pos = (rng.standard_normal((64, He)) * 0.5).astype(np.float32)
    typ, g_, b_ = ((rng.standard_normal(He) * 0.5).astype(np.float32) for _ in range(3))
    hout, hmean, hrstd = np.zeros((Bq * Lq, He), np.uint16), np.zeros(Bq * Lq, np.float32), np.zeros(Bq * Lq, np.float32)
    assert tw.cocodr_embed_ln_fwd_ref(_hp(ids), _hp(word), _hp(pos), _hp(typ), _hp(g_), _hp(b_), _hp(hout), _hp(hmean), _hp(hrstd), Bq, Lq, He, V,
                                      1e-12, None) == 0
    t = [dev(x) for x in (ids, word, pos, typ, g_, b_)]
    dout_e = torch.empty((Bq * Lq, He), dtype=torch.bfloat16, device=DEV)
    dmean, drstd = torch.empty(Bq * Lq, dtype=torch.float32, device=DEV), torch.empty(Bq * Lq, dtype=torch.float32, device=DEV)
    assert dl.cocodr_embed_ln_fwd(*[x.data_ptr() for x in t], dout_e.data_ptr(), dmean.data_ptr(), drstd.data_ptr(), Bq, Lq, He, V, 1e-12, sp) == 0
    assert np.abs(_f32(_bits(dout_e)) - _f32(hout)).max() <= np.abs(_f32(hout)).max() * 2 ** -7
    assert np.allclose(dmean.cpu().numpy(), hmean, atol=1e-5) and np.allclose(drstd.cpu().numpy(), hrstd, rtol=1e-5)
    dy = _bits(_bf(rng.standard_normal((Bq * Lq, He))))
    hg = [np.zeros((V, He), np.float32), np.zeros((Lq, He), np.float32)] + [np.zeros(He, np.float32) for _ in range(3)]
    assert tw.cocodr_embed_ln_bwd_ref(_hp(dy), _hp(ids), _hp(word), _hp(pos), _hp(typ), _hp(g_), _hp(hmean), _hp(hrstd), *[_hp(x) for x in hg], None,
                                      Bq, Lq, He, V, None) == 0
    dg = [torch.zeros((V, He), dtype=torch.float32, device=DEV), torch.zeros((64, He), dtype=torch.float32, device=DEV)] + \
         [torch.empty(He, dtype=torch.float32, device=DEV) for _ in range(3)]
    part = torch.empty(int(dl.cocodr_embed_bwd_partial_floats(Lq, He)), dtype=torch.float32, device=DEV)
    assert dl.cocodr_embed_ln_bwd(dev(dy.view(np.int16)).view(torch.bfloat16).data_ptr(), t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                  t[3].data_ptr(), t[4].data_ptr(), dmean.data_ptr(), drstd.data_ptr(), *[x.data_ptr() for x in dg], part.data_ptr(),
                                  Bq, Lq, He, V, sp) == 0
    np.testing.assert_allclose(dg[0].cpu().numpy(), hg[0], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(dg[1].cpu().numpy()[:Lq], hg[1], rtol=2e-3, atol=2e-3)
    for got, want in zip(dg[2:], hg[2:]):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-3, atol=2e-2)
