"""Parity of the loss and search kernels against the CPU oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402
from cocodr_amd import ops  # noqa: E402
import oracle as O  # noqa: E402  (checker only)

DEV = "cuda"


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


@pytest.fixture(params=[0, 1], ids=["split16", "exact32"])
def score_mode(request):
    """Both score pipelines (include/cocodr.h): split precision on the 16-bit matrix pipe (default) and exact fp32 MFMA."""
    ops.score_set_mode(request.param)
    yield request.param
    ops.score_set_mode(0)


def test_simce_matches_reference_goldens(golden_loss):
    g = golden_loss
    for M in (8, 16, 64):
        E, W = g[f"E_{M}"], int(g[f"W_{M}"])
        loss, rows, dE = ops.simce_fwd_bwd(t(E), world=W)
        # fp32 end to end: tolerance = fp32 round-off of a length-H dot product and exp/log
        np.testing.assert_allclose(rows.cpu().numpy(), g[f"rows_{M}"], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(dE.cpu().numpy(), g[f"dE_{M}"], rtol=1e-4, atol=2e-6)
        assert abs(float(loss) - float(g[f"rows_{M}"].mean())) < 2e-5


@pytest.mark.parametrize("M,H,W", [(6, 128, 1), (64, 768, 1), (64, 1024, 2), (60, 768, 2), (32, 100, 4), (256, 768, 4), (2048, 768, 8)])
def test_simce_local_rows_vs_oracle(M, H, W):
    rng = np.random.Generator(np.random.PCG64(M))
    E = (rng.standard_normal((M, H)) * (6.0 / np.sqrt(H))).astype(np.float32)
    m = M // W
    rank = W - 1
    loss, rows, dE = ops.simce_fwd_bwd(t(E), world=W, row0=rank * m, m_local=m)
    E64 = E.astype(np.float64)
    ref_rows = O.contrastive_loss(E64.copy(), W)
    ref_loss, _ = O.contrastive_loss_grad(E64.copy(), W)
    ref_dE = O.contrastive_local_grad(E64.copy(), W, rank)
    np.testing.assert_allclose(rows.cpu().numpy(), ref_rows, rtol=1e-4, atol=1e-4)
    assert abs(float(loss) - ref_loss) < 1e-4
    np.testing.assert_allclose(dE.cpu().numpy(), ref_dE, rtol=2e-3, atol=2e-6)


def test_triplet_matches_reference_golden(golden_ance):
    g = golden_ance
    q, a, b, w = g["q_emb"], g["a_emb"], g["b_emb"], g["weights"]
    loss, rows, logits, dq, da, db = ops.triplet_nll_fwd_bwd(t(q), t(a), t(b), t(w))
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=1e-5)
    assert abs(float(loss) - float(g["loss"])) < 1e-4
    rl, rq, ra, rb = O.triplet_nll_grad(q.astype(np.float64), a.astype(np.float64), b.astype(np.float64), w)
    # logits are ~130 with gaps ~2: fp32 round-off of the logits (1e-5 abs) moves softmax by ~1e-5 relative,
    # and dq = c1*(b - a) cancels element-wise, so compare in relative L2 (tolerance 1e-4)
    for got, ref in ((dq, rq), (da, ra), (db, rb)):
        got = got.cpu().numpy().astype(np.float64)
        assert np.linalg.norm(got - ref) <= 1e-4 * np.linalg.norm(ref)
    loss2, *_ = ops.triplet_nll_fwd_bwd(t(q), t(a), t(b), None)
    assert abs(float(loss2) - O.triplet_nll_grad(q, a, b)[0]) < 1e-4


@pytest.mark.parametrize("Nq,Np,H,k", [(5, 101, 16, 10), (130, 5000, 768, 100), (64, 40000, 1024, 1000), (3, 50, 64, 80)])
def test_score_topk_vs_oracle(Nq, Np, H, k, score_mode):
    rng = np.random.Generator(np.random.PCG64(Np))
    Q = (rng.standard_normal((Nq, H)) / np.sqrt(H)).astype(np.float32)
    P = (rng.standard_normal((Np, H)) / np.sqrt(H)).astype(np.float32)
    D, I = ops.score_topk(t(Q), t(P), k, id_offset=7)
    Dr, Ir = O.score_topk(Q, P, k)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    kk = min(k, Np)
    np.testing.assert_allclose(D[:, :kk], Dr[:, :kk], rtol=1e-5, atol=1e-6)
    # ids must match wherever neighbouring scores are separated by more than fp32 round-off
    gap_ok = np.ones((Nq, kk), bool)
    d = np.abs(np.diff(Dr[:, :kk], axis=1)) > 1e-6
    gap_ok[:, 1:] &= d
    gap_ok[:, :-1] &= d
    assert np.array_equal(I[:, :kk][gap_ok], (Ir[:, :kk] + 7)[gap_ok])
    assert all(len(set(r)) == kk for r in I[:, :kk])  # no position twice
    # the SET of returned positions is the oracle's on every row whose k-th and (k+1)-th best scores are separated by more
    # than fp32 round-off (near-ties inside the list may permute, near-ties at the cut may swap one member)
    if Np > kk:
        D2, _ = O.score_topk(Q, P, kk + 1)
        clear_cut = (D2[:, kk - 1] - D2[:, kk]) > 1e-6
    else:
        clear_cut = np.ones(Nq, bool)
    assert clear_cut.sum() >= 0.9 * Nq
    assert np.array_equal(np.sort(I[:, :kk], 1)[clear_cut], np.sort(Ir[:, :kk] + 7, 1)[clear_cut])
    if k > Np:
        assert (I[:, Np:] == -1).all() and np.isneginf(D[:, Np:]).all()


def test_score_topk_exact_ties_prefer_lower_position(score_mode):
    # small-integer vectors: every score is exact in fp32, so ties are exact and the order is fully determined
    rng = np.random.Generator(np.random.PCG64(1))
    Q = rng.integers(-2, 3, (7, 32)).astype(np.float32)
    P = rng.integers(-2, 3, (3000, 32)).astype(np.float32)
    P[100:200] = P[0]  # 101 exact duplicates
    D, I = ops.score_topk(t(Q), t(P), 50)
    Dr, Ir = O.score_topk(Q, P, 50)
    assert np.array_equal(I.cpu().numpy(), Ir) and np.array_equal(D.cpu().numpy(), Dr)


@pytest.mark.parametrize("mode", ["narrow", "ties"])
def test_score_topk_streams_the_row_when_the_cut_bin_is_crowded(mode, score_mode):
    """Both selection paths give the same result: scores packed into one exponent / a couple of mantissa steps (or
    thousands of exact ties at the cut) overflow the LDS candidate list, so every radix pass streams the row."""
    rng = np.random.Generator(np.random.PCG64(5))
    Np, k = 20000, 300
    if mode == "narrow":  # integer-valued, exact in fp32: scores 1024 + {0..255} -> one 11-bit key bin holds them all
        Q = np.zeros((3, 8), np.float32); Q[:, 0] = 1.0
        P = np.zeros((Np, 8), np.float32); P[:, 0] = 1024 + rng.integers(0, 256, Np)
    else:  # the k-th score is shared by 5000 passages
        Q = np.ones((3, 4), np.float32)
        P = rng.integers(0, 4, (Np, 4)).astype(np.float32)
        P[5000:10000] = 2.0
    D, I = ops.score_topk(t(Q), t(P), k)
    Dr, Ir = O.score_topk(Q, P, k)
    assert np.array_equal(D.cpu().numpy(), Dr) and np.array_equal(I.cpu().numpy(), Ir)


def test_score_topk_degenerate_sizes(score_mode):
    """One query, one passage, k = 1; and k far above the corpus size."""
    Q = np.array([[1.0, 2.0, 3.0, 4.0]], np.float32)
    P = np.array([[0.5, 0.5, 0.5, 0.5]], np.float32)
    D, I = ops.score_topk(t(Q), t(P), 1)
    assert I.cpu().tolist() == [[0]] and abs(float(D[0, 0]) - 5.0) < 1e-6
    D, I = ops.score_topk(t(Q), t(np.repeat(P, 3, 0) * np.array([[1.0], [3.0], [2.0]], np.float32)), 7, id_offset=100)
    assert I.cpu().tolist() == [[101, 102, 100, -1, -1, -1, -1]]
    assert np.isneginf(D.cpu().numpy()[0, 3:]).all()


def test_split_precision_scores_are_at_least_as_accurate_as_an_fp32_dot_product():
    """The claim the default pipeline rests on: against an fp64 reference the split-precision scores (two IEEE halves per
    operand, three exact partial products, fp32 accumulation) err no more than the exact-fp32 pipeline's fmaf chain does.
    Also: a query chunk boundary (Nq above the chunk size) and a padded passage count."""
    rng = np.random.Generator(np.random.PCG64(3))
    H, Nq, Np, k = 1024, 300, 30001, 200
    base = rng.standard_normal((Np, H)) * 3 + 0.5
    P = ((base - base.mean(1, keepdims=True)) / base.std(1, keepdims=True) * 0.2).astype(np.float32)  # LayerNorm-shaped rows
    Q = (rng.standard_normal((Nq, H)) * np.exp(rng.standard_normal((1, H)))).astype(np.float32) * 0.05  # components over several binades
    ref = Q.astype(np.float64) @ P.astype(np.float64).T
    errs = {}
    try:
        for mode in (0, 1):
            ops.score_set_mode(mode)
            D, I = ops.score_topk(t(Q), t(P), k)
            D, I = D.cpu().numpy().astype(np.float64), I.cpu().numpy()
            want = np.take_along_axis(ref, I, 1)
            errs[mode] = float(np.abs(D - want).max() / np.abs(ref).max())
            # and the set is the true top-k wherever the cut is clear of round-off
            order = np.argsort(-ref, 1, kind="stable")[:, :k + 1]
            kth, nxt = np.take_along_axis(ref, order[:, k - 1:k], 1)[:, 0], np.take_along_axis(ref, order[:, k:k + 1], 1)[:, 0]
            clear = (kth - nxt) > 4e-6 * np.abs(ref).max()
            assert clear.sum() > Nq // 2
            for q in np.nonzero(clear)[0]:
                assert set(I[q]) == set(order[q, :k]), (mode, q)
    finally:
        ops.score_set_mode(0)
    assert errs[0] < 2e-6 and errs[0] <= 1.5 * errs[1], errs


def test_split_precision_search_in_passage_column_blocks(monkeypatch):
    """Shards whose half operands exceed the GEMM's 32-bit operand offsets are scored in column blocks; forced small here."""
    rng = np.random.Generator(np.random.PCG64(9))
    Q = (rng.standard_normal((70, 256)) / 16).astype(np.float32)
    P = (rng.standard_normal((3000, 256)) / 16).astype(np.float32)
    ops.score_set_mode(0)
    D0, I0 = ops.score_topk(t(Q), t(P), 40)
    monkeypatch.setenv("COCODR_SCORE_PBLK", "512")
    D1, I1 = ops.score_topk(t(Q), t(P), 40)
    assert torch.equal(D0, D1) and torch.equal(I0, I1)


@pytest.mark.parametrize("G,D", [(1, 1000), (4, 21_257_216), (8, 4096), (12, 300_001), (16, 1 << 20), (33, 70_000), (64, 10_000)])
def test_gram_of_a_wide_matrix_matches_fp64(G, D):
    """iDRO's `all_grads @ all_grads.T` (ANCE/model/dro_loss.py:236-238) as one streaming pass: against a float64 product,
    symmetric, deterministic (two runs bit-identical); (4, 21 M) is the BERT-base size (layers 9-11, four groups)."""
    g = torch.Generator(device="cpu").manual_seed(G * 7 + D % 1000)
    a = torch.randn(G, D, generator=g, dtype=torch.float32)
    a[:, D // 3] *= 50.0
    dev = a.to("cuda")
    out = ops.gram(dev)
    ref = (a.double() @ a.double().T)
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    assert torch.equal(out, out.T) and torch.equal(out, ops.gram(dev))
    with pytest.raises(ValueError):
        ops.gram(dev.t())  # rows must be contiguous
