"""The filtered search (coco-dr_amd/csrc/score_filter.h: thresholds from a passage sample, the score GEMM's epilogue keeps the
scores at or above them, selection from the candidate blocks, exhaustive pass over the rows handed back) against the exhaustive
search it replaces (COCODR_SCORE_NOFILTER=1: full score slab + radix select) - D and I must be IDENTICAL, ties included - and
against the numpy oracle.  The test hooks (environment, read per call) push rows through every hand-back route."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402,F401
from cocodr_amd import ops  # noqa: E402
import oracle as O  # noqa: E402

DEV = "cuda"
HOOKS = ("COCODR_SCORE_NOFILTER", "COCODR_SCORE_FILTER_FORCE", "COCODR_SCORE_FILTER_J", "COCODR_SCORE_FILTER_CAPT", "COCODR_SCORE_PBLK")


def search(Q, P, k, monkeypatch, id_offset=0, **env):
    for h in HOOKS:
        monkeypatch.delenv(h, raising=False)
    if "NOFILTER" not in env:  # the filtered route whatever the size (the plan keeps small searches off it: it pays from ~1e8 scores)
        env = dict(env, FILTER_FORCE=1)
    for name, v in env.items():
        monkeypatch.setenv("COCODR_SCORE_" + name, str(v))
    if "NOFILTER" not in env:
        assert ops.score_filter_plan(Q.shape[0], P.shape[0], Q.shape[1], k)["filtered"] == 1
    D, I = ops.score_topk(Q, P, k, id_offset=id_offset)
    torch.cuda.synchronize()
    for h in HOOKS:
        monkeypatch.delenv(h, raising=False)
    return D.cpu().numpy(), I.cpu().numpy()


def data(nq, npass, H, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    Q = (rng.standard_normal((nq, H)) / np.sqrt(H)).astype(np.float32)
    P = (rng.standard_normal((npass, H)) / np.sqrt(H)).astype(np.float32)
    return Q, P


def same(a, b):
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[0], b[0])


@pytest.mark.parametrize("nq,npass,H,k", [(300, 40000, 128, 100), (64, 100000, 1024, 1000), (1, 33000, 64, 1), (513, 65537, 64, 2048)])
def test_filtered_search_equals_exhaustive_search_and_oracle(nq, npass, H, k, monkeypatch):
    Q, P = data(nq, npass, H, npass + k)
    Qd, Pd = torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV)
    got = search(Qd, Pd, k, monkeypatch, id_offset=7_000_000_000)
    ref = search(Qd, Pd, k, monkeypatch, id_offset=7_000_000_000, NOFILTER=1)
    same(got, ref)
    if nq * npass <= 64 * 100000:
        Dr, Ir = O.score_topk(Q, P, k)
        np.testing.assert_allclose(got[0], Dr, rtol=1e-5, atol=1e-6)
        assert (got[1] - 7_000_000_000 == Ir).mean() > 0.995  # (fp32 round-off may swap near-ties against the numpy product)


@pytest.mark.parametrize("mode", [0, 2])
def test_small_ragged_searches_through_the_filter(mode, monkeypatch):
    """(forced onto the filtered route) Np not a multiple of the 256-column tile (zero padding columns must never become candidates - every
    true score here is negative), several passage column blocks, both 16-bit score modes."""
    Q, P = data(70, 5003, 64, 3)
    P -= 4.0 * Q[0] / np.sqrt((Q[0] ** 2).sum())  # row 0's scores ~ -4: far below the padding columns' zeros
    Qd, Pd = torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV)
    ops.score_set_mode(mode)
    try:
        ref = search(Qd, Pd, 10, monkeypatch, NOFILTER=1)
        same(search(Qd, Pd, 10, monkeypatch), ref)
        same(search(Qd, Pd, 10, monkeypatch, PBLK=512), ref)
    finally:
        ops.score_set_mode(0)
    assert ref[0][0].max() < 0 and ref[1].max() < 5003


def test_rows_handed_back_threshold_too_high(monkeypatch):
    """J = 1: the threshold is the sample's best score, almost no row finds k scores above it - every row takes the exhaustive pass."""
    Q, P = data(200, 40000, 128, 11)
    Qd, Pd = torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV)
    same(search(Qd, Pd, 100, monkeypatch, FILTER_J=1), search(Qd, Pd, 100, monkeypatch, NOFILTER=1))


def test_rows_handed_back_blocks_overflow(monkeypatch):
    """CAPT = 8: seven entries per (row, column tile) - blocks overflow on many rows, the others are answered from their candidates."""
    Q, P = data(200, 40000, 128, 12)
    Qd, Pd = torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV)
    same(search(Qd, Pd, 500, monkeypatch, FILTER_CAPT=8), search(Qd, Pd, 500, monkeypatch, NOFILTER=1))


def test_exact_ties_and_duplicates(monkeypatch):
    """A corpus of 400 distinct passages repeated 100 times: every score ties 100-fold, the k-th straddles a tie group - lower
    positions first, exactly as the exhaustive search (and faiss) order them; and integer-valued embeddings (exact scores)."""
    Q, base = data(50, 400, 128, 13)
    P = np.tile(base, (100, 1))
    Qd, Pd = torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV)
    got = search(Qd, Pd, 250, monkeypatch)
    same(got, search(Qd, Pd, 250, monkeypatch, NOFILTER=1))
    for r in range(0, 50, 7):  # a tie group is listed in ascending position
        d, i = got[0][r], got[1][r]
        for s in range(249):
            assert d[s] > d[s + 1] or (d[s] == d[s + 1] and i[s] < i[s + 1])
    rng = np.random.Generator(np.random.PCG64(14))
    Qi = rng.integers(-3, 4, (40, 64)).astype(np.float32)
    Pi = rng.integers(-3, 4, (50000, 64)).astype(np.float32)
    Qd, Pd = torch.from_numpy(Qi).to(DEV), torch.from_numpy(Pi).to(DEV)
    got = search(Qd, Pd, 300, monkeypatch)
    same(got, search(Qd, Pd, 300, monkeypatch, NOFILTER=1))
    Dr, Ir = O.score_topk(Qi, Pi, 300)
    np.testing.assert_array_equal(got[0], Dr)
    np.testing.assert_array_equal(got[1], Ir)


def test_corpus_sorted_by_relevance(monkeypatch):
    """An adversarial order for a sample: the passages sorted by their score against query 0 (best first), and a block of
    near-duplicates of query 1 stored back to back (one column tile holds most of its top k)."""
    Q, P = data(33, 60000, 64, 15)
    P = P[np.argsort(-(P @ Q[0]))]
    P[30000:30200] = Q[1] + 0.01 * P[30000:30200]
    Qd, Pd = torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV)
    same(search(Qd, Pd, 200, monkeypatch), search(Qd, Pd, 200, monkeypatch, NOFILTER=1))


def test_nan_query_row_and_constant_corpus(monkeypatch):
    Q, P = data(20, 40000, 64, 16)
    Q[3, 5] = np.nan
    Qd, Pd = torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV)
    got, ref = search(Qd, Pd, 50, monkeypatch), search(Qd, Pd, 50, monkeypatch, NOFILTER=1)
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(np.isnan(got[0]), np.isnan(ref[0]))
    np.testing.assert_array_equal(np.nan_to_num(got[0]), np.nan_to_num(ref[0]))
    Pc = np.ones((40000, 64), np.float32)  # every score of a row ties: positions 0 .. k-1
    got = search(Qd, torch.from_numpy(Pc).to(DEV), 50, monkeypatch)
    np.testing.assert_array_equal(got[1][0], np.arange(50))


def test_more_rows_handed_back_than_one_exhaustive_pass_holds(monkeypatch):
    """4 500 query rows x 125 000 passages: the slab area holds 4 096 rows of plain scores, so with every row handed back (J = 1)
    the exhaustive pass runs twice; and the default route at the same size."""
    Q, P = data(4500, 125000, 64, 17)
    Qd, Pd = torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV)
    ref = search(Qd, Pd, 100, monkeypatch, NOFILTER=1)
    same(search(Qd, Pd, 100, monkeypatch, FILTER_J=1), ref)
    same(search(Qd, Pd, 100, monkeypatch), ref)


def test_default_route_is_the_filter_and_hands_back_almost_nothing():
    """At one GPU's shard of config 5 (10 000 x 125 000 x 1 024, k = 1 000) the plan filters, the thresholds leave every row between
    k and the list's capacity in candidates: the count of rows handed back to the exhaustive pass (read from the workspace) is 0 or a
    handful; with embedding-shaped data of a common offset (LayerNorm-like rows) as well."""
    nq, npass, H, k = 10000, 125000, 1024, 1000
    plan = ops.score_filter_plan(nq, npass, H, k)
    assert plan["filtered"] == 1
    g = torch.Generator().manual_seed(3)
    for offset in (0.0, 0.05):
        Q = (torch.randn(nq, H, generator=g) / H ** 0.5 + offset).to(DEV)
        P = (torch.randn(npass, H, generator=g) / H ** 0.5 + offset).to(DEV)
        ws = torch.empty(ops.lib().cocodr_score_topk_workspace_bytes_dim(nq, npass, H, k), dtype=torch.uint8, device=DEV)
        D, I = ops.score_topk(Q, P, k, workspace=ws)
        torch.cuda.synchronize()
        handed_back = int(ws[plan["handed_back_count_offset"]:plan["handed_back_count_offset"] + 4].view(torch.int32).item())
        assert handed_back <= 10, handed_back
        rows = torch.randint(0, nq, (16,), generator=g)
        S = Q[rows.to(DEV)].double() @ P.double().T
        Dr, Ir = torch.topk(S, k, dim=1)
        got_i, ref_i = I[rows.to(DEV)].cpu().numpy(), Ir.cpu().numpy()  # (fp32 round-off reorders near-ties against the fp64 product)
        assert min(len(np.intersect1d(a, b)) for a, b in zip(got_i, ref_i)) >= 0.99 * k
        torch.testing.assert_close(D[rows.to(DEV)].double(), Dr, rtol=1e-5, atol=1e-6)
        del Q, P, ws, D, I, S


def test_one_million_passages_in_several_query_passes(monkeypatch):
    """3 000 queries x 1 000 000 passages: the candidate blocks of a pass (3 907 column tiles per row) fill the slab area after 2 560
    rows, so the filtered search runs in two query passes; thresholds come from 31 488 sampled passages."""
    plan = ops.score_filter_plan(3000, 1_000_000, 64, 100)  # (taken by default: 3e9 scores)
    assert plan["filtered"] == 1 and plan["rows_per_pass"] < 3000 and plan["sample_passages"] == 31488
    Q, P = data(3000, 1_000_000, 64, 21)
    Qd, Pd = torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV)
    same(search(Qd, Pd, 100, monkeypatch), search(Qd, Pd, 100, monkeypatch, NOFILTER=1))


def test_flat_ip_index_add_once_search_many_matches_search():
    """retrieval.FlatIPIndex (faiss.IndexFlatIP as ANCE/drivers/run_ann_data_gen.py:310-317,390 uses it): the second and later
    searches of a shape run cocodr_score_topk_resident on the passages' resident image - same D / I as a fresh search, bit for bit,
    on the filtered route (forced: small sizes) and the exhaustive one; add() and a new shape invalidate the image."""
    import os
    from cocodr_amd import retrieval
    g = torch.Generator().manual_seed(5)
    Q1 = (torch.randn(300, 256, generator=g) / 16).to(DEV)
    Q2 = (torch.randn(300, 256, generator=g) / 16).to(DEV)
    P = (torch.randn(40000, 256, generator=g) / 16).to(DEV)
    for force in ("1", None):
        if force:
            os.environ["COCODR_SCORE_FILTER_FORCE"] = force
        else:
            os.environ.pop("COCODR_SCORE_FILTER_FORCE", None)
        try:
            index = retrieval.FlatIPIndex(256)
            index.add(P[:25000])
            index.add(P[25000:])
            assert index.ntotal == 40000
            for Q in (Q1, Q2, Q1):   # first call builds the image, the next two reuse it
                D, I = index.search(Q, 100)
                Dr, Ir = retrieval.search(Q, P, 100)
                assert torch.equal(I, Ir) and torch.equal(D, Dr)
            D, I = index.search(Q1[:77], 50)           # another shape: rebuilt, not reused
            Dr, Ir = retrieval.search(Q1[:77], P, 50)
            assert torch.equal(I, Ir) and torch.equal(D, Dr)
            index.add(P[:1000] * 2.0)                  # new passages: the image is stale
            D, I = index.search(Q2, 100)
            Dr, Ir = retrieval.search(Q2, torch.cat([P, P[:1000] * 2.0]), 100)
            assert torch.equal(I, Ir) and torch.equal(D, Dr)
        finally:
            os.environ.pop("COCODR_SCORE_FILTER_FORCE", None)
