"""cocodr_topk_merge (the native k-way merge of per-shard top-k lists) against the oracle's lexicographic merge:
random shards, exact score ties across and within shards, short / empty shards, k_out < and > the candidates of one shard.
Bit-exact: scores are copied, positions are integers.  Reference semantics: one IndexFlatIP search over the rank-major
concatenation of the shards (ANCE/utils/util.py:117-155 + evaluate/evaluation/evaluate_beir.py:200-224)."""
import numpy as np
import pytest
import torch

import cocodr_amd  # noqa: F401
from cocodr_amd import ops
from cocodr_amd import retrieval as R
import oracle as O  # checker

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lists(rng, W, Nq, k, n_per_shard, tie_levels=None):
    """per-shard sorted top-k lists as cocodr_score_topk leaves them: (score desc, local position asc), (-inf, -1) padding"""
    D = np.full((W, Nq, k), -np.inf, np.float32)
    I = np.full((W, Nq, k), -1, np.int32)
    for w in range(W):
        n = n_per_shard[w]
        for q in range(Nq):
            m = min(k, n)
            if m == 0:
                continue
            pos = rng.permutation(n)[:m]
            sc = rng.standard_normal(m).astype(np.float32)
            if tie_levels:
                sc = rng.integers(0, tie_levels, m).astype(np.float32)
            order = np.lexsort((pos, -sc.astype(np.float64)))
            D[w, q, :m], I[w, q, :m] = sc[order], pos[order]
    return D, I


def _check(W, Nq, k, k_out, n_per_shard, seed, tie_levels=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    D, I = _lists(rng, W, Nq, k, n_per_shard, tie_levels)
    offs = np.concatenate([[0], np.cumsum(n_per_shard)[:-1]]).astype(np.int64)
    Dm, Im = ops.topk_merge(torch.from_numpy(D).to(DEV), torch.from_numpy(I).to(DEV), torch.from_numpy(offs).to(DEV), k_out)
    Ig = [np.where(I[w] >= 0, I[w].astype(np.int64) + offs[w], -1) for w in range(W)]
    Dr, Ir = O.merge_topk([D[w] for w in range(W)], Ig, k_out)
    Dr = np.where(Ir >= 0, Dr, -np.inf)
    assert np.array_equal(Im.cpu().numpy(), Ir), (W, Nq, k, k_out)
    assert np.array_equal(Dm.cpu().numpy(), Dr)


def test_merge_random_scores_config5_shape():
    _check(8, 37, 1000, 1000, [125000] * 8, 1)          # BASELINE configs[4]: 8 shards, k = 1000


def test_merge_exact_ties_prefer_lower_shard_then_lower_position():
    _check(4, 25, 64, 64, [500, 300, 800, 100], 2, tie_levels=5)
    _check(8, 9, 200, 150, [1000] * 8, 3, tie_levels=2)   # crowded ties, k_out < k


def test_merge_short_and_empty_shards_and_padding():
    _check(3, 11, 50, 120, [7, 0, 20], 4)                 # fewer candidates than k_out: (-inf, -1) padding
    _check(2, 5, 16, 32, [40, 3], 5)
    _check(1, 6, 10, 10, [100], 6)                        # W = 1: identity


def test_merge_equals_one_search_over_the_whole_corpus():
    """split a corpus in four unequal shards, search each natively, merge natively == one native search (bit for bit)"""
    g = torch.Generator().manual_seed(0)
    Q = torch.randn(300, 256, generator=g).to(DEV) / 16
    P = torch.randn(40000, 256, generator=g).to(DEV) / 16
    P[1234] = P[30001]  # an exact cross-shard tie
    cuts = [0, 9000, 9100, 25000, 40000]
    D, I = R.search(Q, P, 100)
    Ds, Is = [], []
    for a, b in zip(cuts, cuts[1:]):
        d, i = ops.score_topk(Q, P[a:b].contiguous(), 100, 0)
        Ds.append(d)
        Is.append(i.to(torch.int32))
    offs = torch.tensor(cuts[:-1], dtype=torch.int64, device=DEV)
    Dm, Im = R.merge_shard_lists(torch.stack(Ds), torch.stack(Is), offs, 100)
    assert torch.equal(Im, I) and torch.equal(Dm, D)


def test_merge_rejects_bad_arguments():
    D = torch.zeros((2, 3, 4), device=DEV)
    I = torch.zeros((2, 3, 4), dtype=torch.int32, device=DEV)
    offs = torch.zeros(2, dtype=torch.int64, device=DEV)
    with pytest.raises(Exception):
        ops.topk_merge(D, I, offs, 9)            # k_out > W * k
    with pytest.raises(ValueError):
        ops.topk_merge(D, I.to(torch.int64), offs, 4)
    with pytest.raises(ValueError):
        ops.topk_merge(D.cpu(), I.cpu(), offs.cpu(), 4)  # no CPU path


def test_a_real_candidate_scoring_minus_infinity_is_a_candidate():
    """Validity is the position (I < 0 = empty slot), never the score: a passage whose score is -inf keeps its rank behind every
    finite score and ahead of the padding (the C twin `cocodr_topk_merge_ref` has always read it that way)."""
    D = np.full((2, 1, 4), -np.inf, np.float32)
    I = np.full((2, 1, 4), -1, np.int32)
    D[0, 0, :3], I[0, 0, :3] = [2.0, 1.0, -np.inf], [5, 1, 3]       # three candidates, the last one scores -inf
    D[1, 0, :2], I[1, 0, :2] = [1.5, -np.inf], [0, 2]
    offs = np.array([0, 10], np.int64)
    Dm, Im = ops.topk_merge(torch.from_numpy(D).to(DEV), torch.from_numpy(I).to(DEV), torch.from_numpy(offs).to(DEV), 7)
    assert Im.cpu().numpy()[0].tolist() == [5, 10, 1, 3, 12, -1, -1]
    assert Dm.cpu().numpy()[0, :3].tolist() == [2.0, 1.5, 1.0] and np.isneginf(Dm.cpu().numpy()[0, 3:]).all()
