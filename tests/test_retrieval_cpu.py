"""Host-side retrieval logic (product) against the oracle restatement on seeded synthetic runs.  CPU only."""
import numpy as np
import torch

import cocodr_amd
from cocodr_amd import retrieval as R
import oracle as O


def test_shard_rule_and_merge_order():
    for n, w in ((10, 3), (7, 8), (64, 8)):
        for r in range(w):
            assert np.array_equal(R.shard_indices(n, r, w).numpy(), O.shard_indices(n, r, w))
        assert np.array_equal(R.merged_order(n, w).numpy(), O.merged_order(n, w))


def test_merge_topk_ties_and_padding():
    D = torch.tensor([[3.0, 1.0, -float("inf"), 3.0, 2.0, 1.0]])
    I = torch.tensor([[5, 7, -1, 2, 9, 4]])
    d, i = R.merge_topk(D, I, 4)
    assert i.tolist() == [[2, 5, 9, 4]] and d.tolist() == [[3.0, 3.0, 2.0, 1.0]]


def test_eval_dev_query_and_negatives_match_oracle():
    rng = np.random.Generator(np.random.PCG64(2))
    nq, npass, k = 40, 300, 50
    q2id = rng.permutation(1000)[:nq]
    p2id = rng.integers(0, 200, npass)  # duplicated pids: several vectors per document
    I = np.stack([rng.permutation(npass)[:k] for _ in range(nq)])
    qrels = {int(q): {int(p): int(rng.integers(1, 3)) for p in rng.integers(0, 200, 3)} for q in q2id[:35]}
    self_q = {int(q2id[0]): "a", int(q2id[1]): "b"}
    self_p = {int(p2id[I[0, 0]]): "a", int(p2id[I[1, 3]]): "b"}
    got = R.eval_dev_query(q2id, p2id, qrels, torch.from_numpy(I), 30, self_q, self_p)
    ref = O.eval_dev_query(q2id, p2id, qrels, I, 30, (self_q, self_p))
    assert got[2] == ref[2] == 35
    assert abs(got[0] - ref[0]) < 1e-12 and abs(got[1] - ref[1]) < 1e-12
    assert got[3] == ref[3]
    pos = {int(q): int(p2id[I[i, int(rng.integers(0, k))]]) for i, q in enumerate(q2id)}
    eff = [int(q) for q in q2id[::2]]
    n1, r1 = R.generate_negatives(q2id, p2id, pos, I, 7, eff)
    n2, r2 = O.generate_negatives(q2id, p2id, pos, I, 7, eff)
    assert n1 == n2 and np.allclose(r1, r2)


def test_mrr_matches_reference_golden():
    from conftest import load_golden
    g = load_golden("msmarco_mrr.npz")
    ranked = {q: [int(x) for x in row] for q, row in enumerate(g["ranked"])}
    relevant = {q: [int(x) for x in row if x >= 0] for q, row in enumerate(g["relevant"])}
    assert abs(R.mrr_at_10(relevant, ranked) - float(g["mrr10"])) < 1e-12
