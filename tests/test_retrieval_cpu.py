"""Host-side retrieval logic (product) against the oracle restatement on seeded synthetic runs.  CPU only."""
import numpy as np
import pytest
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
    n1, r1 = R.generate_negatives(q2id, p2id, pos, I, 7, eff, ann_measure_topk_mrr=True)
    n2, r2 = O.generate_negatives(q2id, p2id, pos, I, 7, eff)
    assert n1 == n2 and np.allclose(r1, r2)
    # the driver's default: the whole list in a shuffled order; Python's own random stream when none is injected
    import random
    random.seed(123)
    n3, r3 = R.generate_negatives(q2id, p2id, pos, I, 7, eff)
    random.seed(123)
    perms = []
    for _ in eff:
        o = list(range(k))
        random.shuffle(o)
        perms.append(o)
    n4, r4 = O.generate_negatives(q2id, p2id, pos, I, 7, eff, select_topk=False, permutations=perms)
    assert n3 == n4 and np.allclose(r3, r4) and n3 != n1


def _evaldev_inputs():
    from conftest import load_golden
    g = load_golden("evaldev_beir.npz")
    qrels = {}
    for q, p, r in g["qrels"]:
        qrels.setdefault(int(q), {})[int(p)] = int(r)
    off_q = {int(k): str(v) for k, v in zip(g["off_q"], g["off_q_char"])}
    off_p = {int(k): str(v) for k, v in zip(g["off_p"], g["off_p_char"])}
    pred = {}
    for q, p, sc in g["pred"].T:
        pred.setdefault(int(q), {})[int(p)] = int(sc)
    return g, qrels, off_q, off_p, pred


def test_full_eval_dev_query_matches_the_reference_function():
    """tests/golden/evaldev_beir.npz = outputs of the BEIR script's own EvalDevQuery (prediction dictionary, evaluated query
    count, both hole rates, MS MARCO MRR@10); the four trec_eval means come from the oracle's restatement (pytrec_eval is
    absent: unpinned, cross-checked on hand cases below)."""
    g, qrels, off_q, off_p, pred = _evaldev_inputs()
    topN = int(g["topN"])
    ref = O.eval_dev_query_beir(g["q2id"], g["p2id"], qrels, g["I"], topN, (off_q, off_p))
    got = R.EvalDevQuery(g["q2id"], g["p2id"], qrels, torch.from_numpy(g["I"]), topN, off_q, off_p)
    ndcg, cnt, Map, mrr, recall, hole, ms_mrr, ahole, result, prediction, mrrs, ndcgs = got
    for name, mine in (("oracle", ref["prediction"]), ("product", prediction)):
        assert {q: d for q, d in mine.items() if d} == pred, name  # the fixture stores (qid, pid, score) triples: empty dicts drop out
    assert cnt == ref["n_queries"] == int(g["n_queries"])
    for mine in (hole, ref["hole_rate"]):
        assert abs(mine - float(g["hole_rate"])) < 1e-12
    for mine in (ahole, ref["ahole_rate"]):
        assert abs(mine - float(g["ahole_rate"])) < 1e-12
    assert abs(ms_mrr["MRR @10"] - float(g["ms_mrr10"])) < 1e-12 and abs(ref["ms_mrr"] - float(g["ms_mrr10"])) < 1e-12
    assert ms_mrr["QueriesRanked"] == int(g["ms_ranked"])
    want = g["standin_means"]
    for a, b, c in zip((ndcg, Map, mrr, recall), (ref["ndcg"], ref["map"], ref["mrr"], ref["recall"]), want):
        assert abs(a - b) < 1e-12 and abs(a - c) < 1e-12
    assert len(mrrs) == len(ndcgs) == cnt and set(result) == set(prediction)
    # the 4-tuple front end agrees with the full one
    short = R.eval_dev_query(g["q2id"], g["p2id"], qrels, g["I"], topN, off_q, off_p)
    assert abs(short[0] - ndcg) < 1e-12 and abs(short[1] - mrr) < 1e-12 and short[2] == cnt


def test_trec_measures_on_hand_computed_cases():
    qrel = {1: 2, 2: 1, 3: 0, 4: 1}  # three relevant documents
    ranked = [9, 1, 3, 4, 8, 2]
    # relevant at ranks 2, 4, 6 -> AP@10 = (1/2 + 2/4 + 3/6) / 3 = 0.5 ; recall@5 = 2/3 ; recip_rank = 1/2
    for mod in (R, O):
        m = (mod.map_at_10 if mod is R else mod.map_cut)(ranked, qrel)
        assert abs(m - 0.5) < 1e-12
        assert abs(mod.recall_at(ranked, qrel, 5) - 2.0 / 3.0) < 1e-12 and mod.recall_at(ranked, qrel, 1) == 0.0
    assert abs(O.recip_rank(ranked, qrel) - 0.5) < 1e-12
    # map_cut_10 ignores relevant documents ranked beyond 10 but still divides by all relevant ones
    long = list(range(100, 110)) + [1]
    assert R.map_at_10(long, qrel) == 0.0 and O.map_cut(long, qrel) == 0.0
    assert R.map_at_10([], {}) == 0.0


def test_hard_negative_selection_matches_the_reference_function():
    """tests/golden/hard_negatives.npz = outputs of the ANCE driver's own GenerateNegativePassaageID for both branches (the
    shuffled one driven with recorded permutations)."""
    from conftest import load_golden
    g = load_golden("hard_negatives.npz")
    pos = {int(q): int(p) for q, p in g["pos"]}
    eff = [int(x) for x in g["eff"]]
    n_neg = int(g["negative_sample"])
    perms = [list(map(int, p)) for p in g["perms"]]

    def want(tag):
        return {int(q): [int(x) for x in row if x >= 0] for q, row in zip(g[f"{tag}_qids"], g[f"{tag}_negs"])}

    o_top, o_rr = O.generate_negatives(g["q2id"], g["p2id"], pos, g["I"], n_neg, eff)
    r_top, r_rr = R.generate_negatives(g["q2id"], g["p2id"], pos, g["I"], n_neg, eff, ann_measure_topk_mrr=True)
    assert o_top == r_top == want("topk") and np.array_equal(o_rr, g["topk_rr"]) and np.array_equal(r_rr, g["topk_rr"])
    o_sh, o_rr = O.generate_negatives(g["q2id"], g["p2id"], pos, g["I"], n_neg, eff, select_topk=False, permutations=perms)
    it = iter(perms)

    def replay(lst):
        lst[:] = next(it)

    r_sh, r_rr = R.generate_negatives(g["q2id"], g["p2id"], pos, g["I"], n_neg, eff, shuffle=replay)
    assert o_sh == r_sh == want("shuffle") and np.array_equal(o_rr, g["shuffle_rr"]) and np.array_equal(r_rr, g["shuffle_rr"])
    assert int(g["shuffle_nperm"]) == len(want("shuffle"))  # one shuffle per effective query


def test_mrr_matches_reference_golden():
    from conftest import load_golden
    g = load_golden("msmarco_mrr.npz")
    ranked = {q: [int(x) for x in row] for q, row in enumerate(g["ranked"])}
    relevant = {q: [int(x) for x in row if x >= 0] for q, row in enumerate(g["relevant"])}
    assert abs(R.mrr_at_10(relevant, ranked) - float(g["mrr10"])) < 1e-12


def test_flat_ip_index_refuses_what_it_cannot_hold_without_touching_the_gpu():
    """retrieval.FlatIPIndex (faiss.IndexFlatIP's add / search, ANCE/drivers/run_ann_data_gen.py:310-317): argument errors are host
    logic - wrong width, host tensors (there is no CPU fallback), a search of the empty index."""
    import torch
    from cocodr_amd.retrieval import FlatIPIndex
    index = FlatIPIndex(8)
    assert index.ntotal == 0
    with pytest.raises(ValueError):
        index.add(torch.zeros(4, 7))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        index.add(torch.zeros(4, 8))
    with pytest.raises(RuntimeError, match="empty"):
        index.search(torch.zeros(2, 8), 1)
    index.reset()
    assert index.ntotal == 0


def test_pmc_traffic_summary_counts_every_gemm_kernel_family(tmp_path):
    """tools/pmc_traffic.py feeds bench.py's roofline.traffic: its kernel filter must cover all three GEMM translation units (the
    hand-scheduled walk kernels were missing from it for part of round 6 - the figure silently covered 36 of 784 launches)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = ["void cocodr_gemm_v2::gemm_glds_kernel<128, 64>(cocodr_gemm_args, int, int)", "void cocodr_gemm_pp::gemm_pp_kernel<2, 0>(X)",
             "void cocodr_gemm_a4::gemm_a4_walk_kernel<0, 1, false, false, false, false>(cocodr_gemm_a4::A4Multi)", "void (anonymous namespace)::ln_fwd_kernel<3, true>(X)"]
    for counter, path in (("FETCH_SIZE", tmp_path / "f.csv"), ("WRITE_SIZE", tmp_path / "w.csv")):
        with open(path, "w") as f:
            f.write("Counter_Name,Kernel_Name,Counter_Value\n")
            for n in names:
                f.write(f'{counter},"{n}",1024\n')
    out = tmp_path / "o.json"
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), str(tmp_path / "f.csv"), str(tmp_path / "w.csv"), str(out), "cmd", "abc"],
                   check=True, capture_output=True)
    d = json.load(open(out))
    assert d["launches"] == 3 and d["commit"] == "abc"
    assert abs(d["hbm_bytes_per_launch"] - (2 * 1024 + 1024) * 1024) < 1e-6
