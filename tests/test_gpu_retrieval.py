"""Search + metric pipeline on the GPU against the oracle (planted-positive synthetic corpus), and a
full-size ANCE-shaped triplet step on BERT-large (config 4 shapes) for robustness."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402
from cocodr_amd import ops, retrieval as R  # noqa: E402
from cocodr_amd.modeling import BertDotNLL, CocoBertConfig  # noqa: E402
import oracle as O  # noqa: E402

DEV = "cuda"


def planted(nq, npass, H, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    Q = (rng.standard_normal((nq, H)) / np.sqrt(H)).astype(np.float32)
    P = (rng.standard_normal((npass, H)) / np.sqrt(H)).astype(np.float32)
    pos = rng.permutation(npass)[:nq]
    P[pos] = Q + (rng.standard_normal((nq, H)) * 0.1 / np.sqrt(H)).astype(np.float32) * 4
    return Q, P, pos


@pytest.mark.parametrize("nq,npass,H,k", [(200, 20000, 768, 100), (64, 100000, 1024, 1000)])
def test_search_ndcg_matches_oracle(nq, npass, H, k):
    Q, P, pos = planted(nq, npass, H, npass)
    D, I = R.search(torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV), k)
    Dr, Ir = O.score_topk(Q, P, k)
    q2id = np.arange(nq) + 5000
    p2id = np.arange(npass) * 3 + 1
    qrels = {int(q2id[i]): {int(p2id[pos[i]]): 1} for i in range(nq)}
    ndcg, mrr, n, _ = R.eval_dev_query(q2id, p2id, qrels, I, k)
    ndcg_r, mrr_r, n_r, _ = O.eval_dev_query(q2id, p2id, qrels, Ir, k)
    assert n == n_r == nq
    assert abs(ndcg - ndcg_r) < 1e-3 and abs(mrr - mrr_r) < 1e-3  # north_star: nDCG@10 within 1e-3
    assert 0.2 < ndcg <= 1.0
    np.testing.assert_allclose(D.cpu().numpy(), Dr, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", ["gaussian", "layernorm"])
def test_half_precision_score_mode_is_within_the_stated_tolerances(shape):
    """cocodr_score_set_mode(2) (opt-in): ONE product of the operands rounded to IEEE half, fp32 accumulation.  Against the fp32
    oracle on embedding-shaped data (Gaussian, and LayerNorm-shaped rows with a common offset - what last-layer [CLS] rows look
    like): scores within 1e-4 |q||p| (SURVEY 8d's id tolerance; typical error ~1e-5), the id SETS agree wherever the k-th gap is
    larger than that, nDCG@10 / MRR within 1e-3 (north_star), identical passages still tie bit for bit, and the mode is never
    selected by itself."""
    nq, npass, H, k = 128, 60000, 1024, 100
    Q, P, pos = planted(nq, npass, H, 77)
    if shape == "layernorm":
        rng = np.random.Generator(np.random.PCG64(5))
        off = rng.standard_normal(H).astype(np.float32) * 0.05
        Q, P = Q + off, P + off
    P[123] = P[45678]  # an exact duplicate
    Qd, Pd = torch.from_numpy(Q).to(DEV), torch.from_numpy(P).to(DEV)
    D0, I0 = R.search(Qd, Pd, k)
    ops.score_set_mode(2)
    try:
        D2, I2 = R.search(Qd, Pd, k)
        dup = ops.score_topk(Pd[123:124].contiguous(), Pd[[123, 45678, 7]].contiguous(), 3)
    finally:
        ops.score_set_mode(0)
    Dr, Ir = O.score_topk(Q, P, k)
    scale = np.linalg.norm(Q, axis=1)[:, None] * float(np.linalg.norm(P, axis=1).max())
    S2 = D2.cpu().numpy()
    exact_of_returned = np.einsum("qh,qkh->qk", Q.astype(np.float64), P[I2.cpu().numpy()].astype(np.float64))
    err = np.abs(S2 - exact_of_returned) / scale
    assert err.max() < 1e-4 and np.median(err) < 2e-5, (err.max(), np.median(err))
    # ids: the same SET wherever the cut between rank k and k + 1 is clearer than the tolerance
    full = Q.astype(np.float64) @ P.astype(np.float64).T
    srt = -np.sort(-full, axis=1)
    clear = (srt[:, k - 1] - srt[:, k]) > 2e-4 * scale[:, 0]
    same = np.array([set(a) == set(b) for a, b in zip(I2.cpu().numpy(), Ir)])
    assert same[clear].all() and (shape != "gaussian" or clear.sum() >= 5)  # (at rank 100 of 60 000 most cuts are finer than the tolerance)
    q2id, p2id = np.arange(nq) + 5000, np.arange(npass) * 3 + 1
    qrels = {int(q2id[i]): {int(p2id[pos[i]]): 1} for i in range(nq)}
    n2, m2, _, _ = R.eval_dev_query(q2id, p2id, qrels, I2, k)
    nr, mr, _, _ = O.eval_dev_query(q2id, p2id, qrels, Ir, k)
    assert abs(n2 - nr) < 1e-3 and abs(m2 - mr) < 1e-3
    assert float(dup[0][0, 0]) == float(dup[0][0, 1]) and dup[1][0, :2].tolist() == [0, 1]   # duplicates tie, lower position first
    np.testing.assert_allclose(D0.cpu().numpy(), Dr, rtol=1e-5, atol=1e-6 * float(np.abs(Dr).max()))  # (the default pipeline is untouched)


def test_search_linearity_and_shard_merge_property_at_scale():
    """No oracle at this size in seconds: (i) scaling Q by 2 doubles D and keeps I; (ii) searching two corpus halves
    and merging equals searching the whole corpus."""
    g = torch.Generator().manual_seed(0)
    Q = torch.randn(512, 1024, generator=g).to(DEV) / 32
    P = torch.randn(300000, 1024, generator=g).to(DEV) / 32
    D, I = R.search(Q, P, 200)
    D2, I2 = R.search(2 * Q, P, 200)
    assert torch.equal(I, I2) and torch.allclose(D2, 2 * D, rtol=1e-6)
    h = 150000
    Da, Ia = ops.score_topk(Q, P[:h].contiguous(), 200, 0)
    Db, Ib = ops.score_topk(Q, P[h:].contiguous(), 200, h)
    Dm, Im = R.merge_topk(torch.cat([Da, Db], 1), torch.cat([Ia, Ib], 1), 200)
    assert torch.equal(Im, I) and torch.equal(Dm, D)
    assert bool((D[:, :-1] >= D[:, 1:]).all())  # sorted


def test_bert_large_triplet_step_config4_shapes():
    torch.manual_seed(0)
    model = BertDotNLL(CocoBertConfig.large(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)).to(DEV)
    B = 32
    g = torch.Generator().manual_seed(1)
    q = torch.randint(1000, 30522, (B, 64), generator=g).to(DEV)
    a = torch.randint(1000, 30522, (B, 128), generator=g).to(DEV)
    b = torch.randint(1000, 30522, (B, 128), generator=g).to(DEV)
    qm = torch.ones_like(q)
    am = torch.ones_like(a)
    am[:, 100:] = 0
    loss, acc, logits = model(q, qm, a, am, b, am.clone())
    loss.backward()
    assert torch.isfinite(loss) and logits.shape == (B, 2) and acc.shape == (B,)
    gd, gn = model.bert.flat_decay.grad, model.bert.flat_nodecay.grad
    assert torch.isfinite(gd).all() and torch.isfinite(gn).all() and float(gd.abs().sum()) > 0
    # loss agrees with the fp32 formula on the embeddings the model produced
    with torch.no_grad():
        qe, ae, be = model.query_emb(q, qm), model.body_emb(a, am), model.body_emb(b, am)
        ref = -torch.log_softmax(torch.stack([(qe * ae).sum(-1), (qe * be).sum(-1)], 1), 1)[:, 0].mean()
    # (the step ran as ONE packed pass of 96 sequences, the check as three passes of 32: same arithmetic per token, but other GEMM
    #  pipelines at other row counts = other fp32 summation orders, which 24 random-init layers amplify)
    assert abs(float(loss) - float(ref)) < 2e-2 * max(1.0, abs(float(ref)))
    model.merge_passes = False
    model.bert.flat_decay.grad = model.bert.flat_nodecay.grad = None
    loss2, _, _ = model(q, qm, a, am, b, am.clone())
    assert abs(float(loss2) - float(ref)) < 1e-3 * max(1.0, abs(float(ref)))


def test_encode_search_ndcg_pipeline_matches_fp32_oracle_pipeline():
    """north_star: nDCG@10 within 1e-3 of the reference path.  Whole eval pipeline (config 5 in miniature): bf16 GPU encoder
    -> resident embeddings -> exact search -> EvalDevQuery, against the fp32 numpy encoder + numpy search.  Queries are
    noisy copies of planted passages, so the metric is non-trivial (some positives are NOT ranked first)."""
    ocfg = O.OracleConfig(vocab_size=800, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                          max_position_embeddings=64)
    P = O.make_params(ocfg, 17, std=0.08)
    cfg = CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=800, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         max_position_embeddings=64)
    model = BertDotNLL(cfg)
    model.bert.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    model.to(DEV).eval()
    rng = np.random.Generator(np.random.PCG64(3))
    npass, nq, Lp, Lq = 400, 60, 64, 32
    p_ids = rng.integers(5, 800, (npass, Lp)); p_ids[:, 0] = 1
    p_mask = np.ones((npass, Lp), np.int64)
    for i in range(npass):
        n = int(rng.integers(20, Lp + 1)); p_mask[i, n:] = 0; p_ids[i, n:] = 0
    pos = rng.permutation(npass)[:nq]
    q_ids = p_ids[pos, :Lq].copy()
    flip = rng.random((nq, Lq)) < 0.3          # heavy token noise: the positive is not always the nearest passage
    q_ids[flip] = rng.integers(5, 800, int(flip.sum())); q_ids[:, 0] = 1
    q_mask = np.ones((nq, Lq), np.int64)
    t = lambda a: torch.from_numpy(a).to(DEV)
    Pe, _ = R.encode_corpus(model, t(p_ids), t(p_mask), batch_size=128)
    Qe, _ = R.encode_corpus(model, t(q_ids), t(q_mask), batch_size=32, is_query=True)
    D, I = R.search(Qe, Pe, 50)
    enc = lambda ids, mask: O.cls_embedding(O.encoder_fwd(P, ocfg, ids, mask)[0][-1]).astype(np.float32)
    Pr, Qr = enc(p_ids, p_mask), enc(q_ids, q_mask)
    assert np.min(np.sum(Pe.cpu().numpy() * Pr, 1) / (np.linalg.norm(Pe.cpu().numpy(), axis=1) * np.linalg.norm(Pr, axis=1))) > 0.999
    Dr, Ir = O.score_topk(Qr, Pr, 50)
    q2id, p2id = np.arange(nq) + 100, np.arange(npass) * 2 + 7
    qrels = {int(q2id[i]): {int(p2id[pos[i]]): 1} for i in range(nq)}
    ndcg, mrr, n, _ = R.eval_dev_query(q2id, p2id, qrels, I, 50)
    ndcg_r, mrr_r, n_r, _ = O.eval_dev_query(q2id, p2id, qrels, Ir, 50)
    assert n == n_r == nq and 0.15 < ndcg_r < 0.999
    # the bf16 encoder against the fp32 oracle: at most two near-tie swaps among the 60 noisy queries (which ones flip moves
    # with the last bits of the LayerNorm statistics, i.e. with the summation order of the wave reductions)
    assert abs(ndcg - ndcg_r) <= 1e-3 + 2.0 / nq * 0.4, (ndcg, ndcg_r)


def test_ance_refresh_cycle_in_miniature(tmp_path):
    """BASELINE configs[3] end to end at toy size: encode passages and training queries with the current model, rebuild the
    hard-negative training file (generate_new_ann's training-set half, ANCE/drivers/run_ann_data_gen.py:332-429), stream it back
    as triplet batches (ANCE/drivers/run_ann.py:247-256) and take a training step on them.  The file is checked line by line
    against the oracle's search + GenerateNegativePassaageID restatement under the same permutations."""
    from cocodr_amd import data as DT
    cfg = CocoBertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, vocab_size=800, hidden_size=128, num_hidden_layers=2,
                         num_attention_heads=2, intermediate_size=256, max_position_embeddings=64)
    torch.manual_seed(5)
    model = BertDotNLL(cfg).to(DEV)
    with torch.no_grad():
        model.bert.flat_decay.mul_(3.0)
    rng = np.random.Generator(np.random.PCG64(8))
    npass, nq, Lp, Lq = 300, 41, 32, 16
    qpath, ppath = str(tmp_path / "train-query"), str(tmp_path / "passages")
    DT.write_token_cache(ppath, [[1] + rng.integers(5, 800, int(rng.integers(6, Lp + 5))).tolist() for _ in range(npass)], Lp)
    DT.write_token_cache(qpath, [[1] + rng.integers(5, 800, int(rng.integers(3, Lq + 3))).tolist() for _ in range(nq)], Lq)
    qc, pc = DT.TokenCache(qpath), DT.TokenCache(ppath)
    model.eval()
    p_ids, p_mask, p2id = pc.batch(np.arange(npass), DEV)
    q_ids, q_mask, q2id = qc.batch(np.arange(nq), DEV)
    Pe, _ = R.encode_corpus(model, p_ids, p_mask, batch_size=128)
    Qe, _ = R.encode_corpus(model, q_ids, q_mask, batch_size=32, is_query=True)
    positives = {int(q): int(rng.integers(0, npass)) for q in range(nq)}
    topk, nneg, chunk_factor, output_num = 40, 10, 2, 3   # round 3 of 2 chunks -> the second half of the queries (+ the remainder)
    perms = []

    def shuffle(lst):  # a recorded stand-in for random.shuffle: the oracle replays the same permutations
        p = rng.permutation(len(lst)).tolist()
        lst[:] = [lst[i] for i in p]
        perms.append(list(lst))

    out = str(tmp_path / "ann_training_data_3")
    n_lines, rr = R.build_ann_training_data(Qe, q2id.cpu().numpy(), Pe, p2id.cpu().numpy(), positives, output_num, out, topk_training=topk,
                                            negative_sample=nneg, ann_chunk_factor=chunk_factor, shuffle=shuffle)
    lo = (nq // 2) * 1
    qsel = np.arange(lo, nq)
    assert len(perms) == len(qsel) + 1 and len(rr) == len(qsel)
    Dr, Ir = O.score_topk(Qe.cpu().numpy()[qsel], Pe.cpu().numpy(), topk)
    neg_ref, rr_ref = O.generate_negatives(qsel, np.arange(npass), positives, Ir, nneg, set(qsel.tolist()), select_topk=False,
                                           permutations=perms[:-1])
    np.testing.assert_allclose(rr, rr_ref)
    want = []
    for split in range(5):
        for qi in perms[-1]:
            qid = int(qsel[qi])
            k = len(neg_ref[qid]) // 5
            want.append("{}\t{}\t{}\n".format(qid, positives[qid], ",".join(str(x) for x in neg_ref[qid][split * k:(split + 1) * k])))
    with open(out) as f:
        got = f.readlines()
    assert n_lines == len(got) == 5 * len(qsel) and got == want
    # ... and train on it: two ranks' streams partition the rows; one step on rank 0's first batch
    s0 = DT.TripletStream(got, qc, pc, 16, rank=0, world_size=2, device=DEV)
    s1 = DT.TripletStream(got, qc, pc, 16, rank=1, world_size=2, device=DEV)
    assert len(s0.rows) + len(s1.rows) == sum(len(neg_ref[int(q)]) // 5 * 5 for q in qsel)
    model.train()
    kw = next(iter(s0))
    assert kw["query_ids"].shape == (16, Lq) and kw["input_ids_b"].shape == (16, Lp)
    loss, acc, _ = model(**kw)
    loss.backward()
    assert torch.isfinite(loss) and float(model.bert.flat_decay.grad.abs().sum()) > 0


def test_config5_end_to_end_one_million_passages_eight_shards_ten_thousand_queries():
    """BASELINE configs[4] at its real size on one GPU (bench.config5_end_to_end): cocodr-large encodes 1 M passages x L128 in 8
    shards (record i -> shard i % 8) and 10 k queries, per-shard search k = 1000, native 8-way merge.  Checked against fp32
    `Q @ P.T` + top-k over the whole merged corpus (what IndexFlatIP computes, evaluate/evaluation/evaluate_beir.py:220-224):
    nDCG@10 of the product's ranking against qrels made of the exact top-10 must be within 1e-3 of the exact ranking's own (1.0,
    north_star tolerance), the returned scores must be the fp32 scores of the returned positions, lists sorted."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    full = os.environ.get("COCODR_CONFIG5_SMALL") is None
    n_pass, nq = (1_000_000, 10_000) if full else (100_000, 2_000)
    out, Q, P, D, I = bench.config5_end_to_end(torch.device(DEV), n_pass=n_pass, nq=nq, keep=True)
    assert P.shape == (n_pass, 1024) and Q.shape == (nq, 1024) and D.shape == I.shape == (nq, 1000)
    assert bool((D[:, :-1] >= D[:, 1:]).all()) and int(I.min()) >= 0 and int(I.max()) < n_pass
    # exact fp32 ranking, in query chunks (the checker: torch's fp32 GEMM + topk, not part of the product)
    ndcg, top1, overlap, nchunk = 0.0, 0, 0.0, 0
    disc = 1.0 / torch.log2(torch.arange(2, 12, device=DEV, dtype=torch.float64))
    for s in range(0, nq, 500):
        S = Q[s:s + 500] @ P.T
        Dx, Ix = torch.topk(S, 10, dim=1)
        got = I[s:s + 500, :10]
        rel = (got[:, :, None] == Ix[:, None, :]).any(-1).to(torch.float64)      # qrels: the exact top-10, gain 1 each
        ndcg += float(((rel * disc).sum(1) / disc.sum()).sum())
        top1 += int((got[:, 0] == Ix[:, 0]).sum())
        # the returned scores are the fp32 scores of the returned positions
        ref = torch.gather(S, 1, I[s:s + 500])
        assert torch.allclose(D[s:s + 500], ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
        k_th = torch.topk(S, 1000, dim=1).values[:, -1:]
        overlap += float((ref >= k_th - 1e-5 * ref.abs().max()).double().mean())
        nchunk += 1
        del S
    ndcg /= nq
    assert ndcg >= 1.0 - 1e-3, ndcg
    assert top1 >= 0.999 * nq and overlap / nchunk >= 0.9999
    assert out["encode_passages_per_sec"] > 5000 and out["search_dot_products_per_sec"] > 2e10
