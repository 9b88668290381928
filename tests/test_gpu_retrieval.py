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
    assert abs(float(loss) - float(ref)) < 1e-3 * max(1.0, abs(float(ref)))


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
