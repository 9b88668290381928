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
    model = BertDotNLL(CocoBertConfig.large()).to(DEV)
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
