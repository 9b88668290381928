"""numpy / pure-Python restatement of the retrieval half of the COCO-DR hot path:
brute-force inner-product top-k, the shard rule and merge order, ranking metrics and
hard-negative selection.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Third-party pieces restated from their published definitions (absent from the image):
 * faiss-cpu==1.6.4 ``IndexFlatIP.search`` (warmup/commands/install.sh:4) - exact search,
   mathematically ``argsort(-Q.P^T)[:, :k]``; tie order is implementation defined, here
   "lower corpus position first" (parity unpinned vs faiss, irrelevant without exact ties).
 * pytrec_eval (unpinned, warmup/commands/install.sh:3) ``ndcg_cut_10`` / ``recip_rank`` / ``map_cut_10`` /
   ``recall_N`` - trec_eval definitions (parity unpinned; checked on hand-computed cases in tests).
Everything AROUND those two libraries is pinned: ``eval_dev_query_beir`` and ``generate_negatives`` are checked against
outputs of the reference's own functions (tests/golden/make_golden.py ``evaldev`` / ``negatives``), which are run with
``TrecEvaluatorStandIn`` in place of pytrec_eval - so the prediction dictionaries, hole rates, MS MARCO MRR and negative
lists are reference outputs, the four trec_eval numbers are not.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

__all__ = [
    "score_topk", "merge_topk", "shard_indices", "merged_order", "eval_dev_query",
    "ndcg_cut", "recip_rank", "mrr_at_10", "generate_negatives", "map_cut", "recall_at", "trec_metrics",
    "eval_dev_query_beir", "TrecEvaluatorStandIn",
]


def score_topk(Q: np.ndarray, P: np.ndarray, k: int, chunk: int = 1024) -> Tuple[np.ndarray, np.ndarray]:
    """``faiss.IndexFlatIP(dim).add(P); .search(Q, k)`` - evaluate/evaluation/evaluate_beir.py:220-224,
    ANCE/drivers/run_ann_data_gen.py:310-317,390.  Returns (D [Nq,k] float32 descending, I [Nq,k] int64
    positions into P).  Ties broken towards the lower position.  Rows are padded with (-inf, -1) when k > Np
    (faiss pads with -1 ids)."""
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    P = np.ascontiguousarray(P, dtype=np.float32)
    nq, npass = Q.shape[0], P.shape[0]
    kk = min(k, npass)
    D = np.full((nq, k), -np.inf, np.float32)
    I = np.full((nq, k), -1, np.int64)
    for s in range(0, nq, chunk):
        S = Q[s:s + chunk] @ P.T
        # stable argsort on -S gives descending score, ascending index within ties
        idx = np.argsort(-S, axis=1, kind="stable")[:, :kk]
        D[s:s + chunk, :kk] = np.take_along_axis(S, idx, 1)
        I[s:s + chunk, :kk] = idx
    return D, I


def merge_topk(Ds: Sequence[np.ndarray], Is: Sequence[np.ndarray], k: int) -> Tuple[np.ndarray, np.ndarray]:
    """k-way merge of per-shard (score, global-position) lists (SURVEY 8e scoring row): the result of
    searching the concatenated corpus.  Ties -> lower global position first."""
    D = np.concatenate(Ds, 1)
    I = np.concatenate(Is, 1)
    # sort by (-score, index); invalid entries (-1) go last
    key_i = np.where(I < 0, np.iinfo(np.int64).max, I)
    order = np.lexsort((key_i, -D.astype(np.float64)), axis=1)[:, :k]
    return np.take_along_axis(D, order, 1), np.take_along_axis(I, order, 1)


def shard_indices(n: int, rank: int, world: int) -> np.ndarray:
    """``StreamingDataset.__iter__`` - ANCE/utils/util.py:390-392: record i goes to rank i % world."""
    return np.arange(rank, n, world, dtype=np.int64)


def merged_order(n: int, world: int) -> np.ndarray:
    """``barrier_array_merge`` - ANCE/utils/util.py:117-154: rank-major concatenation of the shards,
    i.e. the original record index held at each merged position."""
    return np.concatenate([shard_indices(n, r, world) for r in range(world)])


# --------------------------------------------------------------------------- metrics
def ndcg_cut(ranked_pids: Sequence[int], qrel: Dict[int, int], k: int = 10) -> float:
    """trec_eval ndcg_cut_k: gain = rel, discount 1/log2(rank+1), ideal = qrels sorted descending."""
    dcg = 0.0
    for r, pid in enumerate(ranked_pids[:k], start=1):
        rel = qrel.get(pid, 0)
        if rel > 0:
            dcg += rel / math.log2(r + 1)
    ideal = sorted((v for v in qrel.values() if v > 0), reverse=True)[:k]
    idcg = sum(rel / math.log2(r + 1) for r, rel in enumerate(ideal, start=1))
    return dcg / idcg if idcg > 0 else 0.0


def recip_rank(ranked_pids: Sequence[int], qrel: Dict[int, int]) -> float:
    for r, pid in enumerate(ranked_pids, start=1):
        if qrel.get(pid, 0) > 0:
            return 1.0 / r
    return 0.0


def mrr_at_10(qids_to_relevant: Dict[int, List[int]], qids_to_ranked: Dict[int, List[int]]) -> float:
    """evaluate/evaluation/msmarco_eval.py:109-139 ``compute_metrics`` - MRR@10 over the ranked queries."""
    total = 0.0
    for qid, cand in qids_to_ranked.items():
        if qid not in qids_to_relevant:
            continue
        target = qids_to_relevant[qid]
        for i in range(min(10, len(cand))):
            if cand[i] in target:
                total += 1.0 / (i + 1)
                break
    return total / len(qids_to_ranked) if qids_to_ranked else 0.0


def eval_dev_query(query_embedding2id: Sequence[int], passage_embedding2id: Sequence[int],
                   dev_query_positive_id: Dict[int, Dict[int, int]], I_nearest_neighbor: np.ndarray,
                   topN: int, self_match: Optional[Tuple[Dict[int, str], Dict[int, str]]] = None):
    """``EvalDevQuery`` - evaluate/evaluation/evaluate_beir.py:105-194 (ANCE copy :573-621).

    Per query: walk the top-N positions -> pid, de-duplicate, score = -rank (:146), skip the
    self-match (ArguAna, :143-145) *after* the rank was advanced; nDCG@10 / MRR are means over
    evaluated queries (:178-190).  Returns (ndcg@10, mrr, n_queries, prediction)."""
    prediction: Dict[int, Dict[int, int]] = {}
    off_q, off_p = self_match if self_match is not None else ({}, {})
    for qi in range(len(I_nearest_neighbor)):
        qid = int(query_embedding2id[qi])
        prediction[qid] = {}
        seen = set()
        rank = 0
        for idx in I_nearest_neighbor[qi][:topN]:
            if idx < 0:
                continue
            pid = int(passage_embedding2id[idx])
            if pid in seen:
                continue
            rank += 1
            if qid in off_q and pid in off_p and off_p[pid] == off_q[qid]:
                continue
            prediction[qid][pid] = -rank
            seen.add(pid)
    nd, rr, n = 0.0, 0.0, 0
    for qid, docs in prediction.items():
        if qid not in dev_query_positive_id:
            continue  # pytrec_eval only evaluates queries present in the qrels
        ranked = [pid for pid, _ in sorted(docs.items(), key=lambda kv: -kv[1])]
        nd += ndcg_cut(ranked, dev_query_positive_id[qid], 10)
        rr += recip_rank(ranked, dev_query_positive_id[qid])
        n += 1
    return (nd / n if n else 0.0), (rr / n if n else 0.0), n, prediction


def generate_negatives(query_embedding2id: Sequence[int], passage_embedding2id: Sequence[int],
                       training_query_positive_id: Dict[int, int], I_nearest_neighbor: np.ndarray,
                       negative_sample: int, effective_q_id: Optional[Iterable[int]] = None,
                       select_topk: bool = True, permutations: Optional[Sequence[Sequence[int]]] = None):
    """``GenerateNegativePassaageID`` - ANCE/drivers/run_ann_data_gen.py:497-570.  ``select_topk`` is
    ``args.ann_measure_topk_mrr``: True walks the first ``negative_sample + 1`` retrieved passages (:534-535); False (the
    driver's default) walks the WHOLE retrieved list in a shuffled order (:536-540) - ``permutations[j]`` is what
    ``random.shuffle`` leaves in ``list(range(k))`` for the j-th effective query.  Either way: skip the positive, skip
    duplicates, stop once ``negative_sample`` are kept (:548-566).  Also the per-query reciprocal rank of the positive
    over the whole list (:521-532)."""
    eff = None if effective_q_id is None else set(int(x) for x in effective_q_id)
    out: Dict[int, List[int]] = {}
    rr: List[float] = []
    j = 0
    for qi in range(I_nearest_neighbor.shape[0]):
        qid = int(query_embedding2id[qi])
        if eff is not None and qid not in eff:
            continue
        pos = training_query_positive_id[qid]
        top = I_nearest_neighbor[qi]
        r = 0.0
        for rank, idx in enumerate(top, start=1):
            if int(passage_embedding2id[idx]) == pos:
                r = 1.0 / rank
                break
        rr.append(r)
        if select_topk:
            walk = top[:negative_sample + 1]
        else:
            walk = top[np.asarray(permutations[j], dtype=np.int64)]
        j += 1
        negs: List[int] = []
        for idx in walk:
            pid = int(passage_embedding2id[idx])
            if pid == pos or pid in negs:
                continue
            if len(negs) >= negative_sample:
                break
            negs.append(pid)
        out[qid] = negs
    return out, np.array(rr)

# --------------------------------------------------------------------------- full BEIR evaluation (a18)
def map_cut(ranked_pids: Sequence[int], qrel: Dict[int, int], k: int = 10) -> float:
    """trec_eval map_cut_k: sum of precision@rank over the relevant (rel >= 1) documents ranked <= k, divided by the
    number of relevant documents in the qrels."""
    n_rel = sum(1 for v in qrel.values() if v > 0)
    hits, acc = 0, 0.0
    for r, pid in enumerate(ranked_pids[:k], start=1):
        if qrel.get(pid, 0) > 0:
            hits += 1
            acc += hits / r
    return acc / n_rel if n_rel else 0.0


def recall_at(ranked_pids: Sequence[int], qrel: Dict[int, int], k: int) -> float:
    """trec_eval recall_k: relevant documents ranked <= k over all relevant documents."""
    n_rel = sum(1 for v in qrel.values() if v > 0)
    return sum(1 for pid in ranked_pids[:k] if qrel.get(pid, 0) > 0) / n_rel if n_rel else 0.0


TREC_RECALL_CUTS = (5, 10, 15, 20, 30, 100, 200, 500, 1000)  # trec_eval's default cut-offs for the ``recall`` measure


def trec_metrics(docs: Dict[int, float], qrel: Dict[int, int]) -> Dict[str, float]:
    """The per-query record the reference reads from pytrec_eval (evaluate_beir.py:150-185): documents ordered by score
    descending (trec_eval breaks score ties by document id descending; the scores here are distinct ranks)."""
    ranked = [pid for pid, _ in sorted(docs.items(), key=lambda kv: (-kv[1], -int(kv[0])))]
    out = {"ndcg_cut_10": ndcg_cut(ranked, qrel, 10), "map_cut_10": map_cut(ranked, qrel, 10), "recip_rank": recip_rank(ranked, qrel)}
    for c in TREC_RECALL_CUTS:
        out[f"recall_{c}"] = recall_at(ranked, qrel, c)
    return out


class TrecEvaluatorStandIn:
    """Duck type of ``pytrec_eval.RelevanceEvaluator(qrels, measures).evaluate(run)`` with string ids, built on the
    restatements above - used ONLY by tests/golden/make_golden.py to let the reference's own EvalDevQuery run in a
    container without pytrec_eval."""

    def __init__(self, qrels, measures=None):
        self.qrels = qrels

    def evaluate(self, run):
        res = {}
        for q, docs in run.items():
            if q not in self.qrels:
                continue
            qrel = {int(k): int(v) for k, v in self.qrels[q].items()}
            res[q] = trec_metrics({int(k): v for k, v in docs.items()}, qrel)
        return res


def eval_dev_query_beir(query_embedding2id: Sequence[int], passage_embedding2id: Sequence[int],
                        dev_query_positive_id: Dict[int, Dict[int, int]], I_nearest_neighbor: np.ndarray, topN: int,
                        self_match: Optional[Tuple[Dict[int, str], Dict[int, str]]] = None) -> dict:
    """``EvalDevQuery`` of the BEIR evaluation script, every returned quantity - evaluate/evaluation/evaluate_beir.py:105-194,
    statement by statement: the de-duplicated walk (:131-147; the ArguAna self match is written into the MS MARCO candidate
    list and counted, but neither scored nor marked seen), hole rates = share of UNJUDGED passages among the first ten /
    all walked ranks (:137-142,188-189), MS MARCO MRR@10 over ``qids_to_ranked_candidate_passages`` with relevant =
    judged pids > 0 (:158-170), and the means of ndcg_cut_10 / map_cut_10 / recip_rank / recall_topN over the evaluated
    queries (:178-190)."""
    off_q, off_p = self_match if self_match is not None else ({}, {})
    prediction: Dict[int, Dict[int, int]] = {}
    ranked_1000: Dict[int, List[int]] = {}
    total = labeled = atotal = alabeled = 0
    for qi in range(len(I_nearest_neighbor)):
        qid = int(query_embedding2id[qi])
        prediction[qid] = {}
        seen = set()
        rank = 0
        if qid not in ranked_1000:
            ranked_1000[qid] = [0] * 1000
        for idx in I_nearest_neighbor[qi][:topN]:
            pid = int(passage_embedding2id[idx])
            if pid in seen:
                continue
            ranked_1000[qid][rank] = pid
            atotal += 1
            unjudged = pid not in dev_query_positive_id[qid]
            alabeled += int(unjudged)
            if rank < 10:
                total += 1
                labeled += int(unjudged)
            rank += 1
            if qid in off_q and pid in off_p and off_p[pid] == off_q[qid]:
                continue
            prediction[qid][pid] = -rank
            seen.add(pid)
    result = {qid: trec_metrics(docs, dev_query_positive_id[qid]) for qid, docs in prediction.items() if qid in dev_query_positive_id}
    relevant = {int(q): [pid for pid in rel if pid > 0] for q, rel in dev_query_positive_id.items()}
    n = len(result)
    mean = lambda key: sum(r[key] for r in result.values()) / n if n else 0.0
    return {"ndcg": mean("ndcg_cut_10"), "n_queries": n, "map": mean("map_cut_10"), "mrr": mean("recip_rank"),
            "recall": mean(f"recall_{topN}"), "hole_rate": labeled / total if total else 0.0,
            "ms_mrr": mrr_at_10(relevant, ranked_1000), "ahole_rate": alabeled / atotal if atotal else 0.0,
            "result": result, "prediction": prediction, "ranked_1000": ranked_1000,
            "mrrs": [r["recip_rank"] for r in result.values()], "ndcgs": [r["ndcg_cut_10"] for r in result.values()]}
