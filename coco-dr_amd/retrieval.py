"""Retrieval half of the hot path: corpus encoding, the reference's shard rule / merge order, sharded
brute-force search, ranking metrics and hard-negative selection.

Reference call sites mirrored here (paths under /root/reference):
  * ``StreamingDataset.__iter__``            ANCE/utils/util.py:384-399      -> shard_indices
  * ``barrier_array_merge``                  ANCE/utils/util.py:87-155       -> merged_order (rank-major)
  * ``InferenceEmbeddingFromStreamDataLoader`` ANCE/drivers/run_ann_data_gen.py:157-212 -> encode_corpus
  * ``faiss.IndexFlatIP(dim).add(P); .search(Q,k)`` evaluate/evaluation/evaluate_beir.py:220-224 -> search /
    sharded_search (embeddings stay resident on the GPUs; only queries and per-shard top-k lists travel)
  * ``EvalDevQuery``                         evaluate/evaluation/evaluate_beir.py:105-194 -> eval_dev_query
  * ``GenerateNegativePassaageID``           ANCE/drivers/run_ann_data_gen.py:497-570 -> generate_negatives
The per-query Python walks stay Python, as in the reference; the FLOPs (encode, score, top-k) are native.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from ._native import lib

__all__ = ["shard_indices", "merged_order", "encode_corpus", "search", "sharded_search", "merge_topk", "merge_shard_lists", "eval_dev_query",
           "EvalDevQuery", "generate_negatives", "ndcg_at_10", "map_at_10", "recall_at", "mrr_at_10", "build_ann_training_data"]


# ----------------------------------------------------------------------------------------------- sharding
def shard_indices(n: int, rank: int, world: int) -> torch.Tensor:
    """Record i belongs to rank i % world (ANCE/utils/util.py:390-392)."""
    return torch.arange(rank, n, world, dtype=torch.int64)


def merged_order(n: int, world: int) -> torch.Tensor:
    """Original record index at every position of the rank-major merged array (ANCE/utils/util.py:117-154)."""
    return torch.cat([shard_indices(n, r, world) for r in range(world)])


# ----------------------------------------------------------------------------------------------- encode
@torch.no_grad()
def encode_corpus(model, input_ids: torch.Tensor, attention_mask: torch.Tensor, batch_size: int = 512,
                  record_ids: Optional[torch.Tensor] = None, is_query: bool = False, pack: bool = True,
                  lengths=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval-mode ``query_emb`` / ``body_emb`` over a token cache (under ``torch.no_grad()``, as run_ann_data_gen.py:183 does), fp32
    [n,H] kept ON DEVICE plus the record ids - the reference copies every batch to the host (``.cpu().numpy()``,
    run_ann_data_gen.py:191-199); here the shard stays in HBM for the search that follows.  ``pack``: store each batch's
    sequences back to back instead of padded (bit-identical embeddings, no work on the padding rows, +18-21 % passages/s;
    include/cocodr.h "Packed batches"; a batch whose masks are not prefix masks runs padded).  ``pack=False``: always padded.
    ``lengths``: the n passage lengths on the host (the token cache stores them in front of every record,
    ANCE/data/msmarco_data.py:279) - the packed layouts are then built without reading anything back, batch after batch;
    ``attention_mask`` may then be None (it is rebuilt from the lengths where a padded run needs it).  Batches with
    ``lengths`` go through ``model.bert.encode_cls`` (the embedding of ``BertDotNLL`` - raw last-layer [CLS] - for both
    queries and passages); a wrapper whose ``query_emb`` / ``body_emb`` do more than that must pass masks instead."""
    n = input_ids.shape[0]
    fn = model.query_emb if is_query else model.body_emb
    outs = []
    bert = getattr(model, "bert", None)
    was = getattr(bert, "pack_sequences", False)
    if lengths is None and attention_mask is None:
        raise ValueError("encode_corpus: give attention_mask or lengths")
    if lengths is not None and bert is None:
        raise ValueError("encode_corpus(lengths=): the model has no .bert encoder to hand the lengths to; pass attention_mask")
    if lengths is not None:
        # the lengths route reads the raw last-layer [CLS] straight from the encoder: only valid for a wrapper whose embedding IS that
        # (BertDotNLL.query_emb / body_emb, ANCE/model/models.py:225-232) - a projection / norm head on top would be skipped silently
        from .modeling import BertDotNLL
        cls_ = type(model)
        plain = (getattr(cls_, "query_emb", None) is getattr(BertDotNLL, "query_emb", None)
                 and getattr(cls_, "body_emb", None) is getattr(BertDotNLL, "body_emb", None))
        if not plain and not getattr(model, "embeddings_are_raw_cls", False):
            raise ValueError("encode_corpus(lengths=) bypasses model.query_emb / body_emb (it reads the raw [CLS] rows): this model "
                             "overrides them - pass attention_mask instead, or set model.embeddings_are_raw_cls = True if they are "
                             "the raw last-layer [CLS]")
    if bert is not None:
        bert.pack_sequences = bool(pack)
    try:
        with torch.no_grad():
            for s in range(0, n, batch_size):
                m = None if attention_mask is None else attention_mask[s:s + batch_size]
                if lengths is not None:  # (packed: laid out from the lengths; padded: the mask - given or rebuilt from them)
                    outs.append(bert.encode_cls(input_ids[s:s + batch_size], m, lengths=lengths[s:s + batch_size]).float())
                else:
                    outs.append(fn(input_ids[s:s + batch_size], m).float())
    finally:
        if bert is not None:
            bert.pack_sequences = was
    emb = torch.cat(outs) if outs else torch.empty((0, model.config.hidden_size), device=input_ids.device)
    ids = torch.arange(n, dtype=torch.int64) if record_ids is None else record_ids
    return emb, ids


# ----------------------------------------------------------------------------------------------- search
def search(Q: torch.Tensor, P: torch.Tensor, k: int, id_offset: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """(D, I) = IndexFlatIP(P).search(Q, k) on one GPU."""
    return ops.score_topk(Q.contiguous(), P.contiguous(), k, id_offset)


class FlatIPIndex:
    """``faiss.IndexFlatIP(dim)`` as the reference uses it - ``add(P)`` once, ``search(Q, k)`` several times on the same index
    (ANCE/drivers/run_ann_data_gen.py:310-317,390: the dev queries, then the training queries; evaluate/evaluation/
    evaluate_beir.py:220-224) - with the passages resident in HBM.  The first search of a given (number of queries, k) builds the
    passages' scale, half-precision split image and filter sample inside the index's workspace; later searches of that shape reuse
    them (``cocodr_score_topk_resident``), which takes the pass over P and three launches out of every search.  Exact inner
    products, (score descending, position ascending) - ``search``'s results, bit for bit."""

    def __init__(self, dim: int):
        self.d = int(dim)
        self._P: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self._key = None

    @property
    def ntotal(self) -> int:
        return 0 if self._P is None else int(self._P.shape[0])

    def add(self, P: torch.Tensor) -> None:
        P = torch.as_tensor(P)
        if P.dim() != 2 or P.shape[1] != self.d:
            raise ValueError(f"FlatIPIndex.add: expected [n, {self.d}], got {tuple(P.shape)}")
        if not P.is_cuda:
            raise RuntimeError("FlatIPIndex keeps its passages on the GPU: add a CUDA tensor (there is no CPU fallback)")
        P = P.to(torch.float32).contiguous()
        self._P = P if self._P is None else torch.cat([self._P, P])
        self._key = None   # (the resident image describes the old passage set)

    def reset(self) -> None:
        self._P = self._ws = self._key = None

    def search(self, Q: torch.Tensor, k: int, id_offset: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        if self._P is None:
            raise RuntimeError("FlatIPIndex.search: the index is empty")
        Q = Q.to(torch.float32).contiguous()
        Nq, Np, k = int(Q.shape[0]), self.ntotal, int(k)
        need = int(lib().cocodr_score_topk_workspace_bytes_dim(Nq, Np, self.d, k))
        if self._ws is None or self._ws.numel() < need:
            self._ws, self._key = torch.empty(need, dtype=torch.uint8, device=self._P.device), None
        key = (Nq, k, need)   # the workspace layout (where the passage image sits) follows from (Nq, Np, H, k)
        out = ops.score_topk(Q, self._P, k, id_offset, workspace=self._ws, p_resident=(key == self._key))
        self._key = key
        return out


def merge_topk(D: torch.Tensor, I: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Host-side utility (tests, small glue): merge UNSORTED candidate lists [Nq, C] into the top k by (score descending,
    position ascending); -1 ids last.  The search path itself merges per-shard lists with the native kernel
    (``merge_shard_lists`` -> cocodr_topk_merge)."""
    big = torch.iinfo(torch.int64).max
    key_i = torch.where(I < 0, torch.full_like(I, big), I)
    o1 = torch.argsort(key_i, dim=1, stable=True)
    D1, I1 = torch.gather(D, 1, o1), torch.gather(I, 1, o1)
    o2 = torch.argsort(D1, dim=1, descending=True, stable=True)[:, :k]
    return torch.gather(D1, 1, o2), torch.gather(I1, 1, o2)


def merge_shard_lists(D: torch.Tensor, I: torch.Tensor, shard_offset: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """W sorted per-shard top-k lists per query -> the top k of the merged corpus (native: cocodr_topk_merge).
    D fp32 / I int32 [W, Nq, k'] with shard-local positions, shard_offset int64 [W]; returns (D [Nq, k], I int64 [Nq, k])."""
    return ops.topk_merge(D.contiguous(), I.contiguous(), shard_offset.contiguous(), k)


def sharded_search(Q_local: torch.Tensor, P_local: torch.Tensor, k: int, local_search: Optional[Callable] = None,
                   local_merge: Optional[Callable] = None, gather: bool = True, force_distributed: bool = False):
    """Search ALL queries against the corpus sharded over the ranks: all-gather the queries (Nq x H fp32), search
    the resident shard, hand every rank the W per-shard lists of ITS block of ceil(Nq / W) queries (one all-to-all: scores
    fp32 + shard-local positions int32, Nq k 8 bytes per rank in total - an all-gather of the lists would move W times
    that and make every rank merge everything), merge that block natively (``merge_shard_lists``).  Positions are indices
    into the rank-major merged corpus (what ``barrier_array_merge`` + ``IndexFlatIP`` would produce); map them through
    ``merged_order`` / ``passage_embedding2id`` for record ids.

    ``gather=True``: every rank returns the full ``(D, I)`` [Nq_total, k] (one all-gather of the merged blocks).
    ``gather=False``: returns ``(D_block, I_block, (q_lo, q_hi))`` - this rank's query block only, for per-query work that
    stays sharded (nDCG, hard negatives).  ``local_search`` / ``local_merge`` replace the native kernels (CPU tests);
    ``force_distributed`` takes the exchange + merge path on a 1-rank process group as well (tests of that path)."""
    import torch.distributed as dist
    fn = local_search or search
    merge = local_merge or merge_shard_lists
    force = bool(force_distributed) and dist.is_available() and dist.is_initialized()
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        D, I = fn(Q_local, P_local, k, 0)
        return (D, I) if gather else (D, I, (0, Q_local.shape[0]))
    W, r = dist.get_world_size(), dist.get_rank()
    dev = Q_local.device
    counts = torch.tensor([Q_local.shape[0], P_local.shape[0]], dtype=torch.int64, device=dev)
    allc = [torch.empty_like(counts) for _ in range(W)]
    dist.all_gather(allc, counts)
    nq = [int(c[0]) for c in allc]
    npass = [int(c[1]) for c in allc]
    H = Q_local.shape[1]
    qmax = max(nq)
    qpad = torch.zeros((qmax, H), dtype=Q_local.dtype, device=dev)
    qpad[:Q_local.shape[0]] = Q_local
    qall = torch.empty((W * qmax, H), dtype=Q_local.dtype, device=dev)
    dist.all_gather_into_tensor(qall, qpad)
    Q = torch.cat([qall[i * qmax:i * qmax + nq[i]] for i in range(W)])
    Nq = Q.shape[0]
    D, I = fn(Q, P_local, k, 0)  # positions local to this shard
    kk = D.shape[1]
    bq = (Nq + W - 1) // W       # queries per block; the last block may be short (padded rows are empty lists)
    Dp = torch.full((W * bq, kk), float("-inf"), dtype=torch.float32, device=dev)
    Ip = torch.full((W * bq, kk), -1, dtype=torch.int32, device=dev)
    Dp[:Nq] = D
    if P_local.shape[0] >= 2 ** 31:
        raise ValueError("sharded_search: a shard holds 2^31 or more passages; shard-local positions travel as int32")
    Ip[:Nq] = I.to(torch.int32)
    Dr = torch.empty((W, bq, kk), dtype=torch.float32, device=dev)
    Ir = torch.empty((W, bq, kk), dtype=torch.int32, device=dev)
    if dist.get_backend() == "nccl":  # RCCL: block j of my lists goes to rank j, I receive everybody's lists of block r
        dist.all_to_all_single(Dr.view(-1), Dp.view(-1))
        dist.all_to_all_single(Ir.view(-1), Ip.view(-1))
    else:  # gloo has no all-to-all: W gathers, rank j collecting everybody's lists of block j (CPU / shared-GPU test path; same result)
        for j in range(W):
            dist.gather(Dp[j * bq:(j + 1) * bq], gather_list=list(Dr.unbind(0)) if r == j else None, dst=j)
            dist.gather(Ip[j * bq:(j + 1) * bq], gather_list=list(Ir.unbind(0)) if r == j else None, dst=j)
    offs = torch.tensor([sum(npass[:w]) for w in range(W)], dtype=torch.int64, device=dev)
    Dm, Im = merge(Dr, Ir, offs, min(k, W * kk))
    q_lo, q_hi = min(r * bq, Nq), min((r + 1) * bq, Nq)
    if not gather:
        return Dm[:q_hi - q_lo], Im[:q_hi - q_lo], (q_lo, q_hi)
    Df = torch.empty((W * bq, Dm.shape[1]), dtype=Dm.dtype, device=dev)
    If = torch.empty((W * bq, Im.shape[1]), dtype=Im.dtype, device=dev)
    dist.all_gather_into_tensor(Df, Dm.contiguous())
    dist.all_gather_into_tensor(If, Im.contiguous())
    return Df[:Nq], If[:Nq]


# ----------------------------------------------------------------------------------------------- metrics
def ndcg_at_10(ranked: Sequence[int], qrel: Dict[int, int]) -> float:
    """trec_eval ``ndcg_cut_10`` (gain = rel, discount 1/log2(rank+1), ideal = judged docs by rel)."""
    gains = np.array([max(qrel.get(int(p), 0), 0) for p in ranked[:10]], dtype=np.float64)
    disc = 1.0 / np.log2(np.arange(2, gains.size + 2))
    ideal = np.sort(np.array([v for v in qrel.values() if v > 0], dtype=np.float64))[::-1][:10]
    idcg = float((ideal / np.log2(np.arange(2, ideal.size + 2))).sum())
    return float((gains * disc).sum() / idcg) if idcg > 0 else 0.0


def mrr_at_10(qids_to_relevant: Dict[int, List[int]], qids_to_ranked: Dict[int, List[int]]) -> float:
    """MS MARCO MRR@10 (evaluate/evaluation/msmarco_eval.py:109-139)."""
    if not qids_to_ranked:
        return 0.0
    acc = 0.0
    for qid, cand in qids_to_ranked.items():
        rel = qids_to_relevant.get(qid)
        if rel is None:
            continue
        hits = [i for i, pid in enumerate(cand[:10]) if pid in rel]
        if hits:
            acc += 1.0 / (hits[0] + 1)
    return acc / len(qids_to_ranked)


def eval_dev_query(query_embedding2id, passage_embedding2id, dev_query_positive_id: Dict[int, Dict[int, int]],
                   I_nearest_neighbor, topN: int, offset2qchar: Optional[dict] = None, offset2pchar: Optional[dict] = None):
    """``EvalDevQuery`` (evaluate/evaluation/evaluate_beir.py:105-194): positions -> pids, de-duplicate, score = -rank,
    skip the ArguAna self match after the rank advanced; nDCG@10 and reciprocal rank averaged over the queries that
    have qrels.  Returns (ndcg@10, mrr, n_queries, prediction)."""
    I = np.asarray(I_nearest_neighbor.cpu() if isinstance(I_nearest_neighbor, torch.Tensor) else I_nearest_neighbor)
    q2id = np.asarray(query_embedding2id)
    p2id = np.asarray(passage_embedding2id)
    prediction: Dict[int, Dict[int, int]] = {}
    nd = rr = 0.0
    n = 0
    for qi in range(I.shape[0]):
        qid = int(q2id[qi])
        pos = I[qi, :topN]
        pids = p2id[pos[pos >= 0]]
        _, first = np.unique(pids, return_index=True)
        uniq = pids[np.sort(first)]  # first occurrence order
        docs: Dict[int, int] = {}
        for rank, pid in enumerate(uniq.tolist(), start=1):
            if offset2qchar and offset2pchar and qid in offset2qchar and pid in offset2pchar and \
                    offset2pchar[pid] == offset2qchar[qid]:
                continue
            docs[pid] = -rank
        prediction[qid] = docs
        qrel = dev_query_positive_id.get(qid)
        if qrel is None:
            continue
        ranked = list(docs.keys())  # insertion order == descending score
        nd += ndcg_at_10(ranked, qrel)
        hit = [i for i, pid in enumerate(ranked) if qrel.get(pid, 0) > 0]
        rr += 1.0 / (hit[0] + 1) if hit else 0.0
        n += 1
    return (nd / n if n else 0.0), (rr / n if n else 0.0), n, prediction


def map_at_10(ranked: Sequence[int], qrel: Dict[int, int]) -> float:
    """trec_eval ``map_cut_10``: precision at every relevant document ranked <= 10, over all relevant documents."""
    n_rel = sum(1 for v in qrel.values() if v > 0)
    hits, acc = 0, 0.0
    for r, p in enumerate(ranked[:10], start=1):
        if qrel.get(int(p), 0) > 0:
            hits += 1
            acc += hits / r
    return acc / n_rel if n_rel else 0.0


def recall_at(ranked: Sequence[int], qrel: Dict[int, int], k: int) -> float:
    """trec_eval ``recall_k``."""
    n_rel = sum(1 for v in qrel.values() if v > 0)
    return sum(1 for p in ranked[:k] if qrel.get(int(p), 0) > 0) / n_rel if n_rel else 0.0


def EvalDevQuery(query_embedding2id, passage_embedding2id, dev_query_positive_id: Dict[int, Dict[int, int]], I_nearest_neighbor,
                 topN: int, offset2qchar: Optional[dict] = None, offset2pchar: Optional[dict] = None):
    """The BEIR script's ``EvalDevQuery`` with its full return tuple (evaluate/evaluation/evaluate_beir.py:105-194):
    ``(ndcg@10, n_queries, map@10, mrr, recall@topN, hole_rate, ms_mrr, Ahole_rate, result, prediction, mrrs, ndcgs)``.
    ``result[qid]`` holds the per-query ``ndcg_cut_10 / map_cut_10 / recip_rank / recall_topN`` the script reads from
    pytrec_eval; the hole rates are the share of unjudged passages among the first ten / all de-duplicated ranks; ``ms_mrr``
    is ``msmarco_eval.compute_metrics`` (``{"MRR @10", "QueriesRanked"}``) over the 1000-slot candidate lists.  The two module
    globals the script reads (``offset2qchar`` / ``offset2pchar``, ArguAna's query == document check) are arguments."""
    I = np.asarray(I_nearest_neighbor.cpu() if isinstance(I_nearest_neighbor, torch.Tensor) else I_nearest_neighbor)
    q2id = np.asarray(query_embedding2id)
    p2id = np.asarray(passage_embedding2id)
    off_q, off_p = offset2qchar or {}, offset2pchar or {}
    prediction: Dict[int, Dict[int, int]] = {}
    ranked_1000: Dict[int, List[int]] = {}
    total = labeled = atotal = alabeled = 0
    for qi in range(I.shape[0]):
        qid = int(q2id[qi])
        qrel = dev_query_positive_id[qid]  # KeyError for an unjudged query, as in the script (:137)
        docs: Dict[int, int] = {}
        slots = ranked_1000.setdefault(qid, [0] * 1000)
        seen = set()
        rank = 0
        pos = I[qi, :topN]
        for pid in p2id[pos[pos >= 0]].tolist():
            if pid in seen:
                continue
            slots[rank] = pid
            hole = pid not in qrel
            atotal += 1
            alabeled += hole
            if rank < 10:
                total += 1
                labeled += hole
            rank += 1
            if qid in off_q and pid in off_p and off_p[pid] == off_q[qid]:
                continue  # counted and listed, but neither scored nor marked seen
            docs[pid] = -rank
            seen.add(pid)
        prediction[qid] = docs
    result: Dict[int, Dict[str, float]] = {}
    for qid, docs in prediction.items():
        qrel = dev_query_positive_id[qid]
        ranked = list(docs.keys())  # insertion order == descending score
        hit = [i for i, pid in enumerate(ranked) if qrel.get(pid, 0) > 0]
        result[qid] = {"ndcg_cut_10": ndcg_at_10(ranked, qrel), "map_cut_10": map_at_10(ranked, qrel),
                       "recip_rank": 1.0 / (hit[0] + 1) if hit else 0.0, f"recall_{topN}": recall_at(ranked, qrel, topN)}
    n = len(result)
    mean = lambda key: sum(r[key] for r in result.values()) / n if n else 0.0
    relevant = {int(q): [pid for pid in rel if pid > 0] for q, rel in dev_query_positive_id.items()}
    ms_mrr = {"MRR @10": mrr_at_10(relevant, ranked_1000), "QueriesRanked": len(ranked_1000)}
    return (mean("ndcg_cut_10"), n, mean("map_cut_10"), mean("recip_rank"), mean(f"recall_{topN}"),
            labeled / total if total else 0.0, ms_mrr, alabeled / atotal if atotal else 0.0, result, prediction,
            [r["recip_rank"] for r in result.values()], [r["ndcg_cut_10"] for r in result.values()])


def generate_negatives(query_embedding2id, passage_embedding2id, training_query_positive_id: Dict[int, int],
                       I_nearest_neighbor, negative_sample: int, effective_q_id: Optional[Iterable[int]] = None,
                       ann_measure_topk_mrr: bool = False, shuffle: Optional[Callable[[list], None]] = None):
    """``GenerateNegativePassaageID`` (ANCE/drivers/run_ann_data_gen.py:497-570).  Default (as in the driver): the whole
    retrieved list is walked in a shuffled order - ``shuffle`` permutes ``list(range(k))`` in place, default Python's
    ``random.shuffle`` (the driver seeds ``random`` in ``set_seed``, so the same seed gives the same negatives);
    ``ann_measure_topk_mrr=True`` walks the first ``negative_sample + 1`` instead.  Skip the positive, skip duplicates,
    keep ``negative_sample``.  Also returns the reciprocal rank of the positive over the whole retrieved list."""
    import random
    I = np.asarray(I_nearest_neighbor.cpu() if isinstance(I_nearest_neighbor, torch.Tensor) else I_nearest_neighbor)
    q2id = np.asarray(query_embedding2id)
    p2id = np.asarray(passage_embedding2id)
    eff = None if effective_q_id is None else {int(x) for x in effective_q_id}
    shuffle = shuffle or random.shuffle
    negatives: Dict[int, List[int]] = {}
    rr: List[float] = []
    for qi in range(I.shape[0]):
        qid = int(q2id[qi])
        if eff is not None and qid not in eff:
            continue
        pos = training_query_positive_id[qid]
        pids = p2id[I[qi]]
        where = np.nonzero(pids == pos)[0]
        rr.append(1.0 / (int(where[0]) + 1) if where.size else 0.0)
        if ann_measure_topk_mrr:
            window = pids[:negative_sample + 1]
        else:
            order = list(range(I.shape[1]))
            shuffle(order)
            window = pids[np.asarray(order, dtype=np.int64)]
        window = window[window != pos]
        _, first = np.unique(window, return_index=True)
        negatives[qid] = window[np.sort(first)][:negative_sample].tolist()
    return negatives, np.array(rr)


def build_ann_training_data(query_embedding: torch.Tensor, query_embedding2id, passage_embedding: torch.Tensor, passage_embedding2id,
                            training_query_positive_id: Dict[int, int], output_num: int, out_path: str, topk_training: int = 200,
                            negative_sample: int = 20, ann_chunk_factor: int = 1, ann_measure_topk_mrr: bool = False,
                            shuffle: Optional[Callable[[list], None]] = None, n_splits: int = 5):
    """The training-set half of ``generate_new_ann`` (ANCE/drivers/run_ann_data_gen.py:332-429, the non-clustered branch) on
    resident embeddings: pick this round's chunk of the training queries (``output_num % ann_chunk_factor``, the last chunk takes
    the remainder), search the top ``topk_training`` passages (exact inner product, ``search``), draw the hard negatives
    (``generate_negatives``), shuffle the query order and write ``qid\\tpos\\tneg,neg,...`` in ``n_splits`` passes, each with its
    slice of every query's negatives.  ``shuffle`` permutes a list in place (default ``random.shuffle`` as in the driver) and is
    used for the negatives' walk order and for the query order.  Returns (lines written, reciprocal ranks of the positives)."""
    import random
    from .data import write_triplets
    shuffle = shuffle or random.shuffle
    q2id = np.asarray(query_embedding2id).reshape(-1)
    chunk_factor = int(ann_chunk_factor)
    effective_idx = int(output_num) % chunk_factor if chunk_factor > 0 else 0  # (the driver takes the modulus before its <= 0 guard)
    if chunk_factor <= 0:
        chunk_factor = 1
    n = len(q2id)
    per = n // chunk_factor
    lo = per * effective_idx
    hi = n if effective_idx == chunk_factor - 1 else lo + per
    Q, q2id = query_embedding[lo:hi], q2id[lo:hi]
    _, I = search(Q, passage_embedding, int(topk_training))
    effective = {int(x) for x in q2id}
    negatives, rr = generate_negatives(q2id, passage_embedding2id, training_query_positive_id, I, negative_sample, effective,
                                       ann_measure_topk_mrr, shuffle)
    order = list(range(len(q2id)))
    shuffle(order)
    lines = write_triplets(out_path, [int(q2id[i]) for i in order], training_query_positive_id, negatives, n_splits)
    return lines, rr
