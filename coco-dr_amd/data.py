"""On-disk formats at the edges of the hot path, so the reference's preprocessed data and shard files
interchange with this implementation.

  * token cache  - ANCE/data/msmarco_data.py:279,295 (writers), ANCE/utils/util.py:316-370 ``EmbeddingCache`` (reader):
        record = [4-byte big-endian token count][L x int32, native byte order]   (pre-merge split files carry an extra
        8-byte big-endian id prefix), plus ``{path}_meta`` JSON {"type": "int32", "total_number": n, "embedding_size": L}
  * model inputs - ``GetProcessingFn`` ANCE/data/msmarco_data.py:297-325: ids int32 [L], attention mask = first
        ``len`` positions (token_type is never passed to the model)
  * embedding shards - ``barrier_array_merge`` ANCE/utils/util.py:117-123 / evaluate_beir.py:200-209: pickle protocol 4
        ndarrays ``{prefix}__emb_p__data_obj_{rank}.pb`` and ``{prefix}__embid_p__data_obj_{rank}.pb``
  * hard-negative file - ANCE/drivers/run_ann_data_gen.py:403-429: ``qid\\tpos\\tneg,neg,...`` written in 5 splits
  * training batches - ``GetTripletTrainingDataProcessingFn`` / ``GetTrainingDataProcessingFn`` (ANCE/data/msmarco_data.py:328-384)
        over ``StreamingDataset`` (ANCE/utils/util.py:372-399, line i -> rank i % W) and a DataLoader of ``train_batch_size``:
        ``triplet_records`` / ``pair_records`` / ``TripletStream`` build the same rows in the same order, a batch per gather
The whole cache is memory-mapped and batches are gathered with one fancy-index (no per-record seeks).
"""
from __future__ import annotations

import json
import os
import pickle
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = ["TokenCache", "write_token_cache", "save_embedding_shard", "load_embedding_shards", "write_triplets",
           "read_triplets", "triplet_records", "pair_records", "TripletStream"]


class TokenCache:
    """Memory-mapped view of an ``EmbeddingCache`` file: ``lengths`` [n] int64 and ``tokens`` [n, L] int32."""

    def __init__(self, base_path: str):
        with open(base_path + "_meta") as f:
            meta = json.load(f)
        self.dtype = np.dtype(meta["type"])
        self.total_number = int(meta["total_number"])
        self.max_len = int(meta["embedding_size"])
        self.record_size = self.max_len * self.dtype.itemsize + 4
        rec = np.dtype([("len", ">u4"), ("tok", self.dtype, (self.max_len,))])
        assert rec.itemsize == self.record_size
        self._mm = np.memmap(base_path, dtype=rec, mode="r", shape=(self.total_number,))

    def __len__(self) -> int:
        return self.total_number

    def __getitem__(self, key: int) -> Tuple[int, np.ndarray]:
        """Same return value as EmbeddingCache.__getitem__: (passage_len, tokens)."""
        if key < 0 or key >= self.total_number:
            raise IndexError(f"index {key} out of range for a cache of {self.total_number} records")
        r = self._mm[key]
        return int(r["len"]), np.asarray(r["tok"])

    def batch(self, indices: Sequence[int], device=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(input_ids int64 [b,L], attention_mask int64 [b,L], record index int64 [b]) for the model API -
        GetProcessingFn's ids / mask (ANCE/data/msmarco_data.py:302-305) for a whole batch at once."""
        idx = np.asarray(indices, dtype=np.int64)
        recs = self._mm[idx]
        ids = torch.from_numpy(np.ascontiguousarray(recs["tok"]).astype(np.int64))
        lens = torch.from_numpy(recs["len"].astype(np.int64))
        mask = (torch.arange(self.max_len)[None, :] < lens[:, None]).to(torch.int64)
        out = (ids, mask, torch.from_numpy(idx))
        return tuple(t.to(device) for t in out) if device is not None else out


def write_token_cache(base_path: str, token_lists: Iterable[Sequence[int]], max_len: int, pad_token: int = 0) -> int:
    """Write records the way PassagePreprocessingFn / QueryPreprocessingFn + the merge step do (truncate to max_len,
    right-pad with ``pad_token``), and the ``_meta`` file.  Returns the number of records."""
    n = 0
    with open(base_path, "wb") as f:
        for toks in token_lists:
            toks = list(toks)[:max_len]
            ln = len(toks)
            arr = np.full(max_len, pad_token, np.int32)
            arr[:ln] = toks
            f.write(ln.to_bytes(4, "big") + arr.tobytes())
            n += 1
    with open(base_path + "_meta", "w") as f:
        json.dump({"type": "int32", "total_number": n, "embedding_size": max_len}, f)
    return n


def save_embedding_shard(out_dir: str, prefix: str, rank: int, emb, ids) -> None:
    """``{prefix}__emb_p__data_obj_{rank}.pb`` / ``{prefix}__embid_p__data_obj_{rank}.pb`` (pickle protocol 4)."""
    os.makedirs(out_dir, exist_ok=True)
    e = emb.detach().float().cpu().numpy() if isinstance(emb, torch.Tensor) else np.asarray(emb, np.float32)
    i = ids.detach().cpu().numpy() if isinstance(ids, torch.Tensor) else np.asarray(ids)
    for tag, arr in (("emb_p", e), ("embid_p", i)):
        with open(os.path.join(out_dir, f"{prefix}__{tag}__data_obj_{rank}.pb"), "wb") as h:
            pickle.dump(arr, h, protocol=4)


def load_embedding_shards(out_dir: str, prefix: str, max_ranks: int = 8) -> Tuple[np.ndarray, np.ndarray]:
    """Rank-major concatenation of the shard files that exist, as evaluate_beir.py:200-218 does (stop at the first gap)."""
    embs, ids = [], []
    for r in range(max_ranks):
        pe = os.path.join(out_dir, f"{prefix}__emb_p__data_obj_{r}.pb")
        pi = os.path.join(out_dir, f"{prefix}__embid_p__data_obj_{r}.pb")
        if not (os.path.exists(pe) and os.path.exists(pi)):
            break
        with open(pe, "rb") as h:
            embs.append(pickle.load(h))
        with open(pi, "rb") as h:
            ids.append(pickle.load(h))
    if not embs:
        raise FileNotFoundError(f"no shards for prefix {prefix!r} under {out_dir}")
    return np.concatenate(embs, 0), np.concatenate(ids, 0)


def write_triplets(path: str, query_order: Sequence[int], positives: Dict[int, int], negatives: Dict[int, List[int]],
                   n_splits: int = 5) -> int:
    """The hard-negative training file: for each of ``n_splits`` passes over the queries, one line
    ``qid\\tpos\\tneg,neg,...`` holding that split's slice of the query's negatives."""
    lines = 0
    with open(path, "w") as f:
        for split in range(n_splits):
            for qid in query_order:
                if qid not in positives or qid not in negatives:
                    continue
                negs = negatives[qid]
                k = len(negs) // n_splits
                f.write("{}\t{}\t{}\n".format(qid, positives[qid], ",".join(str(p) for p in negs[split * k:(split + 1) * k])))
                lines += 1
    return lines


def read_triplets(path: str) -> List[Tuple[int, int, List[int]]]:
    out = []
    with open(path) as f:
        for line in f:
            a = line.rstrip("\n").split("\t")
            out.append((int(a[0]), int(a[1]), [int(x) for x in a[2].split(",")] if len(a) > 2 and a[2] else []))
    return out


# ----------------------------------------------------------------------------------------------- training batches
def _parse_line(line: str) -> Tuple[int, int, List[int]]:
    a = line.rstrip("\n").split("\t")  # ANCE/data/msmarco_data.py:330-334 (a line without negatives yields nothing)
    return int(a[0]), int(a[1]), [int(x) for x in a[2].split(",")] if len(a) > 2 and a[2].strip() else []


def triplet_records(lines: Sequence[str], rank: int = 0, world_size: int = 1) -> np.ndarray:
    """[n, 3] int64 rows (qid, positive pid, negative pid) in the order ``StreamingDataset`` over
    ``GetTripletTrainingDataProcessingFn`` yields them on ``rank``: line i belongs to rank i % world_size
    (ANCE/utils/util.py:390-392), one row per negative of the line (ANCE/data/msmarco_data.py:377-382)."""
    rows = []
    for i, line in enumerate(lines):
        if world_size > 1 and i % world_size != rank:
            continue
        qid, pos, negs = _parse_line(line)
        rows.extend((qid, pos, n) for n in negs)
    return np.asarray(rows, dtype=np.int64).reshape(-1, 3)


def pair_records(lines: Sequence[str], rank: int = 0, world_size: int = 1) -> np.ndarray:
    """[n, 3] int64 rows (qid, pid, label) of ``GetTrainingDataProcessingFn`` (ANCE/data/msmarco_data.py:328-356): for every
    negative of a line the positive pair (label 1) then the negative pair (label 0)."""
    t = triplet_records(lines, rank, world_size)
    out = np.empty((2 * len(t), 3), dtype=np.int64)
    out[0::2, 0], out[0::2, 1], out[0::2, 2] = t[:, 0], t[:, 1], 1
    out[1::2, 0], out[1::2, 1], out[1::2, 2] = t[:, 0], t[:, 2], 0
    return out


class TripletStream:
    """The ANCE training stream of one rank (ANCE/drivers/run_ann.py:247-256, 297-308): batches of ``batch_size`` consecutive
    triplet rows as the keyword arguments of ``BertDot_NLL_LN.forward`` - ``query_ids / attention_mask_q`` [b, Lq],
    ``input_ids_a / attention_mask_a`` (positive) and ``input_ids_b / attention_mask_b`` (negative) [b, Lp] - gathered from the two
    token caches with one fancy-index each.  Like the reference's DataLoader the last batch may be short."""

    def __init__(self, lines: Sequence[str], query_cache: "TokenCache", passage_cache: "TokenCache", batch_size: int,
                 rank: int = 0, world_size: int = 1, device=None):
        if batch_size <= 0:
            raise ValueError("batch_size must be positive")
        self.rows = triplet_records(lines, rank, world_size)
        self.q, self.p, self.bs, self.device = query_cache, passage_cache, int(batch_size), device

    def __len__(self) -> int:
        return (len(self.rows) + self.bs - 1) // self.bs

    def __iter__(self):
        for o in range(0, len(self.rows), self.bs):
            r = self.rows[o:o + self.bs]
            q_ids, q_mask, _ = self.q.batch(r[:, 0], self.device)
            a_ids, a_mask, _ = self.p.batch(r[:, 1], self.device)
            b_ids, b_mask, _ = self.p.batch(r[:, 2], self.device)
            yield {"query_ids": q_ids, "attention_mask_q": q_mask, "input_ids_a": a_ids, "attention_mask_a": a_mask,
                   "input_ids_b": b_ids, "attention_mask_b": b_mask}
