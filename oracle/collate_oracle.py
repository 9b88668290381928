"""CPU restatement of the Condenser collator (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py): COCO/data.py:44-55
(word grouping), :68-99 (whole-word mask), :101-117 (truncation window), :119-156 (assembly), plus the 80/10/10 rule
of transformers' ``DataCollatorForWholeWordMask.torch_mask_tokens`` (third-party; published rule: a masked position
becomes [MASK] with p 0.8, a random vocabulary id with p 0.1, stays with p 0.1; labels -100 elsewhere).

The word grouping, the greedy selection and the truncation are pinned against the reference's own methods driven with
the same permutation / offset (tests/test_oracle_golden.py).  Random draws come from ``collate_rand`` - the
counter-based hash the HIP kernel uses - because Python's ``random`` stream cannot exist on a device.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

__all__ = ["collate_rand", "whole_word_groups", "whole_word_mask", "collate_span", "RS_TRUNC", "RS_SHUFFLE", "RS_REPLACE",
           "RS_RANDOM", "RS_WORD"]

RS_TRUNC, RS_SHUFFLE, RS_REPLACE, RS_RANDOM, RS_WORD = 1, 2, 3, 4, 5
_M = (1 << 64) - 1


def collate_rand(seed: int, ex: int, stream: int, ctr: int) -> int:
    x = (seed * 0x9E3779B97F4A7C15 + ex * 0xBF58476D1CE4E5B9 + stream * 0x94D049BB133111EB + ctr) & _M
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & _M
    x ^= x >> 27; x = (x * 0x94D049BB133111EB) & _M
    x ^= x >> 31
    return x >> 32


def _uniform(r: int) -> np.float32:
    return np.float32(r >> 8) * np.float32(1.0 / 16777216.0)


def whole_word_groups(is_subword_of_token: Sequence[bool], is_special_of_token: Sequence[bool] = None) -> List[List[int]]:
    """data.py:44-55: a "##" piece joins the previous word; a special token ([UNK], a stray [SEP] ... inside the span) is
    skipped (:47-48) - it joins no word and a "##" piece behind it still attaches to the last word opened."""
    cand: List[List[int]] = []
    for i, sub in enumerate(is_subword_of_token):
        if is_special_of_token is not None and is_special_of_token[i]:
            continue
        if len(cand) >= 1 and sub:
            cand[-1].append(i)
        else:
            cand.append([i])
    return cand


def whole_word_mask(groups: List[List[int]], order: Sequence[int], n_tokens: int, mlm_probability: float,
                    max_predictions: int = 512) -> List[int]:
    """data.py:76-99 with ``random.shuffle`` replaced by an explicit ``order`` of the word groups."""
    num_to_predict = min(max_predictions, max(1, int(round(n_tokens * mlm_probability))))
    masked: List[int] = []
    covered = set()
    for w in order:
        index_set = groups[w]
        if len(masked) >= num_to_predict:
            break
        if len(masked) + len(index_set) > num_to_predict:
            continue
        if any(i in covered for i in index_set):
            continue
        for i in index_set:
            covered.add(i)
            masked.append(i)
    return [1 if i in covered else 0 for i in range(n_tokens)]


def collate_span(tokens: Sequence[int], is_subword: np.ndarray, seed: int, ex: int, L: int, cls_id: int, sep_id: int, pad_id: int,
                 mask_id: int, mlm_probability: float) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """One row of the batch: (input_ids, labels, attention_mask), int64 [L].  ``is_subword`` is the per-vocabulary class the
    kernel takes: 0 starts a word, 1 continues one ("##"), 2 is a special token."""
    V = len(is_subword)
    toks = list(tokens)
    tgt = L - 2
    if len(toks) > tgt:                                            # data.py:101-117
        trunc = len(toks) - tgt
        left = collate_rand(seed, ex, RS_TRUNC, 0) % (trunc + 1)
        toks = toks[left:left + tgt]
    toks = [min(max(int(t), 0), V - 1) for t in toks]
    n = len(toks)
    groups = whole_word_groups([is_subword[t] == 1 for t in toks], [is_subword[t] == 2 for t in toks])  # vocabulary classes
    order = sorted(range(len(groups)), key=lambda w: (collate_rand(seed, ex, RS_SHUFFLE, w), w))
    m = whole_word_mask(groups, order, n, mlm_probability) if n > 0 else []
    ids = np.full(L, pad_id, np.int64)
    labels = np.full(L, -100, np.int64)
    att = np.zeros(L, np.int64)
    ids[0] = cls_id; att[0] = 1
    for i, t in enumerate(toks):
        p = i + 1
        ids[p] = t; att[p] = 1
        if m[i]:
            labels[p] = t
            if _uniform(collate_rand(seed, ex, RS_REPLACE, p)) < np.float32(0.8):
                ids[p] = mask_id
            elif _uniform(collate_rand(seed, ex, RS_RANDOM, p)) < np.float32(0.5):
                ids[p] = collate_rand(seed, ex, RS_WORD, p) % V
    ids[n + 1] = sep_id; att[n + 1] = 1
    return ids, labels, att
