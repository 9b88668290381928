"""On-device Condenser / coCondenser collators (COCO/data.py:24-168): the batch is truncated, whole-word
masked, wrapped in [CLS] / [SEP], padded and 80/10/10-replaced by one kernel (``cocodr_mlm_collate``) instead of 32
DataLoader workers; only the ragged token lists cross PCIe."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ._native import check, lib, ptr, stream_ptr

__all__ = ["CondenserCollator", "CoCondenserCollator", "CoCondenserDataset", "subword_flags_from_vocab"]


BERT_SPECIALS = ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")  # BertTokenizer.all_special_tokens


def subword_flags_from_vocab(tokens: Sequence[str], specials: Optional[Sequence[str]] = BERT_SPECIALS) -> np.ndarray:
    """uint8 [V] vocabulary classes: 1 where the WordPiece token continues a word (``token.startswith('##')``,
    COCO/data.py:50), 2 for the tokenizer's special tokens (skipped by the word grouping, never masked, :47-48), else 0."""
    sp = set(specials or ())
    return np.fromiter((2 if t in sp else (1 if t.startswith("##") else 0) for t in tokens), dtype=np.uint8, count=len(tokens))


class CondenserCollator:
    """``CondenserCollator`` (COCO/data.py:24-156) for BERT vocabularies.  ``__call__(examples)`` takes the reference's
    ``[{'text': [token ids]}, ...]`` and returns ``{"input_ids", "labels", "attention_mask"}`` (int64 CUDA tensors
    ``[n, max_seq_length]``).  ``seed`` + a running span counter drive the counter-based generator, so a run is
    reproducible and independent of the batch composition.  ``emit_lengths=True`` adds ``"lengths"``: the attended length of every
    row as a CPU int64 tensor (a boundary extension the reference's batch does not have - ``split_tensor_dict``-style code that
    calls ``.split`` on every value, COCO/trainer.py:137-140, still works on it and cuts it consistently; the model then lays its packed
    batch out without reading anything back).  Default off: the model plans the layout on the device from the mask at < 2 % cost."""

    def __init__(self, subword_flags: np.ndarray, cls_id: int = 101, sep_id: int = 102, pad_id: int = 0, mask_id: int = 103,
                 mlm_probability: float = 0.15, max_seq_length: int = 512, seed: int = 0, device="cuda", emit_lengths: bool = False):
        self.emit_lengths = bool(emit_lengths)
        self.device = torch.device(device)
        self.flags = torch.from_numpy(np.ascontiguousarray(subword_flags, dtype=np.uint8)).to(self.device)
        self.vocab = int(self.flags.numel())
        self.cls_id, self.sep_id, self.pad_id, self.mask_id = int(cls_id), int(sep_id), int(pad_id), int(mask_id)
        self.mlm_probability, self.max_seq_length, self.seed = float(mlm_probability), int(max_seq_length), int(seed)
        self.spans_seen = 0

    @classmethod
    def from_tokenizer(cls, tokenizer, **kw):
        vocab = [tokenizer.convert_ids_to_tokens(i) for i in range(len(tokenizer))]
        return cls(subword_flags_from_vocab(vocab, tokenizer.all_special_tokens), cls_id=tokenizer.cls_token_id, sep_id=tokenizer.sep_token_id,
                   pad_id=tokenizer.pad_token_id, mask_id=tokenizer.mask_token_id, **kw)

    def collate_spans(self, spans: Sequence[Sequence[int]]) -> Dict[str, torch.Tensor]:
        n = len(spans)
        if n == 0:
            raise ValueError("collate: empty batch")
        lens = np.fromiter((len(s) for s in spans), dtype=np.int64, count=n)
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=offsets[1:])
        flat = np.empty(max(int(offsets[-1]), 1), np.int32)
        for i, s in enumerate(spans):
            flat[offsets[i]:offsets[i + 1]] = s
        tok = torch.from_numpy(flat).to(self.device, non_blocking=True)
        off = torch.from_numpy(offsets).to(self.device, non_blocking=True)
        L = self.max_seq_length
        ids = torch.empty((n, L), dtype=torch.int32, device=self.device)
        labels = torch.empty_like(ids)
        mask = torch.empty_like(ids)
        check(lib().cocodr_mlm_collate(ptr(tok), ptr(off), n, ptr(self.flags), self.vocab, L, self.cls_id, self.sep_id, self.pad_id,
                                       self.mask_id, self.mlm_probability, self.seed, self.spans_seen, ptr(ids), ptr(labels), ptr(mask),
                                       stream_ptr()), "mlm_collate")
        self.spans_seen += n
        out = {"input_ids": ids.long(), "labels": labels.long(), "attention_mask": mask.long()}  # the reference's batch (COCO/data.py:150-154)
        if self.emit_lengths:  # truncation window + [CLS] + [SEP] (COCO/data.py:131-144), known here on the host
            out["lengths"] = torch.from_numpy(np.minimum(lens, L - 2) + 2).to(torch.int64)
        return out

    def __call__(self, examples: List[Dict[str, List[int]]]):
        return self.collate_spans([e["text"] for e in examples])


class CoCondenserCollator(CondenserCollator):
    """``CoCondenserCollator`` (COCO/data.py:159-166): every example carries two spans of one document; they are laid
    out back to back so rows 2i / 2i+1 are the positive pair ``co_target`` expects."""

    def __call__(self, examples):
        spans = [s for e in examples for s in e["span"]]
        return self.collate_spans(spans)


class CoCondenserDataset(torch.utils.data.Dataset):
    """``CoCondenserDataset`` (COCO/data.py:169-183): item i = two spans of document i - the single span twice when the
    document has only one, otherwise ``random.sample(spans, 2)`` (Python's global ``random``, as the reference: seed it with
    ``random.seed`` / the trainer's ``set_seed`` for reproducible pairs).  ``dataset[i]["spans"]`` is a list of token-id lists."""

    def __init__(self, dataset, data_args=None):
        self.dataset = dataset
        self.data_args = data_args

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, item):
        import random
        spans = self.dataset[item]["spans"]
        if len(spans) == 1:
            return {"span": spans + spans}
        return {"span": random.sample(spans, 2)}
