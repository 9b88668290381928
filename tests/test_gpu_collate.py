"""the on-device Condenser / coCondenser collator against its oracle (same counter-based generator ->
bit-exact), including over-long spans (random truncation window), one-token and empty spans, L = 512."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402
from cocodr_amd.collate import CoCondenserCollator, CondenserCollator, subword_flags_from_vocab  # noqa: E402
import oracle as O  # noqa: E402  (checker only)


def _vocab(V, rng):
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    toks += [("##" if rng.random() < 0.3 else "") + f"t{i}" for i in range(len(toks), V)]
    return toks


@pytest.mark.parametrize("L,seed", [(32, 1), (128, 2), (512, 3)])
def test_collator_matches_oracle_bit_for_bit(L, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    V = 3000
    flags = subword_flags_from_vocab(_vocab(V, rng))
    col = CondenserCollator(flags, max_seq_length=L, seed=seed * 11, mlm_probability=0.15)
    lens = [0, 1, 2, L - 3, L - 2, L - 1, L + 40, 3 * L] + [int(x) for x in rng.integers(1, 2 * L, 40)]
    spans = [rng.integers(100, V, n).tolist() for n in lens]  # ids 100-103: [UNK] / [CLS] / [SEP] / [MASK] occur INSIDE spans
    assert (flags[100:104] == 2).all() and any(t < 104 for s in spans for t in s)
    for rep in range(2):  # the second call continues the span counter
        base = col.spans_seen
        out = col([{"text": s} for s in spans])
        ids, labels, att = (out[k].cpu().numpy() for k in ("input_ids", "labels", "attention_mask"))
        assert ids.dtype == np.int64 and ids.shape == (len(spans), L)
        for i, s in enumerate(spans):
            ri, rl, ra = O.collate_span(s, flags, seed * 11, base + i, L, 101, 102, 0, 103, 0.15)
            assert np.array_equal(ids[i], ri) and np.array_equal(labels[i], rl) and np.array_equal(att[i], ra), (rep, i, len(s))
            m = min(len(s), L - 2)
            inside = np.asarray(ri[1:m + 1])
            assert (rl[1:m + 1][(inside >= 100) & (inside < 104) & (rl[1:m + 1] == -100)] == -100).all()
            assert not any(100 <= int(x) < 104 for x in rl[1:m + 1] if x != -100)  # a special token is never a prediction target


def test_cocondenser_collator_lays_span_pairs_back_to_back():
    rng = np.random.Generator(np.random.PCG64(5))
    flags = subword_flags_from_vocab(_vocab(2000, rng))
    col = CoCondenserCollator(flags, max_seq_length=64, seed=4)
    docs = [{"span": [rng.integers(104, 2000, 20).tolist(), rng.integers(104, 2000, 30).tolist()]} for _ in range(5)]
    out = col(docs)
    assert out["input_ids"].shape == (10, 64)
    att = out["attention_mask"].cpu().numpy()
    assert list(att.sum(1)) == [22, 32] * 5  # rows 2i / 2i+1 are the two spans of document i (COCO/modeling.py:172-177)
    assert set(out) == {"input_ids", "labels", "attention_mask"}  # the reference's batch, nothing else (COCO/data.py:150-154)
    # opt-in: the host-known lengths a model can pack from without reading anything back - a CPU tensor (= attention_mask.sum(1)),
    # so code that calls .split() on every value of the batch (COCO/trainer.py:137-140) cuts it with the rest
    col_l = CoCondenserCollator(flags, max_seq_length=64, seed=4, emit_lengths=True)
    out_l = col_l(docs)
    assert torch.is_tensor(out_l["lengths"]) and not out_l["lengths"].is_cuda and out_l["lengths"].tolist() == [22, 32] * 5
    assert torch.equal(out_l["input_ids"], out["input_ids"]) and [t.shape[0] for t in out_l["lengths"].split(4)] == [4, 4, 2]
    long_doc = [{"span": [rng.integers(104, 2000, 200).tolist(), rng.integers(104, 2000, 62).tolist()]}]
    o2 = col_l(long_doc)
    assert o2["lengths"].tolist() == o2["attention_mask"].sum(1).cpu().tolist() == [64, 64]  # truncated to max_seq_length - 2 (+ [CLS], [SEP])
    lab = out["labels"].cpu().numpy()
    assert ((lab != -100).sum(1) >= 1).all()
