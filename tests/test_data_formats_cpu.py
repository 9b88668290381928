"""On-disk formats: the reader must parse bytes written by the reference's record formula to exactly what
the reference's EmbeddingCache returns (golden), and our writer must emit those same bytes."""
import json
import os
import pickle

import numpy as np
import torch

import cocodr_amd
from cocodr_amd import data as D
from conftest import load_golden


def test_token_cache_reader_matches_reference_embedding_cache(tmp_path):
    g = load_golden("token_cache.npz")
    L = int(g["max_len"])
    path = str(tmp_path / "passages")
    with open(path, "wb") as f:
        f.write(g["blob"].tobytes())
    with open(path + "_meta", "w") as f:
        json.dump({"type": "int32", "total_number": len(g["lengths"]), "embedding_size": L}, f)
    cache = D.TokenCache(path)
    assert len(cache) == len(g["lengths"])
    for i in range(len(cache)):
        ln, toks = cache[i]
        assert ln == int(g["lengths"][i]) and np.array_equal(toks, g["tokens"][i])
    ids, mask, idx = cache.batch([3, 0, 7])
    assert ids.dtype == torch.int64 and np.array_equal(ids.numpy(), g["tokens"][[3, 0, 7]])
    assert mask.sum(1).tolist() == [int(g["lengths"][i]) for i in (3, 0, 7)] and idx.tolist() == [3, 0, 7]
    assert bool((mask[:, :-1] >= mask[:, 1:]).all())  # ones first, then zeros


def test_token_cache_writer_emits_reference_bytes(tmp_path):
    g = load_golden("token_cache.npz")
    L = int(g["max_len"])
    lists = [g["tokens"][i][: int(g["lengths"][i])].tolist() for i in range(len(g["lengths"]))]
    path = str(tmp_path / "out")
    assert D.write_token_cache(path, lists, L) == len(lists)
    assert open(path, "rb").read() == g["blob"].tobytes()
    # over-long inputs are truncated like tokenizer.encode(max_length=...)
    D.write_token_cache(path, [list(range(1, 40))], L)
    ln, toks = D.TokenCache(path)[0]
    assert ln == L and toks.tolist() == list(range(1, L + 1))


def test_embedding_shards_roundtrip_rank_major_and_reference_pickle_protocol(tmp_path):
    rng = np.random.Generator(np.random.PCG64(0))
    embs = [rng.standard_normal((n, 8)).astype(np.float32) for n in (5, 4, 4)]
    ids = [np.arange(r, 13, 3) for r in range(3)]
    for r in range(3):
        D.save_embedding_shard(str(tmp_path), "passage_0", r, torch.from_numpy(embs[r]), torch.from_numpy(ids[r]))
    with open(os.path.join(str(tmp_path), "passage_0__emb_p__data_obj_1.pb"), "rb") as h:
        raw = h.read()
    assert raw[:2] == b"\x80\x04" and np.array_equal(pickle.loads(raw), embs[1])  # protocol 4, plain ndarray
    E, I = D.load_embedding_shards(str(tmp_path), "passage_0")
    assert np.array_equal(E, np.concatenate(embs)) and I.tolist() == [0, 3, 6, 9, 12, 1, 4, 7, 10, 2, 5, 8, 11]


def test_triplet_file_format(tmp_path):
    path = str(tmp_path / "ann_training_data_0")
    pos = {7: 70, 9: 90}
    negs = {7: list(range(100, 110)), 9: list(range(200, 210))}
    n = D.write_triplets(path, [9, 7, 5], pos, negs)
    rows = D.read_triplets(path)
    assert n == len(rows) == 10 and rows[0] == (9, 90, [200, 201]) and rows[1] == (7, 70, [100, 101])
    assert rows[-1] == (7, 70, [108, 109])
    assert open(path).readline() == "9\t90\t200,201\n"


def _caches_from_golden(g, tmp_path):
    out = []
    for name, blob, L in (("queries", g["q_blob"], int(g["Lq"])), ("passages", g["p_blob"], int(g["Lp"]))):
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            f.write(blob.tobytes())
        with open(path + "_meta", "w") as f:
            json.dump({"type": "int32", "total_number": len(blob) // (4 * L + 4), "embedding_size": L}, f)
        out.append(D.TokenCache(path))
    return out


def test_training_stream_matches_reference_processing_fns(tmp_path):
    """ANCE/data/msmarco_data.py:328-384 over ANCE/utils/util.py:372-399 + DataLoader(batch_size): the reference's own
    functions produced tests/golden/training_rows.npz; the stream here must give the same batches on every rank."""
    g = load_golden("training_rows.npz")
    qc, pc = _caches_from_golden(g, tmp_path)
    lines = [str(x) for x in g["lines"]]
    bs = int(g["batch_size"])
    keys = (("query_ids", "q_ids"), ("attention_mask_q", "q_mask"), ("input_ids_a", "a_ids"), ("attention_mask_a", "a_mask"),
            ("input_ids_b", "b_ids"), ("attention_mask_b", "b_mask"))
    for world in (1, 2):
        seen = 0
        for rank in range(world):
            stream = D.TripletStream(lines, qc, pc, bs, rank=rank, world_size=world)
            batches = list(stream)
            assert len(batches) == len(stream) == int(g[f"trip_w{world}_r{rank}_nb"])
            for bi, b in enumerate(batches):
                for ours, theirs in keys:
                    ref = g[f"trip_w{world}_r{rank}_b{bi}_{theirs}"]
                    assert b[ours].dtype == torch.int64 and np.array_equal(b[ours].numpy(), ref), (world, rank, bi, ours)
                seen += b["query_ids"].shape[0]
        assert seen == sum(len(l.rstrip("\n").split("\t")[2].split(",")) for l in lines)  # every negative exactly once
    # the pairwise form: positive pair (label 1), then negative pair (label 0), per negative
    rows = D.pair_records(lines)
    assert rows[:, 2].tolist() == g["pair_label"].tolist()
    q_ids, _, _ = qc.batch(rows[:, 0])
    p_ids, p_mask, _ = pc.batch(rows[:, 1])
    assert np.array_equal(q_ids.numpy(), g["pair_q_ids"]) and np.array_equal(p_ids.numpy(), g["pair_p_ids"])
    assert np.array_equal(p_mask.numpy(), g["pair_p_mask"])


def test_training_stream_edge_cases(tmp_path):
    g = load_golden("training_rows.npz")
    qc, pc = _caches_from_golden(g, tmp_path)
    assert D.triplet_records([]).shape == (0, 3) and list(D.TripletStream([], qc, pc, 4)) == []
    assert D.triplet_records(["3\t5\t\n", "1\t2\t7\n"]).tolist() == [[1, 2, 7]]  # a line without negatives yields nothing
    assert D.triplet_records(["1\t2\t7,8\n", "3\t4\t9\n"], rank=1, world_size=2).tolist() == [[3, 4, 9]]
    try:
        D.TripletStream([], qc, pc, 0)
        raise AssertionError("batch_size 0 accepted")
    except ValueError:
        pass
    try:
        list(D.TripletStream(["1\t2\t999\n"], qc, pc, 2))  # a pid outside the cache
        raise AssertionError("out-of-range pid accepted")
    except IndexError:
        pass


def test_cocondenser_dataset_pairs_match_reference_sampling():
    """COCO/data.py:169-183: the reference's own CoCondenserDataset under random.seed(1234) chose these span pairs
    (tests/golden/coco_dataset.npz); one-span documents give the span twice."""
    import random
    from cocodr_amd.collate import CoCondenserDataset
    g = load_golden("coco_dataset.npz")
    spans = np.split(g["span_tokens"], np.cumsum(g["span_lens"])[:-1])
    docs, o = [], 0
    for n in g["doc_spans"]:
        docs.append({"spans": [s.tolist() for s in spans[o:o + int(n)]]})
        o += int(n)
    ds = CoCondenserDataset(docs)
    assert len(ds) == len(docs)
    random.seed(int(g["seed"]))
    got = [s for _epoch in range(2) for i in range(len(ds)) for s in ds[i]["span"]]
    want = np.split(g["pick_tokens"], np.cumsum(g["pick_lens"])[:-1])
    assert len(got) == len(want) == 4 * len(docs)
    for a, b in zip(got, want):
        assert list(a) == b.tolist()
    one = ds[0]["span"]
    assert one[0] == one[1] == docs[0]["spans"][0]
