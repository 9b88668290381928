"""cocodr_pack_index / cocodr_mask_lengths (include/cocodr.h "Packed batches"): the packed-layout description of a padded batch,
one native launch from the padded ids and the B lengths.  Checked bit for bit against a plain numpy construction of the same
arrays, for host-known lengths (no read-back) and for the device-mask route, int32 and int64 inputs, strided rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cocodr_amd  # noqa: E402,F401
from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel, PackedIndex  # noqa: E402

DEV = "cuda"


def layout_numpy(ids, lens, Lp):
    """The arrays of cocodr_packed_batch, element by element."""
    B, L = ids.shape
    ext = np.maximum(lens, 1).astype(np.int64)  # every sequence on its own length; T rounded up to 32 on the last sequences with room
    pad, cap, j = int(-ext.sum() % 32), Lp, B - 1
    while pad > 0:
        give = min(pad, int(cap - ext[j]))
        ext[j] += give
        pad -= give
        j -= 1
    off = np.concatenate([[0], np.cumsum(ext)])
    T = int(off[-1])
    out = {k: np.zeros(T, np.int64) for k in ("ids", "positions", "mask", "cls_slot", "src")}
    for b in range(B):
        for p in range(ext[b]):
            r = off[b] + p
            out["positions"][r] = p
            out["mask"][r] = int(p < lens[b])
            out["ids"][r] = ids[b, p] if p < lens[b] else 0
            out["cls_slot"][r] = b if p == 0 else -1
            out["src"][r] = b * Lp + p
    return off, T, int((ext.max() + 31) // 32 * 32), out


def batch(lens, L, seed=0, V=30000):
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.asarray(lens, np.int64)
    ids = rng.integers(5, V, (len(lens), L))
    mask = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    return ids * mask, mask, lens


def check_index(pk, ids, lens, Lp):
    off, T, max_len, ref = layout_numpy(ids, lens, Lp)
    assert (pk.T, pk.max_len, pk.B, pk.L) == (T, max_len, ids.shape[0], Lp)
    assert pk.seq_off.cpu().tolist() == off.tolist()
    for k in ("ids", "positions", "mask", "cls_slot", "src"):
        assert np.array_equal(getattr(pk, k).cpu().numpy().astype(np.int64), ref[k]), k
    assert pk.cls_rows.cpu().tolist() == off[:-1].tolist()
    assert (pk.c_struct.T, pk.c_struct.B, pk.c_struct.max_len, pk.c_struct.drop_L) == (T, ids.shape[0], max_len, Lp)


@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("lens,L", [([64, 1, 33, 32, 7, 0], 64), ([128] * 3, 128), ([50, 3, 17], 50), ([1], 32), (list(range(0, 130, 3)), 130)])
def test_pack_index_from_host_lengths(lens, L, dtype):
    ids, mask, lens = batch(lens, L, seed=len(lens))
    Lp = (L + 31) // 32 * 32
    dids = torch.from_numpy(ids).to(dtype).to(DEV)
    for lengths in (lens, lens.tolist(), torch.from_numpy(lens), torch.from_numpy(lens).int()):
        check_index(PackedIndex.build(dids, None, lengths), ids, lens, Lp)
    check_index(PackedIndex.build(dids, None, torch.from_numpy(lens).to(DEV)), ids, lens, Lp)  # (device lengths: read back)


@pytest.mark.parametrize("mdtype", [torch.int64, torch.int32, torch.bool, torch.uint8, torch.float32])
def test_pack_index_from_the_device_mask(mdtype):
    ids, mask, lens = batch([40, 96, 3, 0, 64, 65], 96, seed=3)
    dids = torch.from_numpy(ids).to(DEV)
    dmask = torch.from_numpy(mask).to(mdtype).to(DEV)
    check_index(PackedIndex.build(dids, dmask), ids, lens, 96)
    # strided rows: a [B, 2L] buffer's left half
    wide_i = torch.zeros((6, 192), dtype=torch.int64, device=DEV)
    wide_m = torch.zeros((6, 192), dtype=mdtype, device=DEV)
    wide_i[:, :96], wide_m[:, :96] = dids, dmask
    wide_i[:, 96:], wide_m[:, 96:] = 7, 1
    check_index(PackedIndex.build(wide_i[:, :96], wide_m[:, :96]), ids, lens, 96)
    # no mask at all = every sequence fills L
    full = np.random.Generator(np.random.PCG64(9)).integers(5, 900, (3, 64))
    check_index(PackedIndex.build(torch.from_numpy(full).to(DEV), None), full, np.full(3, 64), 64)


@pytest.mark.parametrize("B,L", [(1, 32), (2, 33), (64, 128), (257, 64), (1500, 96), (4096, 32), (4100, 32)])
def test_device_planned_layout_equals_the_host_arithmetic(B, L):
    """cocodr_pack_plan (extents, offsets, longest-first order, T, longest extent) against packed_extents + numpy's stable argsort,
    bit for bit, on MS MARCO-shaped and degenerate lengths; B > 4096 takes the read-back route with the same result."""
    rng = np.random.Generator(np.random.PCG64(B * 1000 + L))
    for kind in ("marco", "all_full", "all_empty", "ties"):
        lens = {"marco": np.clip(np.rint(rng.normal(0.6 * L, 0.25 * L, B)), 0, L), "all_full": np.full(B, L), "all_empty": np.zeros(B),
                "ties": rng.integers(1, 4, B) * (L // 4)}[kind].astype(np.int64)
        ids, mask, lens = batch(lens, L, seed=B)
        dids, dmask = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
        dev, host = PackedIndex.build(dids, dmask), PackedIndex.build(dids, None, lens)
        assert (dev.T, dev.max_len, dev.B, dev.L) == (host.T, host.max_len, host.B, host.L), kind
        assert torch.equal(dev.seq_off, host.seq_off) and torch.equal(dev.seq_order, host.seq_order), kind
        for k in ("ids", "positions", "mask", "cls_slot", "src"):
            assert torch.equal(getattr(dev, k), getattr(host, k)), (kind, k)
        assert dev.T % 32 == 0 and sorted(dev.seq_order.cpu().tolist()) == list(range(B))


def test_host_lengths_that_contradict_the_mask_are_caught_by_the_debug_check():
    ids, mask, lens = batch([10, 20, 5], 32)
    d = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    PackedIndex.check_lengths = True
    try:
        assert PackedIndex.build(d(ids), d(mask), lens) is not None
        with pytest.raises(ValueError):
            PackedIndex.build(d(ids), d(mask), lens - 1)          # (lengths counted without [SEP])
        hole = mask.copy()
        hole[1, 3] = 0
        with pytest.raises(ValueError):
            PackedIndex.build(d(ids), d(hole), lens)
    finally:
        PackedIndex.check_lengths = False
    assert PackedIndex.build(d(ids), d(mask), lens - 1) is not None  # the fast path trusts the host


def test_padded_run_of_a_batch_given_by_lengths_alone_masks_the_padding():
    """pack_sequences = False with attention_mask = None and host lengths: the padded mask is built from the lengths."""
    cfg = CocoBertConfig(vocab_size=900, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
    torch.manual_seed(0)
    bert = CocoBertModel(cfg).to(DEV).eval()
    ids, mask, lens = batch([40, 64, 3, 17], 64, seed=2, V=900)
    with torch.no_grad():
        bert.pack_sequences = False
        a = bert.encode_cls(torch.from_numpy(ids).to(DEV), None, lengths=lens)
        b = bert.encode_cls(torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV))
        bert.pack_sequences = True
        c = bert.encode_cls(torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV))
    assert torch.equal(a, b) and torch.equal(b, c)


def test_masks_with_holes_or_leading_padding_are_not_packed():
    ids, mask, _ = batch([10, 20], 32)
    hole, lead = mask.copy(), mask.copy()
    hole[1, 4] = 0
    lead[0, 0] = 0
    d = lambda a: torch.from_numpy(a).to(DEV)  # noqa: E731
    assert PackedIndex.build(d(ids), d(hole)) is None and PackedIndex.build(d(ids), d(lead)) is None
    assert PackedIndex.build(d(ids), d(mask)) is not None


def test_forward_with_a_holey_mask_falls_back_to_the_padded_execution():
    """The default forward plans the packed layout on the device and learns only INSIDE the encoder call that a mask is not a prefix
    mask (PackedIndex.from_mask(lazy=True)): it must then run padded - same embeddings and gradients as pack_sequences = False - and
    draw exactly one dropout call / count one forward."""
    cfg = CocoBertConfig(vocab_size=900, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
    torch.manual_seed(0)
    bert = CocoBertModel(cfg).to(DEV)
    bert.dropout_seed = 7
    ids, mask, lens = batch([40, 64, 3, 17, 9, 33, 64, 20], 64, seed=2, V=900)
    mask[1, 5] = 0  # a hole
    dids, dmask = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    out = {}
    for packed in (True, False):
        bert.pack_sequences = packed
        bert.flat_decay.grad = bert.flat_nodecay.grad = None
        bert._dropout_calls = calls0 = 0  # (train() mode drops: both runs draw the masks of call 1)
        e = bert.encode_cls(dids, dmask)
        e.square().sum().backward()
        out[packed] = (e.detach().clone(), bert.flat_decay.grad.clone(), bert._dropout_calls - calls0)
    bert.pack_sequences = True
    assert torch.equal(out[True][0], out[False][0]) and out[True][2] == out[False][2]
    assert float((out[True][1] - out[False][1]).norm() / out[False][1].norm()) < 1e-6  # (word rows: fp32 atomics, free order)
    lazy = PackedIndex.build(dids, dmask, lazy=True)
    assert lazy is not None and lazy.resolve() is False
    with pytest.raises(ValueError):
        lazy.T


def test_unpack_is_the_inverse_of_the_row_map_and_differentiable():
    ids, mask, lens = batch([40, 96, 3], 96, seed=1)
    pk = PackedIndex.build(torch.from_numpy(ids).to(DEV), None, lens)
    x = torch.randn(pk.T, 8, device=DEV, requires_grad=True)
    y = pk.unpack(x)
    assert y.shape == (3, 96, 8)
    assert torch.equal(y.reshape(-1, 8)[pk.src], x)
    untouched = torch.ones(3 * 96, dtype=torch.bool, device=DEV)
    untouched[pk.src] = False
    assert not y.reshape(-1, 8)[untouched].any()      # rows past an extent are zeros
    (y * 2).sum().backward()
    assert torch.equal(x.grad, torch.full_like(x, 2.0))


def test_step_with_host_lengths_equals_step_from_the_mask_and_the_padded_step():
    """The three routes of one training step - packed with host lengths (no read-back), packed from the device mask, padded -
    give the same loss and gradients; a fresh batch every call, nothing prebuilt."""
    cfg = CocoBertConfig(vocab_size=900, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256)
    torch.manual_seed(0)
    bert = CocoBertModel(cfg).to(DEV)
    with torch.no_grad():
        bert.flat_decay.mul_(3.0)
    model = CoCondenserForPretraining(bert)
    assert bert.pack_sequences is True  # the default since round 4
    res = {}
    for step in range(2):
        ids, mask, lens = batch(np.random.Generator(np.random.PCG64(step)).integers(3, 64, 8), 64, seed=10 + step, V=900)
        for route in ("host", "mask", "padded"):
            bert.pack_sequences = route != "padded"
            b = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
            if route == "host":
                b["lengths"] = torch.from_numpy(lens)
            bert.flat_decay.grad = bert.flat_nodecay.grad = None
            loss = model(b, None)
            loss.backward()
            res[route] = (float(loss), bert.flat_decay.grad.clone(), bert.flat_nodecay.grad.clone())
        assert res["host"][0] == res["mask"][0] and torch.equal(res["host"][2], res["mask"][2])
        assert float((res["host"][1] - res["mask"][1]).norm() / res["mask"][1].norm()) < 1e-6  # (word rows: fp32 atomics, free order)
        assert abs(res["host"][0] - res["padded"][0]) < 1e-5 * abs(res["padded"][0])
        for k in (1, 2):
            d = float((res["host"][k] - res["padded"][k]).norm() / res["padded"][k].norm())
            assert d < 5e-3, (k, d)
    bert.pack_sequences = True
