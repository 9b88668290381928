"""Host logic of the packed (variable-length) batch layout (include/cocodr.h "Packed batches"): extents, offsets, position ids,
the padded -> packed row map and its inverse.  CPU only (the index arithmetic is plain torch)."""
import numpy as np
import torch

import cocodr_amd  # noqa: F401
from cocodr_amd.modeling import PackedIndex


def _batch(lens, L, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.asarray(lens)
    ids = rng.integers(5, 900, (len(lens), L))
    mask = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    return torch.from_numpy(ids * mask).int(), torch.from_numpy(mask).int()


def test_extents_offsets_positions_and_masks():
    ids, mask = _batch([64, 1, 33, 32, 7, 0], 64)
    pk = PackedIndex.build(ids, mask)
    assert pk.seq_off.tolist() == [0, 64, 96, 160, 192, 224, 256] and (pk.T, pk.max_len, pk.B, pk.L) == (256, 64, 6, 64)
    pos, m, slot = pk.positions.numpy(), pk.mask.numpy(), pk.cls_slot.numpy()
    for b, (o, n) in enumerate(zip(pk.seq_off[:-1].tolist(), [64, 1, 33, 32, 7, 0])):
        e = pk.seq_off[b + 1].item() - o
        assert e % 32 == 0 and e >= max(n, 1) and e - max(n, 1) < 32
        assert pos[o:o + e].tolist() == list(range(e))
        assert m[o:o + n].all() and not m[o + n:o + e].any()       # alignment rows (and a fully masked sequence) are masked keys
        assert slot[o] == b and (slot[o + 1:o + e] == -1).all()
        assert np.array_equal(pk.ids.numpy()[o:o + n], ids[b, :n].numpy()) and not pk.ids.numpy()[o + n:o + e].any()
    assert pk.c_struct.T == 256 and pk.c_struct.B == 6 and pk.c_struct.drop_L == 64


def test_unpack_is_the_inverse_of_the_row_map_and_differentiable():
    ids, mask = _batch([40, 96, 3], 96, seed=1)
    pk = PackedIndex.build(ids, mask)
    x = torch.randn(pk.T, 8, requires_grad=True)
    y = pk.unpack(x)
    assert y.shape == (3, 96, 8)
    assert torch.equal(y.reshape(-1, 8)[pk.src], x)
    untouched = torch.ones(3 * 96, dtype=torch.bool)
    untouched[pk.src] = False
    assert not y.reshape(-1, 8)[untouched].any()      # rows past an extent are zeros
    (y * 2).sum().backward()
    assert torch.equal(x.grad, torch.full_like(x, 2.0))


def test_masks_with_holes_or_leading_padding_are_not_packed():
    ids, mask = _batch([10, 20], 32)
    hole = mask.clone(); hole[1, 4] = 0
    lead = mask.clone(); lead[0, 0] = 0
    assert PackedIndex.build(ids, hole) is None and PackedIndex.build(ids, lead) is None
    assert PackedIndex.build(ids, mask) is not None
