"""Host arithmetic of the packed (variable-length) batch layout (include/cocodr.h "Packed batches"): extents and row offsets
from the B lengths.  The device half (cocodr_pack_index, cocodr_mask_lengths) is checked in tests/test_gpu_pack_index.py."""
import numpy as np
import pytest
import torch

import cocodr_amd  # noqa: F401
from cocodr_amd.modeling import PackedIndex, packed_extents


def test_extents_and_offsets():
    host, T, max_len = packed_extents([64, 1, 33, 32, 7, 0], 6, 64)
    assert host.dtype == np.int32 and host[:6].tolist() == [64, 1, 33, 32, 7, 0]
    assert host[6:].tolist() == [0, 64, 96, 160, 192, 224, 256] and (T, max_len) == (256, 64)
    for n, o, o1 in zip([64, 1, 33, 32, 7, 0], host[6:-1], host[7:]):
        e = o1 - o
        assert e % 32 == 0 and e >= max(n, 1) and e - max(n, 1) < 32   # a fully masked sequence keeps one (masked) block


def test_lengths_from_a_tensor_list_or_array_agree():
    a, _, _ = packed_extents(torch.tensor([5, 40, 128]), 3, 128)
    b, _, _ = packed_extents([5, 40, 128], 3, 128)
    c, _, _ = packed_extents(np.array([5, 40, 128], np.int32), 3, 128)
    assert np.array_equal(a, b) and np.array_equal(b, c)


@pytest.mark.parametrize("bad", [[5, 40], [5, 40, 129], [-1, 2, 3]])
def test_bad_lengths_raise(bad):
    with pytest.raises(ValueError):
        packed_extents(bad, 3, 128)


def test_the_layout_needs_the_device():
    ids = torch.zeros((2, 32), dtype=torch.int32)
    with pytest.raises(RuntimeError):  # no CPU fallback: the layout arrays are written by the native kernel
        PackedIndex.build(ids, torch.ones_like(ids))
    with pytest.raises(ValueError):
        PackedIndex.build(ids.float(), None, [3, 4])
