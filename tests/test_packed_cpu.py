"""Host arithmetic of the packed (variable-length) batch layout (include/cocodr.h "Packed batches"): extents and row offsets
from the B lengths.  The device half (cocodr_pack_index, cocodr_mask_lengths) is checked in tests/test_gpu_pack_index.py."""
import numpy as np
import pytest
import torch

import cocodr_amd  # noqa: F401
from cocodr_amd.modeling import PackedIndex, packed_extents


def test_extents_and_offsets():
    """default layout (PACK_ALIGN = 1): every sequence on max(len, 1) rows; the rows that make T a multiple of 32 go to the last
    sequences that have room below ceil32(L)"""
    host, T, max_len = packed_extents([64, 1, 33, 32, 7, 0], 6, 64)
    assert host.dtype == np.int32 and host[:6].tolist() == [64, 1, 33, 32, 7, 0]
    assert host[6:].tolist() == [0, 64, 65, 98, 130, 137, 160] and (T, max_len) == (160, 64)   # 138 rows + 22 on the last (empty) sequence
    # no room on the last sequence: the rows go further up
    host, T, max_len = packed_extents([512, 500, 512], 3, 512)
    assert host[3:].tolist() == [0, 512, 1024, 1536] and (T, max_len) == (1536, 512)
    # the layout of rounds 3-4: every extent a multiple of 32
    host, T, max_len = packed_extents([64, 1, 33, 32, 7, 0], 6, 64, align=32)
    assert host[6:].tolist() == [0, 64, 96, 160, 192, 224, 256] and (T, max_len) == (256, 64)


def test_extents_properties_over_random_batches():
    rng = np.random.Generator(np.random.PCG64(0))
    for _ in range(500):
        B, L = int(rng.integers(1, 40)), int(rng.integers(1, 513))
        lens = rng.integers(0, L + 1, B)
        host, T, max_len = packed_extents(lens, B, L)
        ext = np.diff(host[B:])
        assert host[B] == 0 and T == host[-1] and T % 32 == 0 and T - np.maximum(lens, 1).sum() < 32
        assert (ext >= np.maximum(lens, 1)).all() and ext.max() <= (L + 31) // 32 * 32
        assert max_len % 32 == 0 and ext.max() <= max_len < ext.max() + 32


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
