"""numpy restatement of the dropout masks of the native path (include/cocodr.h "Dropout").  TEST INFRASTRUCTURE ONLY.

What is restated: WHERE the reference drops - the nn.Dropout modules of transformers' BertEmbeddings (hf:68-108, on the
LayerNorm output), eager_attention_forward (hf:111-203, on the softmax probabilities), BertSelfOutput (hf:282-293) and
BertOutput (hf:338-351) (on the dense outputs, in front of the residual add), active under model.train()
(ANCE/drivers/run_ann.py:293; the c_head layers of COCO/modeling.py:212-220) - and the inverted-dropout arithmetic
(kept elements x 1 / (1 - p)).  The placement is pinned against transformers' BertModel in train() mode driven with these
masks (tests/golden/dropout_sites.npz, made by tests/golden/make_golden.py).  WHICH elements drop cannot follow torch's
Philox stream; the mask is the counter-based hash the header specifies, restated here bit for bit.
"""
from __future__ import annotations

import numpy as np

KIND_ATTN_PROBS, KIND_ATTN_OUT, KIND_FFN_OUT, KIND_EMBED = 0, 1, 2, 3
_M64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def site_keys(seed: int, call: int, layer: int, kind: int):
    """cocodr_dropout_mask_for: (k0, k1) of one site of one forward call."""
    site = 4 * layer + kind
    z = splitmix64((seed + 0x9E3779B97F4A7C15 * (call + 1)) & _M64)
    z = splitmix64(z ^ ((0xD1B54A32D192ED03 * (site + 1)) & _M64))
    return z & 0xFFFFFFFF, z >> 32


def threshold_scale(p: float):
    thr = min(int(p * 65536.0 + 0.5), 65535)
    return thr, np.float32(65536.0) / np.float32(65536 - thr)


def lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def keep_mask(shape, p: float, seed: int, call: int, layer: int, kind: int) -> np.ndarray:
    """bool array of ``shape``: element with row-major flat index i is kept iff its 16-bit value >= round(p * 65536)."""
    thr, _ = threshold_scale(p)
    n = int(np.prod(shape))
    if thr == 0:
        return np.ones(shape, bool)
    k0, k1 = site_keys(seed, call, layer, kind)
    i = np.arange(n, dtype=np.uint64)
    w = lowbias32((i >> np.uint64(1)).astype(np.uint32) ^ np.uint32(k0)) ^ np.uint32(k1)
    u = np.where((i & np.uint64(1)) != 0, w >> np.uint32(16), w & np.uint32(0xFFFF))
    return (u >= np.uint32(thr)).reshape(shape)


def multiplier(shape, p: float, seed: int, call: int, layer: int, kind: int, dtype=np.float32) -> np.ndarray:
    """keep_mask x scale: what the dropped tensor is multiplied with, element-wise (forward and backward)."""
    _, scale = threshold_scale(p)
    return keep_mask(shape, p, seed, call, layer, kind).astype(dtype) * dtype(scale)
