#!/usr/bin/env python3
"""Persistent walk of the ping-pong GEMM (COCODR_PP_PERSIST=1, read when the library first launches a ping-pong form): bit-compare
against the one-barrier pipeline (impl 9, same arithmetic in the same order, not affected by the switch) on forward / dgrad forms
with every fused epilogue, whole and ragged row counts, 2 ... 7 tiles per workgroup.  Run with COCODR_PP_PERSIST=1; repeats each
case to catch a race between a tile's epilogue and the next tile's prefetch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import _native as N, ops  # noqa: E402

assert os.environ.get("COCODR_PP_PERSIST") == "1", "run with COCODR_PP_PERSIST=1"
CASES = [  # M, N, K, trans_b, epilogue, bias, residual
    (25600, 3072, 1024, 0, "none", True, False), (25600, 1024, 1024, 0, "add", True, True), (25600, 4096, 1024, 0, "gelu", True, False),
    (25600, 1024, 4096, 0, "add", True, True), (25600, 4096, 1024, 1, "dgelu", False, True), (25600, 1024, 4096, 1, "add", False, True),
    (25600, 1024, 3072, 1, "add", False, True), (17896, 3072, 1024, 0, "none", True, False), (17896, 1024, 4096, 0, "add", True, True),
    (17896, 4096, 1024, 0, "gelu", True, False), (17896, 1024, 3072, 1, "add", False, True), (32768, 2304, 768, 0, "none", True, False),
    (32768, 768, 3072, 0, "add", True, True), (65536, 2304, 768, 0, "none", False, False), (16640, 1024, 256, 0, "none", True, False),
    (25600, 3072, 1024, 0, "none", False, False),
]
codes = {"none": N.EPI_NONE, "add": N.EPI_ADD, "gelu": N.EPI_GELU, "dgelu": N.EPI_DGELU}
g = torch.Generator().manual_seed(0)
ok = True
for (M, Nn, K, tb, epi, has_bias, has_r) in CASES:
    a = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
    b = (torch.randn((K, Nn) if tb else (Nn, K), generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(Nn, generator=g).cuda() if has_bias else None
    r = torch.randn((M, Nn), generator=g).to(torch.bfloat16).cuda() if has_r else None
    ref = None
    same = True
    for impl, reps in ((9, 1), (13, 4)):
        ops.gemm_set_impl(impl)
        for _ in range(reps):
            res = ops.gemm(a, b, trans_b=bool(tb), bias=bias, epi=codes[epi], r=r)
            res = res if isinstance(res, tuple) else (res,)
            if ref is None:
                ref = [x.clone() for x in res]
            else:
                same &= all(torch.equal(x, y) for x, y in zip(ref, res))
    ok &= same
    tiles = ((M + 255) // 256) * (Nn // 256)
    print(f"M={M} N={Nn} K={K} tb={tb} {epi:5s} bias={int(has_bias)} r={int(has_r)} tiles={tiles}: {'identical' if same else 'DIFFERENT'}", flush=True)
ops.gemm_set_impl(0)
sys.exit(0 if ok else 1)
