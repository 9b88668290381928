#!/usr/bin/env python3
"""Ten-slot operand ring of the ping-pong GEMM (COCODR_PP_RING=10, read at the library's first ping-pong launch): bit-compare
against the one-barrier pipeline (impl 9: same arithmetic in the same order, not affected by the switch) on all three forms,
1 ... 64 K-tiles (the ring's period is 5 K-tiles), whole and ragged row counts, fused epilogues, batched fp32-out weight gradients.
Run with COCODR_PP_RING=10."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import _native as N, ops  # noqa: E402

assert os.environ.get("COCODR_PP_RING") == "10", "run with COCODR_PP_RING=10"
codes = {"none": N.EPI_NONE, "add": N.EPI_ADD, "gelu": N.EPI_GELU, "dgelu": N.EPI_DGELU}
g = torch.Generator().manual_seed(0)
ok = True
CASES = []  # M, N, K, ta, tb, batch, f32, epi, bias, r
for K in (64, 128, 192, 256, 320, 384, 448, 704, 1024, 4096):
    CASES.append((1024, 512, K, 0, 0, 1, 0, "none", False, False))
    CASES.append((1000, 256, K, 0, 1, 1, 0, "none", False, False))
CASES += [(25600, 1024, 1024, 0, 0, 1, 0, "add", True, True), (25600, 4096, 1024, 0, 0, 1, 0, "gelu", True, False), (17896, 1024, 4096, 0, 1, 1, 0, "add", False, True),
          (17896, 4096, 1024, 0, 1, 1, 0, "dgelu", False, True), (768, 256, 8192, 1, 1, 3, 1, "none", False, False), (1024, 1024, 5000, 1, 1, 2, 1, "none", False, False),
          (3072, 1024, 8192, 1, 1, 6, 1, "none", False, False)]
for (M, Nn, K, ta, tb, nb, f32, epi, has_bias, has_r) in CASES:
    ash = (K, M) if ta else (M, K)
    bsh = (K, Nn) if tb else (Nn, K)
    if nb > 1:
        ash, bsh = (nb,) + ash, (nb,) + bsh
    a = torch.randn(ash, generator=g).to(torch.bfloat16).cuda()
    b = (torch.randn(bsh, generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(Nn, generator=g).cuda() if has_bias else None
    r = torch.randn((M, Nn), generator=g).to(torch.bfloat16).cuda() if has_r else None
    ref, same = None, True
    for impl, reps in ((9, 1), (13, 3)):
        ops.gemm_set_impl(impl)
        for _ in range(reps):
            res = ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), bias=bias, epi=codes[epi], r=r, out_f32=bool(f32))
            res = res if isinstance(res, tuple) else (res,)
            if ref is None:
                ref = [x.clone() for x in res]
            else:
                same &= all(torch.equal(x, y) for x, y in zip(ref, res))
    ok &= same
    print(f"M={M} N={Nn} K={K} ta={ta} tb={tb} batch={nb} f32={f32} {epi}: {'identical' if same else 'DIFFERENT'}", flush=True)
ops.gemm_set_impl(0)
sys.exit(0 if ok else 1)
