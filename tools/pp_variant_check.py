#!/usr/bin/env python3
"""Bit-compare a ping-pong schedule variant (experiment build, e.g. impl 19) against the shipped schedule (impl 13) on the three
forms: same arithmetic in the same order, so every output must be identical.  Usage: python tools/pp_variant_check.py 19"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import ops  # noqa: E402

var = int(sys.argv[1]) if len(sys.argv) > 1 else 19
g = torch.Generator().manual_seed(0)
ok = True
base = int(sys.argv[2]) if len(sys.argv) > 2 else 13
for (M, N, K, ta, tb, f32, nb) in [(2048, 1024, 1024, 0, 0, 0, 1), (2000, 512, 64, 0, 0, 0, 1), (2048, 768, 128, 0, 1, 0, 1), (1024, 1024, 4096, 0, 1, 0, 1),
                                   (1024, 1024, 192, 0, 0, 0, 1), (512, 1024, 2048, 1, 1, 1, 6), (768, 256, 8192, 1, 1, 1, 3)]:
    ash = (K, M) if ta else (M, K)
    bsh = (K, N) if tb else (N, K)
    if nb > 1:
        ash, bsh = (nb,) + ash, (nb,) + bsh
    a = torch.randn(ash, generator=g).to(torch.bfloat16).cuda()
    b = (torch.randn(bsh, generator=g) * 0.05).to(torch.bfloat16).cuda()
    outs = []
    for impl in (base, var):
        ops.gemm_set_impl(impl)
        outs.append(ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out_f32=bool(f32)).clone())
    same = torch.equal(outs[0], outs[1])
    ok &= same
    print(f"M={M} N={N} K={K} ta={ta} tb={tb} batch={nb}: {'identical' if same else 'DIFFERENT'}", flush=True)
ops.gemm_set_impl(0)
sys.exit(0 if ok else 1)
