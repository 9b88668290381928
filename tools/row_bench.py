#!/usr/bin/env python3
"""Micro-benchmark of the row kernels at the encoder's shape (8192 x 768 bf16).  Usage (GPU box): python tools/row_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa
from cocodr_amd import ops

M, H = 8192, (int(sys.argv[1]) if len(sys.argv) > 1 else 768)
y = torch.randn(M, H, device="cuda").to(torch.bfloat16)
d = torch.randn(M, H, device="cuda").to(torch.bfloat16)
g = torch.ones(H, device="cuda"); b = torch.zeros(H, device="cuda")
out, mean, rstd = ops.ln_fwd(y, g, b)
fns = {"ln_fwd": lambda: ops.ln_fwd(y, g, b), "ln_bwd(+colsum)": lambda: ops.ln_bwd(d, y, g, mean, rstd, colsum=True)}
for name, fn in fns.items():
    best = 1e9
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{name}: {best:.1f} us per call (incl. allocation + reduce launch)")
