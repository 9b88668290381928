"""Column sums of a bf16 matrix (bias gradients; the dV slice of the fused dQKV under attention dropout), us per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402
from cocodr_amd import ops  # noqa: E402

for M, N in ((8192, 1024), (2048, 1024), (25600, 1024), (8192, 3072), (1216, 30592)):
    x = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    ref = x.float().sum(0)
    got = ops.colsum(x)
    err = float((got - ref).abs().max() / ref.abs().max())
    for _ in range(3):
        ops.colsum(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.colsum(x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{M} x {N}: {us:.1f} us (incl. the partial reduction), {M * N * 2 / us / 1e6:.2f} TB/s, max rel err {err:.1e}")
