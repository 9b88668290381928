"""LayerNorm forward (ops.ln_fwd) at the encoder's row counts, us per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import ops  # noqa: E402

for M, H in ((4768, 768), (8192, 768), (18944, 1024), (32768, 1024), (77056, 1024)):
    y = torch.randn(M, H).to(torch.bfloat16).cuda()
    gamma, beta = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda")
    for _ in range(3):
        ops.ln_fwd(y, gamma, beta)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.ln_fwd(y, gamma, beta)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"{M} x {H}: {us:.1f} us ({M * H * 4 / us / 1e6:.2f} TB/s)")
