"""LayerNorm backward with / without the dropout form at the encoder's row counts (ops.ln_bwd), us per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402
from cocodr_amd import ops  # noqa: E402

for M, H in ((8192, 1024), (2048, 1024), (25600, 1024), (8192, 768)):
    g = torch.Generator().manual_seed(0)
    dout = torch.randn(M, H, generator=g).to(torch.bfloat16).cuda()
    y = torch.randn(M, H, generator=g).to(torch.bfloat16).cuda()
    gamma = torch.ones(H, device="cuda")
    mean = torch.zeros(M, device="cuda")
    rstd = torch.ones(M, device="cuda")
    dm = ops.dropout_mask(0.1, 1, 1, 0, ops.KIND_FFN_OUT)
    for name, kw in (("plain", {}), ("dropout", {"drop": dm})):
        for _ in range(3):
            ops.ln_bwd(dout, y, gamma, mean, rstd, colsum=True, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.ln_bwd(dout, y, gamma, mean, rstd, colsum=True, **kw)
        e1.record()
        torch.cuda.synchronize()
        print(f"{M} x {H} {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us (incl. the partial reduction)")
