"""Attention forward / backward at the encoder's shapes, us per launch (best of 5 x 10 back-to-back launches).
COCODR_ATTN_TWO_PHASE=1 selects the round-1..3 two-phase backward at L <= 128 (A/B of the one-pass kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cocodr_amd import ops  # noqa: E402

for B, L, heads in ((64, 128, 12), (256, 128, 16), (200, 128, 16), (32, 64, 16), (64, 64, 12)):
    H = heads * 64
    qkv = (torch.randn(B * L, 3 * H, device="cuda") * 0.5).to(torch.bfloat16)
    mask = torch.ones(B, L, dtype=torch.int32, device="cuda")
    dctx = torch.randn(B * L, H, device="cuda").to(torch.bfloat16)
    ctx, lse = ops.attn_fwd(qkv, mask, B, L, heads)
    for name, fn in (("fwd", lambda: ops.attn_fwd(qkv, mask, B, L, heads)), ("bwd", lambda: ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, heads)),
                     ("bwd + q/k bias partials", lambda: ops.attn_bwd(qkv, mask, ctx, dctx, lse, B, L, heads, qk_bias=True))):
        best = 1e9
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        print(f"B {B:4d} L {L:4d} heads {heads:3d}  attn {name}: {best:.1f} us", flush=True)
