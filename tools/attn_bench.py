import sys, os, torch
sys.path.insert(0, "/root/repo")
from cocodr_amd import ops
B, L, heads = 64, 128, 12
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
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    print(f"attn {name}: {best:.1f} us", flush=True)
