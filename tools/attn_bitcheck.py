#!/usr/bin/env python3
"""Bit-compare two builds of csrc/attention.hip (tools/_abl/libattn_old.so = the committed source, libattn_new.so = the working
tree) on forward and backward, plain / masked / dropout, L = 32 ... 256, and time both.  A schedule change must not move a bit."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cocodr_amd import _native as N  # noqa: E402

libs = {k: C.CDLL(os.path.join(ROOT, "tools", "_abl", f"libattn_{k}.so")) for k in ("old", "new")}
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
ok = True
heads = 12
H = heads * 64
for (B, L, masked, drop) in [(64, 128, False, 0.0), (64, 128, True, 0.0), (16, 32, True, 0.0), (8, 96, True, 0.1), (8, 256, True, 0.0), (64, 128, True, 0.1), (200, 128, True, 0.0)]:
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + L)
    qkv = (torch.randn(B * L, 3 * H, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    mask = torch.ones(B, L, dtype=torch.int32, device="cuda")
    if masked:
        lens = torch.randint(1, L + 1, (B,), device="cuda", generator=g)
        mask = (torch.arange(L, device="cuda")[None, :] < lens[:, None]).to(torch.int32)
    dctx = torch.randn(B * L, H, device="cuda", generator=g).to(torch.bfloat16)
    dm = N.DropoutMask()
    if drop > 0:
        assert libs["new"].cocodr_dropout_mask_for(C.c_double(drop), C.c_ulonglong(7), C.c_ulonglong(1), 3, 1, C.byref(dm)) == 0
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {}
    for k, lib in libs.items():
        ctx = torch.zeros(B * L, H, dtype=torch.bfloat16, device="cuda")
        lse = torch.zeros(B, heads, L, dtype=torch.float32, device="cuda")
        dqkv = torch.zeros_like(qkv)
        part = torch.zeros(B * 4 * 2 * H, dtype=torch.float32, device="cuda")
        assert lib.cocodr_attn_fwd_drop(p(qkv), p(mask), p(ctx), p(lse), B, L, heads, C.byref(dm), st) == 0
        assert lib.cocodr_attn_bwd_drop(p(qkv), p(mask), p(ctx), p(dctx), p(lse), p(dqkv), p(part), B, L, heads, C.byref(dm), st) == 0
        torch.cuda.synchronize()
        t = []
        for fn in ("fwd", "bwd"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                if fn == "fwd":
                    lib.cocodr_attn_fwd_drop(p(qkv), p(mask), p(ctx), p(lse), B, L, heads, C.byref(dm), st)
                else:
                    lib.cocodr_attn_bwd_drop(p(qkv), p(mask), p(ctx), p(dctx), p(lse), p(dqkv), p(part), B, L, heads, C.byref(dm), st)
            e1.record()
            torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) * 100)
        res[k] = (ctx, lse, dqkv, part, t)
    same = all(torch.equal(a, b) for a, b in zip(res["old"][:4], res["new"][:4]))
    ok &= same
    to, tn = res["old"][4], res["new"][4]
    print(f"B={B} L={L} masked={int(masked)} drop={drop}: {'identical' if same else 'DIFFERENT'}   fwd {to[0]:.1f} -> {tn[0]:.1f} us   bwd {to[1]:.1f} -> {tn[1]:.1f} us", flush=True)
sys.exit(0 if ok else 1)
