#!/usr/bin/env python3
"""Per-workgroup phase timeline of the attention backward kernel (COCODR_ABL_TIMELINE build of csrc/attention.hip).
  python tools/attn_timeline.py --build      # here
  python tools/attn_timeline.py              # on the GPU box
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_abl")
LIB = os.path.join(OUT, "libabl_attn_timeline.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--L", type=int, default=128)
    args = ap.parse_args()
    if args.build:
        os.makedirs(OUT, exist_ok=True)
        csrc = os.path.join(ROOT, "coco-dr_amd", "csrc")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCOCODR_ABL_TIMELINE",
                        os.path.join(csrc, "attention.hip"), os.path.join(csrc, "core.hip"), "-o", LIB], check=True)
        print("built", LIB)
        return
    import numpy as np
    import torch
    lib = C.CDLL(LIB)
    B, L, heads = args.B, args.L, 12
    H = heads * 64
    qkv = (torch.randn(B * L, 3 * H, device="cuda") * 0.5).to(torch.bfloat16)
    mask = torch.ones(B, L, dtype=torch.int32, device="cuda")
    ctx = torch.empty(B * L, H, dtype=torch.bfloat16, device="cuda")
    dctx = torch.randn(B * L, H, device="cuda").to(torch.bfloat16)
    lse = torch.empty(B, heads, L, dtype=torch.float32, device="cuda")
    dqkv = torch.empty_like(qkv)
    stamps = torch.zeros(B * heads, 8, dtype=torch.int64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.cocodr_debug_attn_timeline(p(stamps)) == 0
    assert lib.cocodr_attn_fwd(p(qkv), p(mask), p(ctx), p(lse), B, L, heads, st) == 0
    for _ in range(3):
        stamps.zero_()
        assert lib.cocodr_attn_bwd(p(qkv), p(mask), p(ctx), p(dctx), p(lse), p(dqkv), None, B, L, heads, st) == 0
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().astype(np.float64)
    t0 = s[:, 0].min()
    s = (s - t0) / 100.0
    names = ["start", "staged", "phaseA computed (wave 0)", "phaseA done", "phaseB computed (wave 0)", "end"]
    print(f"attn_bwd B={B} L={L}: {len(s)} workgroups, span {s[:, 5].max():.1f} us")
    print("   start times: " + " ".join(f"{x:.1f}" for x in np.percentile(s[:, 0], [0, 25, 50, 75, 100])))
    for i in range(1, 6):
        d = s[:, i] - s[:, i - 1]
        print(f"   {names[i - 1]:26s} -> {names[i]:26s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f} us")


if __name__ == "__main__":
    main()
