"""Packed row counts (any multiple of 32) through the forward / dgrad GEMM forms of one BERT layer: the shipped selection without
a split workspace (whole tiles only: round-3 behaviour), the 256 x 256-tile pipeline with its last partial round cut into
contraction slices, and the selection with the workspace.  us per launch, best of `rounds` x 5 back-to-back launches.
    python tools/gemm_tail_sweep.py [--hidden 1024] [--rows 4416,7520,...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import ops  # noqa: E402


def time_us(fn, rounds=3, n=5):
    best = 1e9
    for _ in range(rounds):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--rows", default="4416,5664,7520,9984,12512,15008,17888,20480,23360,26016")
    args = ap.parse_args()
    H, I = args.hidden, 4 * args.hidden
    ws = torch.empty(ops.lib().cocodr_gemm_split_workspace_floats(), dtype=torch.float32, device="cuda")
    g = torch.Generator().manual_seed(0)
    print(f"{'form':26s} {'rows':>6s} {'tiles':>6s} {'auto/no-ws':>11s} {'pp whole':>9s} {'pp cut':>8s} {'auto+ws':>8s}   (us)")
    for T in [int(x) for x in args.rows.split(",")]:
        forms = [("fwd qkv", 3 * H, H, False), ("fwd out", H, H, False), ("fwd ffn1", I, H, False), ("fwd ffn2", H, I, False),
                 ("dgrad ffn2", I, H, True), ("dgrad ffn1", H, I, True), ("dgrad qkv", H, 3 * H, True)]
        for name, Nn, K, nn in forms:
            if Nn % 256:
                continue
            a = (torch.randn(T, K, generator=g)).to(torch.bfloat16).cuda()
            w = (torch.randn((K, Nn) if nn else (Nn, K), generator=g) * 0.03).to(torch.bfloat16).cuda()
            out = torch.empty((T, Nn), dtype=torch.bfloat16, device="cuda")
            call = lambda **kw: ops.gemm(a, w, trans_b=nn, out=out, **kw)  # noqa: E731
            tiles = (T + 255) // 256 * (Nn // 256)
            ops.gemm_set_impl(0)
            t_auto = time_us(lambda: call())
            t_ws = time_us(lambda: call(split_ws=ws))
            ops.gemm_set_impl(13)
            t_whole = time_us(lambda: call())
            t_cut = time_us(lambda: call(split_ws=ws))
            ops.gemm_set_impl(0)
            print(f"{name + f' N={Nn} K={K}':26s} {T:6d} {tiles:6d} {t_auto:11.1f} {t_whole:9.1f} {t_cut:8.1f} {t_ws:8.1f}", flush=True)


if __name__ == "__main__":
    main()
