#!/usr/bin/env python3
"""Where does the automatic geometry choice of cocodr_gemm lose to a fixed one?  Sweeps the encoder's GEMM shapes for
BERT-base / large at several token counts.  Usage (GPU box): python tools/gemm_autotune_check.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa
from cocodr_amd import ops

IMPLS = [0, 2, 3, 4, 5, 9]


def time_gemm(a, b, ta, tb, f32, out, impl):
    ops.gemm_set_impl(impl)
    best = 1e9
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm(a, b, trans_a=ta, trans_b=tb, out_f32=f32, out=out)
        e1.record(); torch.cuda.synchronize()
        if r:
            best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
    return best


def main():
    shapes = []
    for model, H, I, NL in (("base", 768, 3072, 12), ("large", 1024, 4096, 24)):
        for M in (2048, 8192, 32768):
            for name, N, K in (("qkv", 3 * H, H), ("out", H, H), ("ffn1", I, H), ("ffn2", H, I)):
                shapes.append((f"{model} fwd {name} M{M}", M, N, K, 0, 0, 1))
                shapes.append((f"{model} dgrad {name} M{M}", M, K, N, 0, 1, 1))
            if M <= 8192:
                for name, N, K in (("qkv", 3 * H, H), ("out", H, H), ("ffn1", I, H), ("ffn2", H, I)):
                    shapes.append((f"{model} wgrad {name} M{M} x{NL}", N, K, M, 1, 1, NL))
    bad = 0
    for name, M, N, K, ta, tb, nb in shapes:
        ash = (nb, K, M) if ta else (nb, M, K)
        bsh = (nb, K, N) if tb else (nb, N, K)
        a = torch.randn(ash, device="cuda").to(torch.bfloat16)
        b = (torch.randn(bsh, device="cuda") * 0.05).to(torch.bfloat16)
        if nb == 1:
            a, b = a[0], b[0]
        out = torch.empty((nb, M, N) if nb > 1 else (M, N), dtype=torch.float32 if ta else torch.bfloat16, device="cuda")
        t = {i: 1e9 for i in IMPLS}
        for _rep in range(2):  # interleaved repeats: clocks drift between launches
            for i in IMPLS:
                t[i] = min(t[i], time_gemm(a, b, bool(ta), bool(tb), bool(ta), out, i))
        best = min(IMPLS[1:], key=lambda i: t[i])
        flag = "  <-- auto loses %.0f %%" % (100 * (t[0] / t[best] - 1)) if t[0] > 1.04 * t[best] else ""
        bad += bool(flag)
        print(f"{name:34s} auto {t[0]:8.1f} us   best impl {best} {t[best]:8.1f} us{flag}", flush=True)
        del a, b, out
    ops.gemm_set_impl(0)
    print(f"{bad} of {len(shapes)} shapes where auto is > 4 % off the best fixed geometry")


if __name__ == "__main__":
    main()
