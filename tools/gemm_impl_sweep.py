"""Every pipeline geometry (cocodr_gemm_set_impl) on the forward / dgrad GEMM forms of one BERT layer at packed row counts: is the
shipped selection (impl 0) the fastest?  us per launch, best of `rounds` x 5 back-to-back launches.
    python tools/gemm_impl_sweep.py [--hidden 768] [--rows 5024,5664,6304] [--impls 0,2,4,5,9,12,13]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import ops  # noqa: E402
from cocodr_amd import _native as N  # noqa: E402


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
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--rows", default="5024,5664,6304,8192")
    ap.add_argument("--impls", default="0,2,3,4,5,9,12,13")
    args = ap.parse_args()
    H, I = args.hidden, 4 * args.hidden
    impls = [int(x) for x in args.impls.split(",")]
    g = torch.Generator().manual_seed(0)
    print(f"{'form':30s} {'rows':>6s} " + " ".join(f"impl{i:>2d}".rjust(8) for i in impls) + "   best   (us)")
    for T in [int(x) for x in args.rows.split(",")]:
        forms = [("fwd qkv", 3 * H, H, False, N.EPI_NONE), ("fwd out +res", H, H, False, N.EPI_ADD), ("fwd ffn1 gelu", I, H, False, N.EPI_GELU),
                 ("fwd ffn2 +res", H, I, False, N.EPI_ADD), ("dgrad ffn2 xgelu'", I, H, True, N.EPI_DGELU), ("dgrad ffn1 +res", H, I, True, N.EPI_ADD),
                 ("dgrad out", H, H, True, N.EPI_NONE), ("dgrad qkv +res", H, 3 * H, True, N.EPI_ADD)]
        tot = {i: 0.0 for i in impls}
        for name, Nn, K, nn, epi in forms:
            a = (torch.randn(T, K, generator=g)).to(torch.bfloat16).cuda()
            w = (torch.randn((K, Nn) if nn else (Nn, K), generator=g) * 0.03).to(torch.bfloat16).cuda()
            r = torch.randn(T, Nn, generator=g).to(torch.bfloat16).cuda() if epi in (N.EPI_ADD, N.EPI_DGELU) else None
            bias = None if nn else torch.zeros(Nn, device="cuda")
            res = {}
            for i in impls:
                try:
                    ops.gemm_set_impl(i)
                    res[i] = time_us(lambda: ops.gemm(a, w, trans_b=nn, bias=bias, epi=epi, r=r))
                except Exception:
                    res[i] = float("nan")
                tot[i] += res[i]
            ops.gemm_set_impl(0)
            best = min((v, k) for k, v in res.items() if v == v)
            print(f"{name + f' N={Nn} K={K}':30s} {T:6d} " + " ".join(f"{res[i]:8.1f}" for i in impls) + f"   impl{best[1]}", flush=True)
        print(f"{'sum of the eight forms':30s} {T:6d} " + " ".join(f"{tot[i]:8.1f}" for i in impls), flush=True)


if __name__ == "__main__":
    main()
