"""The forward / dgrad GEMM forms of one BERT layer at PACKED row counts (the small-M regime of VERDICT r04 item 3): the shipped
selection against the vendor library (torch.matmul -> hipBLASLt, plain product: no bias / GELU / residual epilogue, so the library
number is a floor for what it would need with them) - a yardstick only, nothing in the product calls the library.
us per launch, best of rounds x 5 back-to-back launches; TFLOP/s in brackets.
    python tools/gemm_small_m_vs_library.py [--hidden 768] [--rows 4416,4776,5024]"""
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
    ap.add_argument("--rows", default="4416,4776,5024,5664")
    args = ap.parse_args()
    H, I = args.hidden, 4 * args.hidden
    g = torch.Generator().manual_seed(0)
    print(f"{'form':30s} {'rows':>6s} {'shipped (with epilogue)':>26s} {'shipped (plain)':>22s} {'library (plain)':>22s}")
    for T in [int(x) for x in args.rows.split(",")]:
        forms = [("fwd qkv", 3 * H, H, False, N.EPI_NONE), ("fwd out +res", H, H, False, N.EPI_ADD), ("fwd ffn1 gelu", I, H, False, N.EPI_GELU),
                 ("fwd ffn2 +res", H, I, False, N.EPI_ADD), ("dgrad ffn2 xgelu'", I, H, True, N.EPI_DGELU), ("dgrad ffn1 +res", H, I, True, N.EPI_ADD),
                 ("dgrad out", H, H, True, N.EPI_NONE), ("dgrad qkv +res", H, 3 * H, True, N.EPI_ADD)]
        tot = [0.0, 0.0, 0.0]
        for name, Nn, K, nn, epi in forms:
            a = (torch.randn(T, K, generator=g)).to(torch.bfloat16).cuda()
            w = (torch.randn((K, Nn) if nn else (Nn, K), generator=g) * 0.03).to(torch.bfloat16).cuda()
            r = torch.randn(T, Nn, generator=g).to(torch.bfloat16).cuda() if epi in (N.EPI_ADD, N.EPI_DGELU) else None
            bias = None if nn else torch.zeros(Nn, device="cuda")
            out = torch.empty(T, Nn, dtype=torch.bfloat16, device="cuda")
            t_epi = time_us(lambda: ops.gemm(a, w, trans_b=nn, bias=bias, epi=epi, r=r))
            t_plain = time_us(lambda: ops.gemm(a, w, trans_b=nn))
            t_lib = time_us(lambda: torch.matmul(a, w if nn else w.t(), out=out))
            fl = 2.0 * T * Nn * K
            tot = [tot[0] + t_epi, tot[1] + t_plain, tot[2] + t_lib]
            print(f"{name + f' N={Nn} K={K}':30s} {T:6d} {t_epi:14.1f} ({fl / t_epi / 1e6:6.0f}) {t_plain:12.1f} ({fl / t_plain / 1e6:6.0f}) {t_lib:12.1f} ({fl / t_lib / 1e6:6.0f})",
                  flush=True)
        print(f"{'sum of the eight forms':30s} {T:6d} {tot[0]:14.1f} {'':8s} {tot[1]:12.1f} {'':8s} {tot[2]:12.1f}", flush=True)


if __name__ == "__main__":
    main()
