"""Row-split probe: a GEMM whose 256 x 256 tiles need a little more than k whole rounds of the 256 CUs, run as (a) the leading rows
that fill whole rounds on the 256 x 256-tile pipeline (impl 13) + (b) the remaining rows on a small-tile geometry, two launches
on one stream - against the single-launch selections.  us per call (pair), best of 3 x 10.
    python tools/gemm_rowsplit_probe.py [--hidden 768] [--rows 5024,5664,6304]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import _native as N  # noqa: E402
from cocodr_amd._native import lib, stream_ptr  # noqa: E402


def args_for(a, w, out, r, bias, nn, epi, row0, rows, c2=None):
    K = a.shape[1]
    Nn = w.shape[1] if nn else w.shape[0]
    g = N.GemmArgs()
    g.A = a.data_ptr() + row0 * K * 2
    g.B = w.data_ptr()
    g.C = out.data_ptr() + row0 * Nn * 2
    if c2 is not None:
        g.C2 = c2.data_ptr() + row0 * Nn * 2
    if r is not None:
        g.R, g.ldr = r.data_ptr() + row0 * Nn * 2, Nn
    if bias is not None:
        g.bias = bias.data_ptr()
    g.M, g.N, g.K = rows, Nn, K
    g.lda, g.ldb, g.ldc = K, w.shape[1], Nn
    g.trans_a, g.trans_b, g.epi, g.batch = 0, int(nn), epi, 1
    return g


def time_us(fn, rounds=3, n=10):
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
    ap.add_argument("--rows", default="5024,5664,6304")
    ap.add_argument("--small", default="2,3,4,5,9,12")
    a_ = ap.parse_args()
    H, I = a_.hidden, 4 * a_.hidden
    small = [int(x) for x in a_.small.split(",")]
    L = lib()
    sp = stream_ptr()
    g0 = torch.Generator().manual_seed(0)

    def run(g, impl):
        L.cocodr_gemm_set_impl(impl)
        rc = L.cocodr_gemm(C.byref(g), sp)
        assert rc == 0, rc

    forms = [("fwd qkv", 3 * H, H, False, N.EPI_NONE), ("fwd out +res", H, H, False, N.EPI_ADD), ("fwd ffn1 gelu", I, H, False, N.EPI_GELU),
             ("fwd ffn2 +res", H, I, False, N.EPI_ADD), ("dgrad ffn2 xgelu'", I, H, True, N.EPI_DGELU), ("dgrad ffn1 +res", H, I, True, N.EPI_ADD),
             ("dgrad out", H, H, True, N.EPI_NONE), ("dgrad qkv +res", H, 3 * H, True, N.EPI_ADD)]
    print(f"{'form':30s} {'rows':>6s} {'impl0':>7s} {'impl13':>7s} {'R':>6s} {'head13':>7s} " + " ".join(f"+tail{i}".rjust(8) for i in small) + "   best pair")
    for T in [int(x) for x in a_.rows.split(",")]:
        tot0 = totb = 0.0
        for name, Nn, K, nn, epi in forms:
            a = torch.randn(T, K, generator=g0).to(torch.bfloat16).cuda()
            w = (torch.randn((K, Nn) if nn else (Nn, K), generator=g0) * 0.03).to(torch.bfloat16).cuda()
            r = torch.randn(T, Nn, generator=g0).to(torch.bfloat16).cuda() if epi in (N.EPI_ADD, N.EPI_DGELU) else None
            bias = None if nn else torch.zeros(Nn, device="cuda")
            out = torch.empty(T, Nn, dtype=torch.bfloat16, device="cuda")
            c2 = torch.empty_like(out) if epi == N.EPI_GELU else None
            full = args_for(a, w, out, r, bias, nn, epi, 0, T, c2)
            t0 = time_us(lambda: run(full, 0))
            t13 = time_us(lambda: run(full, 13)) if Nn % 256 == 0 else float("nan")
            ncol = Nn // 256
            tiles = (T + 255) // 256 * ncol
            rounds = tiles // 256
            rem = tiles - rounds * 256
            line = f"{name + f' N={Nn} K={K}':30s} {T:6d} {t0:7.1f} {t13:7.1f}"
            best = t0
            if rounds >= 1 and 0 < rem:
                R = (rounds * 256 // ncol) * 256
                head = args_for(a, w, out, r, bias, nn, epi, 0, R, c2)
                tail = args_for(a, w, out, r, bias, nn, epi, R, T - R, c2)
                th = time_us(lambda: run(head, 13))
                line += f" {R:6d} {th:7.1f} "
                res = {}
                for i in small:
                    def pair():
                        run(head, 13)
                        run(tail, i)
                    try:
                        res[i] = time_us(pair)
                    except AssertionError:
                        res[i] = float("nan")
                line += " ".join(f"{res[i]:8.1f}" for i in small)
                bp = min((v, k) for k, v in res.items() if v == v)
                line += f"   {bp[0]:.1f} (tail impl {bp[1]})"
                best = min(best, bp[0])
            L.cocodr_gemm_set_impl(0)
            tot0 += t0
            totb += best
            print(line, flush=True)
        print(f"{'sum: shipped / with row split':30s} {T:6d} {tot0:7.1f} {totb:7.1f}", flush=True)


if __name__ == "__main__":
    main()
