"""Fixed cost of a one-round launch of the 256 x 256-tile pipeline: time against the contraction length at a fixed output
(5664 x 2304: 207 tiles on 256 CUs), NT form + bias.  T(K) = F + c K: F = launch + prologue + epilogue, c = the loop's rate."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import _native as N  # noqa: E402
from cocodr_amd._native import lib, stream_ptr  # noqa: E402

L = lib()
sp = stream_ptr()


def time_us(fn, rounds=5, n=20):
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


for impl in (13, 18, 5, 9, 12):
    for M, Nn in ((5664, 2304), (5664, 768), (8192, 3072)):
        row = []
        for K in (64, 128, 256, 512, 768, 1536, 3072):
            a = torch.randn(M, K).to(torch.bfloat16).cuda()
            w = (torch.randn(Nn, K) * 0.03).to(torch.bfloat16).cuda()
            out = torch.empty(M, Nn, dtype=torch.bfloat16, device="cuda")
            bias = torch.zeros(Nn, device="cuda")
            g = N.GemmArgs()
            g.A, g.B, g.C, g.bias = a.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr()
            g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.batch = M, Nn, K, K, K, Nn, 1
            L.cocodr_gemm_set_impl(impl)
            try:
                row.append(time_us(lambda: L.cocodr_gemm(C.byref(g), sp)))
            except Exception:
                row.append(float("nan"))
        L.cocodr_gemm_set_impl(0)
        c = (row[-1] - row[4]) / (3072 - 768) * 64
        print(f"impl {impl:2d} {M} x {Nn}: " + " ".join(f"K={k}: {t:5.1f}" for k, t in zip((64, 128, 256, 512, 768, 1536, 3072), row)) + f"   us; slope {c:.2f} us per 64-wide K step, intercept {row[4] - c * 12:.1f} us", flush=True)
