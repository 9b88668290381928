"""The hand-scheduled one-wave-per-SIMD GEMM (csrc/gemm_a4.hip, impl 14) against the shipped ping-pong pipeline (impl 13) and the
vendor library (torch.matmul, plain product; a yardstick, nothing in the product calls it): identical results?  us per launch, TFLOP/s.
  python tools/a4_check.py [--quick]"""
import ctypes as C
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import _native as N  # noqa: E402
from cocodr_amd._native import lib, stream_ptr  # noqa: E402

L = lib()
sp = stream_ptr()
IMPLS = [int(x) for x in os.environ.get("A4_IMPLS", "13,14").split(",")]


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


shapes = [("tiny 256x256x256", 256, 256, 256, N.EPI_NONE), ("one tile K=1024", 256, 256, 1024, N.EPI_NONE), ("ragged rows", 1000, 512, 384, N.EPI_ADD),
          ("NS fwd qkv", 32768, 3072, 1024, N.EPI_NONE), ("NS fwd out +res", 32768, 1024, 1024, N.EPI_ADD), ("NS fwd ffn1 gelu", 32768, 4096, 1024, N.EPI_GELU),
          ("NS fwd ffn2 +res", 32768, 1024, 4096, N.EPI_ADD), ("XL fwd qkv", 25600, 3072, 1024, N.EPI_NONE), ("cube 8192", 8192, 8192, 8192, N.EPI_NONE),
          ("ragged 17896 qkv", 17896, 3072, 1024, N.EPI_NONE), ("base fwd qkv", 8192, 2304, 768, N.EPI_NONE), ("base fwd ffn1 gelu", 8192, 3072, 768, N.EPI_GELU),
          ("base fwd ffn2 +res", 8192, 768, 3072, N.EPI_ADD), ("packed base qkv", 5664, 2304, 768, N.EPI_NONE), ("packed large ffn1", 5664, 4096, 1024, N.EPI_GELU),
          ("enc qkv 65536", 65536, 2304, 768, N.EPI_NONE), ("fp32 out", 4096, 1024, 1024, -1)]
if "--quick" in sys.argv:
    shapes = shapes[:5]
if os.environ.get("A4_ONLY"):
    keep = os.environ["A4_ONLY"].split(",")
    shapes = [s_ for s_ in shapes if any(k in s_[0] for k in keep)]
NOLIB = bool(os.environ.get("A4_NOLIB"))
g0 = torch.Generator().manual_seed(0)
bad = 0
for name, M, Nn, K, epi in shapes:
    f32 = epi == -1
    epi = N.EPI_NONE if f32 else epi
    a = torch.randn(M, K, generator=g0).to(torch.bfloat16).cuda()
    w = (torch.randn(Nn, K, generator=g0) * 0.03).to(torch.bfloat16).cuda()
    bias = torch.randn(Nn, generator=g0).cuda()
    r = torch.randn(M, Nn, generator=g0).to(torch.bfloat16).cuda() if epi == N.EPI_ADD else None
    outs = []
    ts = []
    for impl in IMPLS:
        out = torch.zeros(M, Nn, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
        c2 = torch.zeros(M, Nn, dtype=torch.bfloat16, device="cuda") if epi == N.EPI_GELU else None
        g = N.GemmArgs()
        g.A, g.B, g.C, g.bias = a.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr()
        if c2 is not None:
            g.C2 = c2.data_ptr()
        if r is not None:
            g.R, g.ldr = r.data_ptr(), Nn
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.batch, g.epi, g.out_f32 = M, Nn, K, K, K, Nn, 1, epi, int(f32)
        L.cocodr_gemm_set_impl(impl)
        fn = lambda: L.cocodr_gemm(C.byref(g), sp)  # noqa: E731
        assert fn() == 0
        torch.cuda.synchronize()
        ts.append(time_us(fn))
        outs.append((out.clone(), None if c2 is None else c2.clone()))
        L.cocodr_gemm_set_impl(0)
    t_lib = 1.0 if NOLIB else time_us(lambda: torch.matmul(a, w.t()))
    same = all(torch.equal(outs[0][0], o[0]) and (outs[0][1] is None or torch.equal(outs[0][1], o[1])) for o in outs[1:])
    d = max(float((outs[0][0].float() - o[0].float()).abs().max()) for o in outs[1:]) if len(outs) > 1 else 0.0
    bad += 0 if same else 1
    fl = 2.0 * M * Nn * K
    cols = "   ".join(f"impl{i} {t:7.1f} us ({fl / t / 1e6:5.0f} TF)" for i, t in zip(IMPLS, ts))
    print(f"{name:20s} {M}x{Nn}x{K}: {cols}   library {t_lib:7.1f} us ({fl / t_lib / 1e6:5.0f} TF)   {'identical' if same else f'DIFFERENT (max abs {d:.3g})'}", flush=True)
print("check rc=%d" % bad)
sys.exit(1 if bad else 0)
