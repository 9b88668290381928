"""tools/experiments/gemm_w4s.hip (192 x 256 tile, 4 waves of 96 x 128, 12 accumulators per wave) against the shipped ping-pong pipeline
(impl 13): identical results?  us per launch, TFLOP/s.  Build first:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Icoco-dr_amd/csrc tools/experiments/gemm_w4s.hip -o tools/experiments/_build/libw4s.so"""
import ctypes as C
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import _native as N  # noqa: E402
from cocodr_amd._native import lib, stream_ptr  # noqa: E402

w4 = C.CDLL(os.path.join(root, "tools", "experiments", "_build", os.environ.get("W4LIB", "libw4s.so")))
w4.w4s_gemm.restype = C.c_int
w4.w4s_gemm.argtypes = [C.POINTER(N.GemmArgs), C.c_void_p]
L = lib()
sp = stream_ptr()


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


shapes = [("XL fwd qkv", 25600, 3072, 1024, N.EPI_NONE), ("XL fwd out +res", 25600, 1024, 1024, N.EPI_ADD), ("XL fwd ffn1 gelu", 25600, 4096, 1024, N.EPI_GELU),
          ("XL fwd ffn2 +res", 25600, 1024, 4096, N.EPI_ADD), ("cube 8192", 8192, 8192, 8192, N.EPI_NONE), ("ragged 17896 qkv", 17896, 3072, 1024, N.EPI_NONE),
          ("base fwd qkv", 8192, 2304, 768, N.EPI_NONE), ("base fwd ffn1 gelu", 8192, 3072, 768, N.EPI_GELU), ("base fwd ffn2 +res", 8192, 768, 3072, N.EPI_ADD),
          ("packed base qkv", 5664, 2304, 768, N.EPI_NONE), ("packed base ffn1", 5664, 3072, 768, N.EPI_GELU), ("packed large ffn1", 5664, 4096, 1024, N.EPI_GELU),
          ("enc qkv 65536", 65536, 2304, 768, N.EPI_NONE), ("fp32 out", 4096, 1024, 1024, -1)]
g0 = torch.Generator().manual_seed(0)
for name, M, Nn, K, epi in shapes:
    f32 = epi == -1
    epi = N.EPI_NONE if f32 else epi
    a = torch.randn(M, K, generator=g0).to(torch.bfloat16).cuda()
    w = (torch.randn(Nn, K, generator=g0) * 0.03).to(torch.bfloat16).cuda()
    bias = torch.randn(Nn, generator=g0).cuda()
    r = torch.randn(M, Nn, generator=g0).to(torch.bfloat16).cuda() if epi == N.EPI_ADD else None
    outs = []
    ts = []
    for which in ("pp", "w4s"):
        out = torch.zeros(M, Nn, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
        c2 = torch.zeros(M, Nn, dtype=torch.bfloat16, device="cuda") if epi == N.EPI_GELU else None
        g = N.GemmArgs()
        g.A, g.B, g.C, g.bias = a.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr()
        if c2 is not None:
            g.C2 = c2.data_ptr()
        if r is not None:
            g.R, g.ldr = r.data_ptr(), Nn
        g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.batch, g.epi, g.out_f32 = M, Nn, K, K, K, Nn, 1, epi, int(f32)
        if which == "pp":
            L.cocodr_gemm_set_impl(13)
            fn = lambda: L.cocodr_gemm(C.byref(g), sp)  # noqa: E731
        else:
            fn = lambda: w4.w4s_gemm(C.byref(g), sp)  # noqa: E731
        assert fn() == 0
        torch.cuda.synchronize()
        ts.append(time_us(fn))
        outs.append((out.clone(), None if c2 is None else c2.clone()))
        L.cocodr_gemm_set_impl(0)
    same = torch.equal(outs[0][0], outs[1][0]) and (outs[0][1] is None or torch.equal(outs[0][1], outs[1][1]))
    d = float((outs[0][0].float() - outs[1][0].float()).abs().max())
    fl = 2.0 * M * Nn * K
    print(f"{name:22s} {M}x{Nn}x{K}: pp {ts[0]:7.1f} us ({fl / ts[0] / 1e6:5.0f} TF)   w4s {ts[1]:7.1f} us ({fl / ts[1] / 1e6:5.0f} TF)   {'identical' if same else f'DIFFERENT (max abs {d:.3g})'}", flush=True)
