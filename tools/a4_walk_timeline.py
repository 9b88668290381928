"""Per-tile timeline of the persistent hand-scheduled GEMM (csrc/gemm_a4.hip built with -DCOCODR_A4_TIMELINE): 100 MHz wall-clock
stamps of wave 0 of the first 256 workgroups - before the loop statement, behind it, behind the epilogue - for up to 21 tiles each.
Build:  tools/a4_timeline_build.sh  (-> tools/experiments/_build/lib_a4_timeline.so)
  python tools/a4_walk_timeline.py M N K [epi] [nn]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import _native as N  # noqa: E402

N.LIB_PATH = os.path.join(root, "tools", "experiments", "_build", "lib_a4_timeline%s.so" % os.environ.get("A4_TL_TAG", ""))
L = N.lib()
sp = N.stream_ptr()
L.cocodr_a4_timeline_read.restype = C.c_int
L.cocodr_a4_timeline_read.argtypes = [C.c_void_p, C.c_int]

M, Nn, K = (int(x) for x in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else N.EPI_NONE
nn = len(sys.argv) > 5 and sys.argv[5] == "nn"
g0 = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g0).to(torch.bfloat16).cuda()
w = (torch.randn(*((K, Nn) if nn else (Nn, K)), generator=g0) * 0.03).to(torch.bfloat16).cuda()
bias = torch.randn(Nn, generator=g0).cuda()
r = torch.randn(M, Nn, generator=g0).to(torch.bfloat16).cuda()
out = torch.zeros(M, Nn, dtype=torch.bfloat16, device="cuda")
c2 = torch.zeros(M, Nn, dtype=torch.bfloat16, device="cuda")
g = N.GemmArgs()
g.A, g.B, g.C, g.bias, g.C2, g.R, g.ldr = a.data_ptr(), w.data_ptr(), out.data_ptr(), 0 if nn else bias.data_ptr(), c2.data_ptr(), r.data_ptr(), Nn
g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.batch, g.epi, g.out_f32 = M, Nn, K, K, Nn if nn else K, Nn, 1, epi, 0
g.trans_b = 1 if nn else 0
L.cocodr_gemm_set_impl(15)
for _ in range(5):
    assert L.cocodr_gemm(C.byref(g), sp) == 0
torch.cuda.synchronize()
tiles = ((M + 255) // 256) * (Nn // 256)
grid = min(tiles, 256)
rounds = min(21, (tiles + grid - 1) // grid)
buf = np.zeros(256 * 64, np.uint64)
assert L.cocodr_a4_timeline_read(buf.ctypes.data, 256 * 64) == 0
t = buf.reshape(256, 64)[:grid, :rounds * 3].astype(np.float64).reshape(grid, rounds, 3)
t0 = t[:, 0, 0].min()
t = (t - t0) / 100.0   # microseconds
print(f"{M}x{Nn}x{K} epi {epi} {'nn' if nn else 'nt'}: {tiles} tiles, {grid} workgroups x {rounds} rounds; microseconds")
print("  round:   loop entry (min..max)     loop        epilogue    gap to the next loop entry")
for q in range(rounds):
    loop = t[:, q, 1] - t[:, q, 0]
    epi_ = t[:, q, 2] - t[:, q, 1]
    gap = (t[:, q + 1, 0] - t[:, q, 2]) if q + 1 < rounds else np.zeros(grid)
    print(f"  {q:3d}   {t[:, q, 0].min():8.2f} .. {t[:, q, 0].max():8.2f}   {loop.mean():6.2f} ({loop.min():5.2f}..{loop.max():5.2f})   "
          f"{epi_.mean():6.2f} ({epi_.min():5.2f}..{epi_.max():5.2f})   {gap.mean():5.2f}")
print(f"  last epilogue ends at {t[:, rounds - 1, 2].max():.2f} (first entry at 0)")
