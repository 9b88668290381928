"""Per-workgroup timeline of the one-tile-per-workgroup hand-scheduled GEMM (csrc/gemm_a4.hip built with -DCOCODR_A4_TIMELINE):
shader-clock stamps of wave 0: kernel entry, loop statement entry, loop statement exit, end of epilogue pass 0 / 1.
Build:  hipcc ... -DCOCODR_A4_TIMELINE -c coco-dr_amd/csrc/gemm_a4.hip -o /tmp/a4t.o && link with coco-dr_amd/build/*.o into tools/experiments/_build/lib_a4_timeline.so
  python tools/a4_timeline.py M N K [epi]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import _native as N  # noqa: E402

N.LIB_PATH = os.path.join(root, "tools", "experiments", "_build", "lib_a4_timeline.so")
L = N.lib()
sp = N.stream_ptr()
L.cocodr_a4_timeline_read.restype = C.c_int
L.cocodr_a4_timeline_read.argtypes = [C.c_void_p, C.c_int]

M, Nn, K = (int(x) for x in sys.argv[1:4])
epi = int(sys.argv[4]) if len(sys.argv) > 4 else N.EPI_NONE
g0 = torch.Generator().manual_seed(0)
a = torch.randn(M, K, generator=g0).to(torch.bfloat16).cuda()
w = (torch.randn(Nn, K, generator=g0) * 0.03).to(torch.bfloat16).cuda()
bias = torch.randn(Nn, generator=g0).cuda()
r = torch.randn(M, Nn, generator=g0).to(torch.bfloat16).cuda()
out = torch.zeros(M, Nn, dtype=torch.bfloat16, device="cuda")
c2 = torch.zeros(M, Nn, dtype=torch.bfloat16, device="cuda")
g = N.GemmArgs()
g.A, g.B, g.C, g.bias, g.C2, g.R, g.ldr = a.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr(), c2.data_ptr(), r.data_ptr(), Nn
g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.batch, g.epi, g.out_f32 = M, Nn, K, K, K, Nn, 1, epi, 0
L.cocodr_gemm_set_impl(14)
for _ in range(3):
    assert L.cocodr_gemm(C.byref(g), sp) == 0
torch.cuda.synchronize()
tiles = ((M + 255) // 256) * (Nn // 256)
n = min(tiles, 2048)
buf = np.zeros(n * 8, np.uint64)
assert L.cocodr_a4_timeline_read(buf.ctypes.data, n * 8) == 0
t = buf.reshape(n, 8)[:, :5].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0  # s_memtime = shader clocks: units of 100 clocks
order = np.argsort(t[:, 0])
t = t[order]
print(f"{M}x{Nn}x{K} epi {epi}: {tiles} tiles; durations in units of 100 shader clocks (s_memtime; the stamps of different XCDs are not synchronised)")
print("  columns: entry, loop entry, loop exit, epilogue pass 0 done, epilogue pass 1 done")
for q in (0, 1, 127, 255, 256, 257, 511, 512, 767, 1023, n - 1):
    if q < n:
        print(f"  wg #{q:5d} (by entry time): " + "  ".join(f"{x:8.2f}" for x in t[q]))
d = np.diff(t, axis=1)
print("  mean durations: setup %.2f  loop %.2f  epilogue pass 0 %.2f  pass 1 %.2f   (loop: min %.2f max %.2f)" %
      (d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean(), d[:, 1].min(), d[:, 1].max()))
first = t[:256] if n >= 256 else t
print("  first round: loop %.2f, epilogue %.2f + %.2f" % (np.diff(first, axis=1)[:, 1].mean(), np.diff(first, axis=1)[:, 2].mean(), np.diff(first, axis=1)[:, 3].mean()))
if n > 256:
    later = t[256:]
    dl = np.diff(later, axis=1)
    print("  later rounds: loop %.2f, epilogue %.2f + %.2f" % (dl[:, 1].mean(), dl[:, 2].mean(), dl[:, 3].mean()))
    # gap between a workgroup's end and the next entry on the chip: k-th exit vs (k+256)-th entry
    ends = np.sort(t[:, 4])
    starts = np.sort(t[:, 0])
    k = min(len(starts) - 256, len(ends))
    print("  entry of workgroup k + 256 minus exit of the k-th finished: mean %.2f us" % float((starts[256:256 + k] - ends[:k]).mean()))
print("  kernel span (first entry -> last exit): %.2f us" % float(t[:, 4].max()))
