"""A A^T of iDRO-sized gradient matrices: the native streaming gram (ops.gram) against torch (rocBLAS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import cocodr_amd
from cocodr_amd import ops
for G, D in ((4, 21_257_216), (16, 21_257_216), (50, 21_257_216), (50, 37_800_000), (64, 37_800_000)):
    a = torch.randn(G, D, device="cuda")
    for f, name in ((ops.gram, "native"), (lambda x: x @ x.T, "torch")):
        f(a); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5): f(a)
        torch.cuda.synchronize()
        print(G, D, name, round((time.perf_counter() - t) / 5 * 1e3, 3), "ms")
    del a
