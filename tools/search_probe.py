"""Where the brute-force search's time goes (one GPU's shard of config 5: 10 000 x 125 000 x 1024, k = 1000).
python tools/search_probe.py   - the score GEMM's shape timed alone with fp32 / bf16 output and as one launch over all queries,
then the whole search (piped / COCODR_SCORE_SERIAL=1 in a second process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cocodr_amd  # noqa: F401
from cocodr_amd import ops

dev = torch.device("cuda:0")


def t_us(fn, it=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it


def main():
    nq, npass, dim, k = 10000, 125000, 1024, 1000
    np_pad = (npass + 255) // 256 * 256
    g = torch.Generator().manual_seed(7)
    if "--gemm" in sys.argv or len(sys.argv) == 1:
        Bm = (torch.randn(np_pad, 3 * dim, generator=g) / dim ** 0.5).to(dev).to(torch.bfloat16)
        for M in (2048, 10000):
            A = (torch.randn(M, 3 * dim, generator=g) / dim ** 0.5).to(dev).to(torch.bfloat16)
            fl = 2.0 * M * np_pad * 3 * dim
            for f32 in (True, False):
                out = torch.empty(M, np_pad, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
                us = t_us(lambda: ops.gemm(A, Bm, out_f32=f32, out=out))
                print(f"gemm {M} x {np_pad} x {3 * dim}  out {'f32' if f32 else 'bf16'}: {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
                del out
            del A
        del Bm
    Q = (torch.randn(nq, dim, generator=g) / dim ** 0.5).to(dev)
    P = (torch.randn(npass, dim, generator=g) / dim ** 0.5).to(dev)
    ws = torch.empty(ops.lib().cocodr_score_topk_workspace_bytes_dim(nq, npass, dim, k), dtype=torch.uint8, device=dev)
    us = t_us(lambda: ops.score_topk(Q, P, k, workspace=ws))
    print(f"search {nq} x {npass} k={k} ({'serial' if os.environ.get('COCODR_SCORE_SERIAL') else 'piped'}): {us / 1e3:.2f} ms  "
          f"{nq * npass / us / 1e3:.1f} G dot products/s", flush=True)


main()
