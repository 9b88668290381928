"""Filtered against exhaustive search over a grid of sizes (which route should the plan pick?).  python tools/search_crossover.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cocodr_amd  # noqa: F401
from cocodr_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)


def t_ms(fn, it=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


for H in (768,):
    for npass in (33000, 50000, 125000, 500000):
        P = (torch.randn(npass, H, generator=g) / H ** 0.5).to(dev)
        for nq in (16, 256, 2000, 10000):
            Q = (torch.randn(nq, H, generator=g) / H ** 0.5).to(dev)
            for k in (10, 100, 1000):
                os.environ.pop("COCODR_SCORE_NOFILTER", None)
                os.environ["COCODR_SCORE_FILTER_FORCE"] = "1"
                plan = ops.score_filter_plan(nq, npass, H, k)
                if not plan["filtered"]:
                    continue
                ws = torch.empty(ops.lib().cocodr_score_topk_workspace_bytes_dim(nq, npass, H, k), dtype=torch.uint8, device=dev)
                tf = t_ms(lambda: ops.score_topk(Q, P, k, workspace=ws))
                os.environ["COCODR_SCORE_NOFILTER"] = "1"
                tx = t_ms(lambda: ops.score_topk(Q, P, k, workspace=ws))
                print(f"H {H} Np {npass:7d} Nq {nq:6d} k {k:5d}: filtered {tf:8.3f} ms  exhaustive {tx:8.3f} ms  ratio {tx / tf:5.2f}", flush=True)
                del ws
        del P
