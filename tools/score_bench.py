#!/usr/bin/env python3
"""Metric 2 (BASELINE.json): eval query x passage dot-products/sec = Nq*Np / wall time of (score + top-k), embeddings
resident in HBM.  One shard of config 5 by default (cocodr-large: H=1024, 125k passages per GPU, 10k queries, k=1000).
Usage (GPU box): python tools/score_bench.py [--nq 10000 --np 125000 --dim 1024 --k 1000]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402
from cocodr_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--np", type=int, default=125000)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--exact", action="store_true", help="exact fp32 MFMA scores instead of the split-precision pipeline")
    a = ap.parse_args()
    ops.score_set_mode(1 if a.exact else 0)
    g = torch.Generator().manual_seed(7)
    Q = (torch.randn(a.nq, a.dim, generator=g) / a.dim ** 0.5).cuda()
    P = (torch.randn(a.np, a.dim, generator=g) / a.dim ** 0.5).cuda()
    need = ops.lib().cocodr_score_topk_workspace_bytes_dim(a.nq, a.np, a.dim, a.k)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    ops.score_topk(Q, P, a.k, workspace=ws)
    torch.cuda.synchronize()
    ops.prof_begin(3)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        D, I = ops.score_topk(Q, P, a.k, workspace=ws)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    n, ms, fl = ops.prof_end()
    print(json.dumps({"metric": "eval query x passage dot-products/sec", "value": a.nq * a.np / dt, "ms": dt * 1e3,
                      "nq": a.nq, "np": a.np, "dim": a.dim, "k": a.k,
                      "score_gemm_tflops_fp32": fl / (ms * 1e-3) / 1e12 if ms else None,
                      "score_gemm_share": ms / a.iters / (dt * 1e3) if ms else None}))


if __name__ == "__main__":
    main()
