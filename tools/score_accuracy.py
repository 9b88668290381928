#!/usr/bin/env python3
"""Accuracy of the search's score pipelines against an fp64 reference: the default split-precision pipeline
(two IEEE halves per operand, three partial products on the 16-bit matrix pipe), the exact fp32-MFMA pipeline (fmaf chain), an
fp32 BLAS product, and - through the bf16 GEMM - the bfloat16 splits that were considered and not used.
Usage (GPU box): python tools/score_accuracy.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402
from cocodr_amd import ops  # noqa: E402


def bf16_split(x):
    h = x.to(torch.bfloat16)
    r = x - h.float()
    m = r.to(torch.bfloat16)
    l = (r - m.float()).to(torch.bfloat16)
    return h, m, l


def main():
    torch.manual_seed(0)
    H, nq, npass, k = 1024, 512, 4096, 64
    gens = {"gaussian / sqrt(H)": lambda n: torch.randn(n, H) / H ** 0.5,
            "LayerNorm-shaped x 0.2": lambda n: torch.nn.functional.layer_norm(torch.randn(n, H) * 3 + 0.5, (H,)) * 0.2,
            "components over 6 binades": lambda n: torch.randn(n, H) * torch.exp2(torch.randint(-5, 1, (1, H)).float()) * 0.05}
    print(f"{'data':28s} {'pipeline':44s} {'max |err| / max |S|':>20s} {'rms / max |S|':>16s}")
    for name, gen in gens.items():
        Q, P = gen(nq).cuda(), gen(npass).cuda()
        ref = Q.double() @ P.double().T
        scale = ref.abs().max()

        def report(tag, S=None, DI=None):
            if DI is not None:
                D, I = DI
                err = (D.double() - torch.gather(ref, 1, I)).abs()
            else:
                err = (S.double() - ref).abs()
            print(f"{name:28s} {tag:44s} {(err.max() / scale).item():20.3e} {(err.pow(2).mean().sqrt() / scale).item():16.3e}")

        ops.score_set_mode(0)
        report("split precision, 2 x fp16, 3 products (default)", DI=ops.score_topk(Q, P, k))
        ops.score_set_mode(1)
        report("exact fp32 MFMA = fmaf chain (mode 1)", DI=ops.score_topk(Q, P, k))
        ops.score_set_mode(0)
        report("fp32 BLAS product (torch.matmul)", S=Q @ P.T)
        qh, qm, ql = bf16_split(Q)
        ph, pm, pl = bf16_split(P)
        for order, tag in (([(qm, pm), (qh, pl), (ql, ph), (qh, pm), (qm, ph), (qh, ph)], "3 x bf16, 6 products, small terms first"),
                           ([(qh, ph), (qh, pm), (qm, ph), (qh, pl), (ql, ph), (qm, pm)], "3 x bf16, 6 products, large terms first"),
                           ([(qh, pm), (qm, ph), (qh, ph)], "2 x bf16, 3 products (TF32-like)")):
            Qc = torch.cat([a for a, _ in order], 1).contiguous()
            Pc = torch.cat([b for _, b in order], 1).contiguous()
            report(tag, S=ops.gemm(Qc, Pc, out_f32=True))


if __name__ == "__main__":
    main()
