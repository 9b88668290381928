#!/usr/bin/env python3
"""A/B micro-benchmark of the GEMM implementations on the encoder's shapes (interleaved rounds, one process).
Usage (GPU box): python tools/gemm_bench.py [--rounds 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402
from cocodr_amd import ops  # noqa: E402

SHAPES = [  # (name, M, N, K, ta, tb, batch, out_f32)
    ("fwd qkv   8192x2304x768", 8192, 2304, 768, 0, 0, 1, 0),
    ("fwd out   8192x768x768", 8192, 768, 768, 0, 0, 1, 0),
    ("fwd ffn1  8192x3072x768", 8192, 3072, 768, 0, 0, 1, 0),
    ("fwd ffn2  8192x768x3072", 8192, 768, 3072, 0, 0, 1, 0),
    ("dgrad ffn2 8192x3072x768", 8192, 3072, 768, 0, 1, 1, 0),
    ("dgrad ffn1 8192x768x3072", 8192, 768, 3072, 0, 1, 1, 0),
    ("dgrad qkv 8192x768x2304", 8192, 768, 2304, 0, 1, 1, 0),
    ("wgrad qkv 12x 2304x768x8192", 2304, 768, 8192, 1, 1, 12, 1),
    ("wgrad out 12x 768x768x8192", 768, 768, 8192, 1, 1, 12, 1),
    ("wgrad ffn1 12x 3072x768x8192", 3072, 768, 8192, 1, 1, 12, 1),
    ("wgrad ffn2 12x 768x3072x8192", 768, 3072, 8192, 1, 1, 12, 1),
    ("large fwd ffn1 8192x4096x1024", 8192, 4096, 1024, 0, 0, 1, 0),
    ("large fwd qkv 2048x3072x1024", 2048, 3072, 1024, 0, 0, 1, 0),
    # BERT-large (H 1024, I 4096) at 8192 tokens
    ("L fwd qkv  8192x3072x1024", 8192, 3072, 1024, 0, 0, 1, 0),
    ("L fwd out  8192x1024x1024", 8192, 1024, 1024, 0, 0, 1, 0),
    ("L fwd ffn1 8192x4096x1024", 8192, 4096, 1024, 0, 0, 1, 0),
    ("L fwd ffn2 8192x1024x4096", 8192, 1024, 4096, 0, 0, 1, 0),
    ("L dgrad ffn2 8192x4096x1024", 8192, 4096, 1024, 0, 1, 1, 0),
    ("L dgrad ffn1 8192x1024x4096", 8192, 1024, 4096, 0, 1, 1, 0),
    ("L dgrad qkv 8192x1024x3072", 8192, 1024, 3072, 0, 1, 1, 0),
    ("L wgrad qkv 24x 3072x1024x8192", 3072, 1024, 8192, 1, 1, 24, 1),
    ("L wgrad out 24x 1024x1024x8192", 1024, 1024, 8192, 1, 1, 24, 1),
    ("L wgrad ffn1 24x 4096x1024x8192", 4096, 1024, 8192, 1, 1, 24, 1),
    # BERT-large at 25600 tokens (COCO/README.md:59-63: 100 documents = 200 spans per GPU)
    ("XL fwd qkv  25600x3072x1024", 25600, 3072, 1024, 0, 0, 1, 0),
    ("XL fwd out  25600x1024x1024", 25600, 1024, 1024, 0, 0, 1, 0),
    ("XL fwd ffn1 25600x4096x1024", 25600, 4096, 1024, 0, 0, 1, 0),
    ("XL fwd ffn2 25600x1024x4096", 25600, 1024, 4096, 0, 0, 1, 0),
    ("XL dgrad ffn2 25600x4096x1024", 25600, 4096, 1024, 0, 1, 1, 0),
    ("XL dgrad ffn1 25600x1024x4096", 25600, 1024, 4096, 0, 1, 1, 0),
    ("XL dgrad qkv 25600x1024x3072", 25600, 1024, 3072, 0, 1, 1, 0),
    ("XL wgrad qkv 24x 3072x1024x25600", 3072, 1024, 25600, 1, 1, 24, 1),
    ("XL wgrad ffn1 24x 4096x1024x25600", 4096, 1024, 25600, 1, 1, 24, 1),
    # packed batches (SURVEY 7 iii): row counts that are no multiple of the tile height (64 / 512 MS MARCO-shaped sequences)
    ("pk fwd qkv   5664x2304x768", 5664, 2304, 768, 0, 0, 1, 0),
    ("pk fwd out   5664x768x768", 5664, 768, 768, 0, 0, 1, 0),
    ("pk fwd ffn1  5664x3072x768", 5664, 3072, 768, 0, 0, 1, 0),
    ("pk fwd ffn2  5664x768x3072", 5664, 768, 3072, 0, 0, 1, 0),
    ("pk dgrad ffn1 5664x768x3072", 5664, 768, 3072, 0, 1, 1, 0),
    ("pk wgrad ffn1 12x 3072x768x5664", 3072, 768, 5664, 1, 1, 12, 1),
    ("pk enc qkv  45312x2304x768", 45312, 2304, 768, 0, 0, 1, 0),
    ("pk enc out  45312x768x768", 45312, 768, 768, 0, 0, 1, 0),
    ("pk enc ffn1 45312x3072x768", 45312, 3072, 768, 0, 0, 1, 0),
    ("pk enc ffn2 45312x768x3072", 45312, 768, 3072, 0, 0, 1, 0),
    ("cube 4096", 4096, 4096, 4096, 0, 0, 1, 0),
    ("cube 8192", 8192, 8192, 8192, 0, 0, 1, 0),
    # corpus-encode batch (512 x 128 tokens, forward only)
    ("enc qkv  65536x2304x768", 65536, 2304, 768, 0, 0, 1, 0),
    ("enc out  65536x768x768", 65536, 768, 768, 0, 0, 1, 0),
    ("enc ffn1 65536x3072x768", 65536, 3072, 768, 0, 0, 1, 0),
    ("enc ffn2 65536x768x3072", 65536, 768, 3072, 0, 0, 1, 0),
    # K sweep at a fixed 8192x3072 output (3 full rounds of 128x128 tiles at 2 WG/CU): time = fixed + per-K-step
    ("ksweep 8192x3072x64", 8192, 3072, 64, 0, 0, 1, 0),
    ("ksweep 8192x3072x128", 8192, 3072, 128, 0, 0, 1, 0),
    ("ksweep 8192x3072x256", 8192, 3072, 256, 0, 0, 1, 0),
    ("ksweep 8192x3072x1536", 8192, 3072, 1536, 0, 0, 1, 0),
    ("ksweep 8192x3072x3072", 8192, 3072, 3072, 0, 0, 1, 0),
    ("ksweep 8192x3072x6144", 8192, 3072, 6144, 0, 0, 1, 0),
    ("pkXL fwd qkv  17896x3072x1024", 17896, 3072, 1024, 0, 0, 1, 0),   # 54-60: the 200-sequence BERT-large batch, packed
    ("pkXL fwd out  17896x1024x1024", 17896, 1024, 1024, 0, 0, 1, 0),
    ("pkXL fwd ffn1 17896x4096x1024", 17896, 4096, 1024, 0, 0, 1, 0),
    ("pkXL fwd ffn2 17896x1024x4096", 17896, 1024, 4096, 0, 0, 1, 0),
    ("pkXL dgrad ffn2 17896x4096x1024", 17896, 4096, 1024, 0, 1, 1, 0),
    ("pkXL dgrad ffn1 17896x1024x4096", 17896, 1024, 4096, 0, 1, 1, 0),
    ("pkXL dgrad qkv 17896x1024x3072", 17896, 1024, 3072, 0, 1, 1, 0),
    # 61: one query chunk of the config-5 search: 2048 queries x 125 184 passages, depth 3 x 1024 (split precision), fp32 slab out
    ("score chunk 2048x125184x3072 f32", 2048, 125184, 3072, 0, 0, 1, 1),
]


EPI_SHAPES = [  # the encoder's fused epilogues: (name, M, N, K, tb, epi, bias, residual)
    ("fwd qkv  +bias        8192x2304x768", 8192, 2304, 768, 0, "none", True, False),
    ("fwd out  +bias+res    8192x768x768", 8192, 768, 768, 0, "add", True, True),
    ("fwd ffn1 +bias+gelu   8192x3072x768", 8192, 3072, 768, 0, "gelu", True, False),
    ("fwd ffn2 +bias+res    8192x768x3072", 8192, 768, 3072, 0, "add", True, True),
    ("dgrad ffn2 *gelu'     8192x3072x768", 8192, 3072, 768, 1, "dgelu", False, True),
    ("dgrad ffn1 +res       8192x768x3072", 8192, 768, 3072, 1, "add", False, True),
    ("dgrad qkv +res        8192x768x2304", 8192, 768, 2304, 1, "add", False, True),
    ("L fwd qkv  +bias      8192x3072x1024", 8192, 3072, 1024, 0, "none", True, False),
    ("L fwd out  +bias+res  8192x1024x1024", 8192, 1024, 1024, 0, "add", True, True),
    ("L fwd ffn1 +bias+gelu 8192x4096x1024", 8192, 4096, 1024, 0, "gelu", True, False),
    ("L fwd ffn2 +bias+res  8192x1024x4096", 8192, 1024, 4096, 0, "add", True, True),
    ("L dgrad ffn2 *gelu'   8192x4096x1024", 8192, 4096, 1024, 1, "dgelu", False, True),
    ("L dgrad ffn1 +res     8192x1024x4096", 8192, 1024, 4096, 1, "add", False, True),
    ("L dgrad qkv +res      8192x1024x3072", 8192, 1024, 3072, 1, "add", False, True),
    ("XL fwd qkv  +bias     25600x3072x1024", 25600, 3072, 1024, 0, "none", True, False),
    ("XL fwd out  +bias+res 25600x1024x1024", 25600, 1024, 1024, 0, "add", True, True),
    ("XL fwd ffn1 +bias+gelu 25600x4096x1024", 25600, 4096, 1024, 0, "gelu", True, False),
    ("XL fwd ffn2 +bias+res 25600x1024x4096", 25600, 1024, 4096, 0, "add", True, True),
    ("XL dgrad ffn2 *gelu'  25600x4096x1024", 25600, 4096, 1024, 1, "dgelu", False, True),
    ("XL dgrad ffn1 +res    25600x1024x4096", 25600, 1024, 4096, 1, "add", False, True),
    ("XL dgrad qkv +res     25600x1024x3072", 25600, 1024, 3072, 1, "add", False, True),
]


def epi_bench(args, impls):
    from cocodr_amd import _native as N
    codes = {"none": N.EPI_NONE, "add": N.EPI_ADD, "gelu": N.EPI_GELU, "dgelu": N.EPI_DGELU}
    print(f"{'shape (fused epilogue)':42s} " + " ".join(f"impl{i:d} TF/s(us)".rjust(18) for i in impls))
    for name, M, N_, K, tb, epi, has_bias, has_r in EPI_SHAPES:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        b = (torch.randn((K, N_) if tb else (N_, K), device="cuda") * 0.05).to(torch.bfloat16)
        bias = torch.randn(N_, device="cuda") if has_bias else None
        r = torch.randn(M, N_, device="cuda").to(torch.bfloat16) if has_r else None
        out = torch.empty(M, N_, dtype=torch.bfloat16, device="cuda")
        best = {i: 1e9 for i in impls}
        for rnd in range(args.rounds + 1):
            for i in impls:
                ops.gemm_set_impl(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.gemm(a, b, trans_b=bool(tb), bias=bias, epi=codes[epi], r=r, out=out)
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    best[i] = min(best[i], e0.elapsed_time(e1) / 5 * 1e3)
        flops = 2.0 * M * N_ * K
        print(f"{name:42s} " + " ".join(f"{flops / best[i] / 1e6:8.0f} ({best[i]:7.1f})".rjust(18) for i in impls), flush=True)
    ops.gemm_set_impl(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epi", action="store_true", help="time the encoder's fused-epilogue forms instead of plain GEMMs")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--impls", default="1,2,3")
    ap.add_argument("--shapes", default="", help="comma separated indices into SHAPES (default all)")
    args = ap.parse_args()
    impls = [int(x) for x in args.impls.split(",")]
    if args.epi:
        return epi_bench(args, impls)
    dev = "cuda"
    # impl -1 = torch.matmul (hipBLASLt / rocBLAS) on the same operands: the vendor library's rate on this box, as a yardstick only
    print(f"{'shape':34s} " + " ".join((f"impl{i:d} TF/s(us)" if i >= 0 else "torch TF/s(us)").rjust(18) for i in impls))
    sel = [int(x) for x in args.shapes.split(',')] if args.shapes else range(len(SHAPES))
    for name, M, N, K, ta, tb, nb, f32 in [SHAPES[i] for i in sel]:
        g = torch.Generator().manual_seed(0)
        ashape = (K, M) if ta else (M, K)
        bshape = (K, N) if tb else (N, K)
        if nb > 1:
            ashape, bshape = (nb,) + ashape, (nb,) + bshape
        a = torch.randn(ashape, generator=g).to(torch.bfloat16).to(dev)
        b = (torch.randn(bshape, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        out = torch.empty((nb, M, N) if nb > 1 else (M, N), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
        best = {i: 1e9 for i in impls}
        for r in range(args.rounds + 1):
            for i in impls:
                if i >= 0:
                    ops.gemm_set_impl(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if i < 0:
                    at = a.transpose(-1, -2) if ta else a
                    bt = b if tb else b.transpose(-1, -2)
                    lib_out = out if not f32 else torch.empty(out.shape, dtype=torch.bfloat16, device=dev)  # (bf16 result: no fp32-out library form via torch)
                e0.record()
                for _ in range(5):
                    if i < 0:
                        torch.matmul(at, bt, out=lib_out)
                    else:
                        ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb), out_f32=bool(f32), out=out)
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    best[i] = min(best[i], e0.elapsed_time(e1) / 5 * 1e3)
        flops = 2.0 * M * N * K * nb
        print(f"{name:34s} " + " ".join(f"{flops / best[i] / 1e6:8.0f} ({best[i]:7.1f})".rjust(18) for i in impls), flush=True)
    ops.gemm_set_impl(0)


if __name__ == "__main__":
    main()
