#!/usr/bin/env python3
"""Ablation of the direct-to-LDS GEMM main loop: builds libcocodr variants with the operand DMA, the LDS fragment
reads or the MFMAs compiled out (COCODR_ABL_* in csrc/gemm.hip) and times them on the encoder's shapes.  Results are
wrong by construction; only the timings mean anything.

  python tools/gemm_ablate.py --build          # here (hipcc cross-compiles)
  python tools/gemm_ablate.py --impls 3,5      # on the GPU box
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build", "abl")  # git-ignored, but travels to the GPU box
VARIANTS = {
    "full": [],
    "no_mfma": ["-DCOCODR_ABL_NO_MFMA"],
    "no_ldsread": ["-DCOCODR_ABL_NO_LDSREAD"],
    "no_dma": ["-DCOCODR_ABL_NO_DMA"],
    "dma_only": ["-DCOCODR_ABL_NO_MFMA", "-DCOCODR_ABL_NO_LDSREAD"],
    "mfma_only": ["-DCOCODR_ABL_NO_DMA", "-DCOCODR_ABL_NO_LDSREAD"],
    "epi_nostore": ["-DCOCODR_ABL_EPI_NOSTORE"],
    "no_barrier": ["-DCOCODR_ABL_NO_BARRIER"],
    "w4_burst": ["-DCOCODR_ABL_W4_BURST"],
}
TIMELINE = {"timeline": ["-DCOCODR_ABL_TIMELINE"]}


def build(only=None):
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, "coco-dr_amd", "csrc")
    for name, defs in {**VARIANTS, **TIMELINE}.items():
        if only and name not in only:
            continue
        lib = os.path.join(OUT, f"libabl_{name}.so")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
               "-DCOCODR_ABL_ALIAS_LD", "-DCOCODR_W4"] + defs + [
            os.path.join(csrc, "gemm.hip"), os.path.join(csrc, "gemm_pp.hip"), os.path.join(csrc, "gemm_w4.hip"), os.path.join(csrc, "core.hip"), os.path.join(csrc, "rowops.hip"), "-o", lib]
        subprocess.run(cmd, check=True)
        print("built", lib)


def timeline(args, impls, stream):
    import numpy as np
    import torch
    from cocodr_amd import _native
    from tools.gemm_bench import SHAPES
    lib = C.CDLL(os.path.join(OUT, "libabl_timeline.so"))
    lib.cocodr_gemm.argtypes = [C.POINTER(_native.GemmArgs), C.c_void_p]
    lib.cocodr_gemm_set_impl.argtypes = [C.c_int]
    for name, M, N, K, ta, tb, nb, f32 in [SHAPES[i] for i in ([int(x) for x in args.shapes.split(",")] if args.shapes else (2, 3, 1, 9))]:
        ashape = (nb, K, M) if ta else (nb, M, K)
        bshape = (nb, K, N) if tb else (nb, N, K)
        a = torch.randn(ashape, device="cuda").to(torch.bfloat16)
        b = (torch.randn(bshape, device="cuda") * 0.05).to(torch.bfloat16)
        out = torch.empty((nb, M, N), dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
        stamps = torch.zeros(1 << 16, 8, dtype=torch.int64, device="cuda")
        g = _native.GemmArgs(A=a.data_ptr(), B=b.data_ptr(), C=out.data_ptr(), C2=stamps.data_ptr(), M=M, N=N, K=K,
                             lda=a.shape[-1], ldb=b.shape[-1], ldc=N, trans_a=ta, trans_b=tb, out_f32=f32, batch=nb,
                             strideA=a[0].numel(), strideB=b[0].numel(), strideC=M * N)
        for impl in impls:
            lib.cocodr_gemm_set_impl(impl)
            for _ in range(3):
                stamps.zero_()
                assert lib.cocodr_gemm(C.byref(g), stream) == 0
                torch.cuda.synchronize()
            st = stamps.cpu().numpy()
            st = st[st[:, 0] > 0]
            print("   HW_REG_LDS_ALLOC values:", sorted(set(hex(int(x)) for x in st[:, 4]))[:8])
            ex = (st[:, :8].astype(np.float64) - st[:, 0].min()) / 100.0
            st = st[:, :4].astype(np.float64)
            t0 = st[:, 0].min()
            st = (st - t0) / 100.0  # us
            order = np.argsort(st[:, 0])
            st = st[order]
            ex = ex[order]
            n = len(st)
            print(f"== {name} impl {impl}: {n} workgroups, kernel span {st[:, 3].max():.1f} us")
            pro, loop, epi = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
            for lbl, v in (("start", st[:, 0]), ("prologue", pro), ("main loop", loop), ("epilogue", epi),
                           ("  epi: LDS write", ex[:, 5] - st[:, 2]), ("  epi: barrier", ex[:, 6] - ex[:, 5]), ("  epi: copy-out", st[:, 3] - ex[:, 6])):
                print(f"   {lbl:10s} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  "
                      f"p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
            # waves of dispatch: start-time histogram in 2-us bins
            hist, edges = np.histogram(st[:, 0], bins=np.arange(0, st[:, 0].max() + 2, 2.0))
            print("   starts per 2-us bin:", " ".join(str(int(x)) for x in hist))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--impls", default="3,5")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--ld64", action="store_true", help="NT shapes only: lda = ldb = 64, i.e. operands alias a ~1 MB L2-resident window")
    ap.add_argument("--variants", default="", help="comma separated subset of the built variants to time (default: the ablations)")
    ap.add_argument("--timeline", action="store_true", help="per-workgroup phase stamps instead of timings")
    ap.add_argument("--shapes", default="", help="comma separated indices into tools/gemm_bench.py SHAPES")
    args = ap.parse_args()
    if args.build:
        return build(args.variants.split(",") if args.variants else None)
    import torch
    from cocodr_amd import _native
    from tools.gemm_bench import SHAPES
    impls = [int(x) for x in args.impls.split(",")]
    libs = {}
    for name in (args.variants.split(",") if args.variants else VARIANTS):
        lib = C.CDLL(os.path.join(OUT, f"libabl_{name}.so"))
        lib.cocodr_gemm.argtypes = [C.POINTER(_native.GemmArgs), C.c_void_p]
        lib.cocodr_gemm.restype = C.c_int
        lib.cocodr_gemm_set_impl.argtypes = [C.c_int]
        libs[name] = lib
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if args.timeline:
        return timeline(args, impls, stream)
    print(f"{'shape':32s} impl " + " ".join(f"{n:>12s}" for n in libs) + "   (us per launch)")
    for name, M, N, K, ta, tb, nb, f32 in ([SHAPES[int(x)] for x in args.shapes.split(",")] if args.shapes else (SHAPES[:4] if args.ld64 else SHAPES[:11])):
        ashape = (nb, K, M) if ta else (nb, M, K)
        bshape = (nb, K, N) if tb else (nb, N, K)
        a = torch.randn(ashape, device="cuda").to(torch.bfloat16)
        b = (torch.randn(bshape, device="cuda") * 0.05).to(torch.bfloat16)
        out = torch.empty((nb, M, N), dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
        g = _native.GemmArgs(A=a.data_ptr(), B=b.data_ptr(), C=out.data_ptr(), M=M, N=N, K=K,
                             lda=64 if args.ld64 else a.shape[-1], ldb=64 if args.ld64 else b.shape[-1], ldc=N, trans_a=ta, trans_b=tb, out_f32=f32, batch=nb, strideA=a[0].numel(), strideB=b[0].numel(),
                             strideC=M * N)
        for impl in impls:
            res = []
            for vn, lib in libs.items():
                lib.cocodr_gemm_set_impl(impl)
                best = 1e9
                for r in range(args.rounds + 1):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        rc = lib.cocodr_gemm(C.byref(g), stream)
                        assert rc == 0
                    e1.record()
                    torch.cuda.synchronize()
                    if r:
                        best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
                res.append(best)
            print(f"{name:32s} {impl:4d} " + " ".join(f"{x:12.1f}" for x in res), flush=True)


if __name__ == "__main__":
    main()
