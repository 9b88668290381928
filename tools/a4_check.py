"""The hand-scheduled one-wave-per-SIMD GEMM (csrc/gemm_a4.hip, impl 14) against the shipped ping-pong pipeline (impl 13) and the
vendor library (torch.matmul, plain product; a yardstick, nothing in the product calls it): identical results?  us per launch, TFLOP/s.
  python tools/a4_check.py [--quick]"""
import ctypes as C
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import _native as N  # noqa: E402
from cocodr_amd._native import lib, stream_ptr  # noqa: E402

L = lib()
sp = stream_ptr()
IMPLS = [int(x) for x in os.environ.get("A4_IMPLS", "13,14").split(",")]


def time_many(fns, rounds=4, n=10, warm_ms=40.0):
    """us per launch of every fn: the chip is warmed up first (an idle GPU clocks down while the host builds the operands, and the
    first thing timed afterwards pays the ramp: 4-10 % - that bias made whatever was measured FIRST look slow in rounds 1-5's
    tables), then the candidates are timed in alternation, `rounds` times each, best of the rounds."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while True:
        for f in fns:
            f()
        e1.record()
        torch.cuda.synchronize()
        if e0.elapsed_time(e1) >= warm_ms:
            break
    best = [1e9] * len(fns)
    for _ in range(rounds):
        for k, f in enumerate(fns):
            f()
            e0.record()
            for _ in range(n):
                f()
            e1.record()
            torch.cuda.synchronize()
            best[k] = min(best[k], e0.elapsed_time(e1) * 1e3 / n)
    return best


shapes = [("tiny 256x256x256", 256, 256, 256, N.EPI_NONE), ("one tile K=1024", 256, 256, 1024, N.EPI_NONE), ("ragged rows", 1000, 512, 384, N.EPI_ADD),
          ("NS fwd qkv", 32768, 3072, 1024, N.EPI_NONE), ("NS fwd out +res", 32768, 1024, 1024, N.EPI_ADD), ("NS fwd ffn1 gelu", 32768, 4096, 1024, N.EPI_GELU),
          ("NS fwd ffn2 +res", 32768, 1024, 4096, N.EPI_ADD), ("XL fwd qkv", 25600, 3072, 1024, N.EPI_NONE), ("cube 8192", 8192, 8192, 8192, N.EPI_NONE),
          ("ragged 17896 qkv", 17896, 3072, 1024, N.EPI_NONE), ("base fwd qkv", 8192, 2304, 768, N.EPI_NONE), ("base fwd ffn1 gelu", 8192, 3072, 768, N.EPI_GELU),
          ("base fwd ffn2 +res", 8192, 768, 3072, N.EPI_ADD), ("packed base qkv", 5664, 2304, 768, N.EPI_NONE), ("packed large ffn1", 5664, 4096, 1024, N.EPI_GELU),
          ("enc qkv 65536", 65536, 2304, 768, N.EPI_NONE), ("fp32 out", 4096, 1024, 1024, -1)]
if "--sweep" in sys.argv:   # the encoder's forward (NT) forms over the row counts the steps run at
    shapes = []
    for H, tag in ((768, "base"), (1024, "large")):
        for rows in (1024, 2048, 4416, 4776, 5664, 8192, 15024, 25600, 32768):
            shapes += [(f"{tag} qkv {rows}", rows, 3 * H, H, N.EPI_NONE), (f"{tag} out+res {rows}", rows, H, H, N.EPI_ADD),
                       (f"{tag} ffn1 gelu {rows}", rows, 4 * H, H, N.EPI_GELU), (f"{tag} ffn2+res {rows}", rows, H, 4 * H, N.EPI_ADD)]
if "--sweep-nn" in sys.argv:   # the encoder's dgrad (NN) forms over the row counts the steps run at
    shapes = []
    for H, tag in ((768, "base"), (1024, "large")):
        for rows in (4416, 4776, 5664, 8192, 15024, 25600, 32768):
            shapes += [(f"{tag} dffn2 xgelu' {rows}", rows, 4 * H, H, N.EPI_DGELU, "nn", 1), (f"{tag} dffn1+res {rows}", rows, H, 4 * H, N.EPI_ADD, "nn", 1),
                       (f"{tag} dout {rows}", rows, H, H, N.EPI_NONE, "nn", 1), (f"{tag} dqkv+res {rows}", rows, H, 3 * H, N.EPI_ADD, "nn", 1)]
if "--forms" in sys.argv:   # the backward forms: dgrad (NN: B stored [K, N]) and the grouped weight gradient (TN, fp32 result)
    shapes = [("nn tiny", 256, 256, 256, N.EPI_NONE, "nn", 1), ("nn ragged +res", 1000, 512, 384, N.EPI_ADD, "nn", 1),
              ("NS dgrad ffn2 xgelu'", 32768, 4096, 1024, N.EPI_DGELU, "nn", 1), ("NS dgrad ffn1 +res", 32768, 1024, 4096, N.EPI_ADD, "nn", 1),
              ("NS dgrad out", 32768, 1024, 1024, N.EPI_NONE, "nn", 1), ("NS dgrad qkv +res", 32768, 1024, 3072, N.EPI_ADD, "nn", 1),
              ("base dgrad ffn2", 8192, 3072, 768, N.EPI_DGELU, "nn", 1), ("packed large dgrad qkv", 5664, 1024, 3072, N.EPI_ADD, "nn", 1),
              ("tn tiny", 256, 256, 256, -1, "tn", 1), ("tn ragged M", 1000, 512, 384, -1, "tn", 1), ("tn batch 3", 768, 1024, 512, -1, "tn", 3),
              ("NS wgrad qkv x4", 3072, 1024, 32768, -1, "tn", 4), ("NS wgrad ffn1 x4", 4096, 1024, 32768, -1, "tn", 4),
              ("NS wgrad ffn2 x4", 1024, 4096, 32768, -1, "tn", 4), ("NS wgrad out x8", 1024, 1024, 32768, -1, "tn", 8),
              ("base wgrad ffn1 x12", 3072, 768, 8192, -1, "tn", 12)]
if "--quick" in sys.argv:
    shapes = shapes[:5]
if os.environ.get("A4_ONLY"):
    keep = os.environ["A4_ONLY"].split(",")
    shapes = [s_ for s_ in shapes if any(k in s_[0] for k in keep)]
NOLIB = bool(os.environ.get("A4_NOLIB"))
g0 = torch.Generator().manual_seed(0)
bad = 0
for sh in shapes:
    name, M, Nn, K, epi = sh[:5]
    form, batch = (sh[5], sh[6]) if len(sh) > 5 else ("nt", 1)
    f32 = epi == -1
    epi = N.EPI_NONE if f32 else epi
    if form == "tn":   # A stored [K, M], B stored [K, N], per batch item
        a = torch.randn(batch, K, M, generator=g0).to(torch.bfloat16).cuda()
        w = (torch.randn(batch, K, Nn, generator=g0) * 0.03).to(torch.bfloat16).cuda()
    elif form == "nn":  # B stored [K, N]
        a = torch.randn(M, K, generator=g0).to(torch.bfloat16).cuda()
        w = (torch.randn(K, Nn, generator=g0) * 0.03).to(torch.bfloat16).cuda()
    else:
        a = torch.randn(M, K, generator=g0).to(torch.bfloat16).cuda()
        w = (torch.randn(Nn, K, generator=g0) * 0.03).to(torch.bfloat16).cuda()
    bias = torch.randn(Nn, generator=g0).cuda()
    r = torch.randn(M, Nn, generator=g0).to(torch.bfloat16).cuda() if epi in (N.EPI_ADD, N.EPI_DGELU) else None
    outs = []
    fns = []
    keep = []
    for impl in IMPLS:
        out = torch.zeros(batch, M, Nn, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
        c2 = torch.zeros(M, Nn, dtype=torch.bfloat16, device="cuda") if epi == N.EPI_GELU else None
        g = N.GemmArgs()
        g.A, g.B, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
        if form != "tn" and epi != N.EPI_DGELU:
            g.bias = bias.data_ptr()
        if c2 is not None:
            g.C2 = c2.data_ptr()
        if r is not None:
            g.R, g.ldr = r.data_ptr(), Nn
        g.M, g.N, g.K, g.ldc, g.batch, g.epi, g.out_f32 = M, Nn, K, Nn, batch, epi, int(f32)
        g.lda, g.ldb = (M if form == "tn" else K), (K if form == "nt" else Nn)
        g.trans_a, g.trans_b = int(form == "tn"), int(form != "nt")
        g.strideA, g.strideB, g.strideC = K * M, K * Nn, M * Nn

        def fn(g=g, impl=impl):
            L.cocodr_gemm_set_impl(impl)
            rc = L.cocodr_gemm(C.byref(g), sp)
            L.cocodr_gemm_set_impl(0)
            return rc
        assert fn() == 0
        torch.cuda.synchronize()
        outs.append((out.clone(), None if c2 is None else c2.clone()))
        out.zero_()
        fns.append(fn)
        keep.append((g, out, c2))
    if not NOLIB:
        if form == "tn":
            fns.append(lambda: torch.matmul(a.transpose(1, 2), w))
        else:
            fns.append(lambda: torch.matmul(a, w if form == "nn" else w.t()))
    ts = time_many(fns)
    t_lib = 1.0 if NOLIB else ts.pop()
    same = all(torch.equal(outs[0][0], o[0]) and (outs[0][1] is None or torch.equal(outs[0][1], o[1])) for o in outs[1:])
    d = max(float((outs[0][0].float() - o[0].float()).abs().max()) for o in outs[1:]) if len(outs) > 1 else 0.0
    bad += 0 if same else 1
    fl = 2.0 * M * Nn * K * batch
    cols = "   ".join(f"impl{i} {t:7.1f} us ({fl / t / 1e6:5.0f} TF)" for i, t in zip(IMPLS, ts))
    print(f"{name:20s} {M}x{Nn}x{K}: {cols}   library {t_lib:7.1f} us ({fl / t_lib / 1e6:5.0f} TF)   {'identical' if same else f'DIFFERENT (max abs {d:.3g})'}", flush=True)
print("check rc=%d" % bad)
sys.exit(1 if bad else 0)
