import os, sys, torch
sys.path.insert(0, "/root/repo")
import cocodr_amd
from cocodr_amd import ops
g = torch.Generator().manual_seed(0)
ok = True
for (M, N, K, f32) in [(256, 256, 32, 0), (256, 256, 64, 1), (512, 512, 160, 0), (2048, 1024, 1024, 0), (2000, 512, 96, 0), (1024, 1024, 4096, 1), (300, 256, 128, 0)]:
    a = torch.randn((M, K), generator=g).to(torch.bfloat16).cuda()
    b = (torch.randn((N, K), generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    outs = []
    for impl in (13, 20):
        ops.gemm_set_impl(impl)
        outs.append(ops.gemm(a, b, out_f32=bool(f32), bias=bias).clone())
    ref = (a.float() @ b.float().T + bias)
    same = torch.equal(outs[0], outs[1])
    err = float((outs[1].float() - ref).norm() / ref.norm())
    ok &= same
    print(f"M={M} N={N} K={K} f32={f32}: {'identical to impl 13' if same else 'DIFFERENT'}; rel err vs fp32 torch {err:.2e}", flush=True)
ops.gemm_set_impl(0)
print("OK" if ok else "MISMATCH")
