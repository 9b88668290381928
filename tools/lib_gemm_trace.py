#!/usr/bin/env python3
"""Which kernels the vendor library (torch.matmul -> hipBLASLt / rocBLAS) runs on the encoder's GEMM shapes: run under
`rocprofv3 --kernel-trace` and read kernel names (macro tile, MFMA shape) and the register / LDS / workgroup columns.
Yardstick only - nothing of the product calls the library."""
import torch

SHAPES = [("cube 8192", 8192, 8192, 8192), ("enc qkv", 65536, 2304, 768), ("XL fwd qkv", 25600, 3072, 1024), ("XL fwd out", 25600, 1024, 1024),
          ("XL fwd ffn2", 25600, 1024, 4096), ("L fwd ffn1", 8192, 4096, 1024), ("fwd qkv", 8192, 2304, 768), ("fwd out", 8192, 768, 768)]
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    for _ in range(3):
        c = torch.matmul(a, b.t())
    torch.cuda.synchronize()
    print(name, M, N, K, flush=True)
