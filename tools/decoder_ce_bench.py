"""Fused decoder GEMM + cross entropy (cocodr_decoder_ce) against the two-kernel path (fp32 logits GEMM + cocodr_ce_fwd_bwd), at the
coCondenser step's shape (2 x 608 labelled rows padded, H = 768, V = 30 522) and inside the full step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cocodr_amd import ops  # noqa: E402
from cocodr_amd._native import check, lib, ptr, stream_ptr  # noqa: E402
from cocodr_amd.condenser import CondenserHead  # noqa: E402
from cocodr_amd.modeling import CocoBertConfig  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    for n, H, V in ((1216, 768, 30522), (2432, 768, 30522), (640, 1024, 30522)):
        vp = (V + 255) // 256 * 256
        t = torch.randn(n, H, device=dev).bfloat16()
        W = torch.zeros(vp, H, device=dev, dtype=torch.bfloat16)
        W[:V] = (torch.randn(V, H, device=dev) * 0.2).bfloat16()
        bias = torch.full((vp,), -1e30, device=dev)
        bias[:V] = 0.1
        lab = torch.randint(0, V, (n,), device=dev, dtype=torch.int32)
        sc = torch.full((n,), 1.0 / n, device=dev)
        vp128 = (V + 127) // 128 * 128
        W128, b128 = W[:vp128].contiguous(), bias[:vp128].contiguous()
        loss2 = torch.empty(n, device=dev)
        d2 = torch.empty((n, vp128), device=dev, dtype=torch.bfloat16)

        def two():
            lg = ops.gemm(t, W128, bias=b128, out_f32=True)
            check(lib().cocodr_ce_fwd_bwd(ptr(lg), ptr(lab), ptr(sc), n, V, vp128, ptr(loss2), ptr(d2), stream_ptr()), "ce")

        fused = timed(lambda: ops.decoder_ce(t, W, bias, lab, sc))
        unf = timed(two)
        flops = 2.0 * n * vp * H
        print(f"n={n} H={H} V={V}: fused {fused:.1f} us (2 GEMM passes: {2 * flops / fused / 1e6:.0f} TFLOP/s executed), two-kernel {unf:.1f} us")
    cfg = CocoBertConfig.base(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    ids, mask, lens = bench.synth_batch_lens(0, 64, 128, cfg.vocab_size, dev)
    for fused in (True, False, True, False):
        CondenserHead.fused_ce = fused
        r = bench.full_coco_step(cfg, dev, ids, mask, lens, steps=20, warmup=5, padded_too=False)
        print("fused_ce", fused, {k: r[k] for k in r if "ms" in k or "per_s" in k or k == "loss"})
