"""Streaming kernels of the optimizer step alone: AdamW / gradient-norm (110 M parameters) and LAMB (335 M), us and TB/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa
from cocodr_amd.optim import FlatAdamW, FlatLamb, clip_grad_norm_
from cocodr_amd.modeling import CocoBertConfig, CocoBertModel


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, cfg, Opt in (("base AdamW", CocoBertConfig.base(), FlatAdamW), ("large LAMB", CocoBertConfig.large(), FlatLamb)):
    m = CocoBertModel(cfg).cuda()
    n = m.flat_decay.numel() + m.flat_nodecay.numel()
    m.flat_decay.grad = torch.randn_like(m.flat_decay) * 1e-3
    m.flat_nodecay.grad = torch.randn_like(m.flat_nodecay) * 1e-3
    opt = Opt.for_model(m, lr=1e-5, weight_decay=0.01)
    t_clip = timed(lambda: clip_grad_norm_([m.flat_decay, m.flat_nodecay], 1.0))
    t_step = timed(lambda: opt.step())
    print(f"{name}: {n / 1e6:.0f} M parameters: clip_grad_norm_ {t_clip:.0f} us ({n * 4 / t_clip / 1e6:.2f} TB/s), step {t_step:.0f} us", flush=True)
    del m, opt
