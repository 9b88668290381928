#!/usr/bin/env python3
"""Probe (timing only, the overlapped variant races on the weights it reads): how much of the optimizer pass disappears when AdamW
of step n runs on a side stream under the forward of step n + 1 - the upper bound of a range-by-range pipelined update.
Usage: python tools/adamw_overlap_probe.py [base|large] [seqs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402,F401
from bench import synth_batch  # noqa: E402
from cocodr_amd.modeling import CocoBertConfig, CocoBertModel, _SimCEFn  # noqa: E402
from cocodr_amd.optim import FlatAdamW, clip_grad_norm_  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "base"
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
cfg = CocoBertConfig.base() if name == "base" else CocoBertConfig.large()
torch.manual_seed(0)
bert = CocoBertModel(cfg).to(dev)
bert.eval()
opt = FlatAdamW.for_model(bert, lr=1e-4, weight_decay=0.01)
pool = [synth_batch(10007 * i, n_seq, 128, cfg.vocab_size, dev) for i in range(8)]
flats = bert.flat_parameters()
side = torch.cuda.Stream(device=dev)


def step_serial(i):
    ids, mask = pool[i % 8]
    opt.zero_grad(set_to_none=True)
    cls = bert.encode_cls(ids, mask)
    loss, _ = _SimCEFn.apply(cls, 1, 0, cls.shape[0])
    loss.backward()
    opt.step(clip=clip_grad_norm_(flats, 1.0))
    return loss


def step_overlap(i):
    ids, mask = pool[i % 8]
    opt.zero_grad(set_to_none=True)
    cls = bert.encode_cls(ids, mask)
    loss, _ = _SimCEFn.apply(cls, 1, 0, cls.shape[0])
    loss.backward()
    clip = clip_grad_norm_(flats, 1.0)
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    for p in flats:
        p.grad.record_stream(side)
    clip.record_stream(side)
    with torch.cuda.stream(side):
        opt.step(clip=clip)
    return loss


def step_noopt(i):
    ids, mask = pool[i % 8]
    opt.zero_grad(set_to_none=True)
    cls = bert.encode_cls(ids, mask)
    loss, _ = _SimCEFn.apply(cls, 1, 0, cls.shape[0])
    loss.backward()
    clip_grad_norm_(flats, 1.0)
    return loss


for fn in (step_serial, step_overlap, step_noopt, step_serial, step_overlap, step_noopt):
    for i in range(4):
        loss = fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 16
    for i in range(n):
        loss = fn(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name} {n_seq} {fn.__name__}: {dt * 1e3:.3f} ms/step  {n_seq / dt:.1f} seq/s", flush=True)
