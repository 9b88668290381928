#!/usr/bin/env python3
"""Kernel-level profile of the ANCE config-4 step (bench.py ance_step shapes).  Usage (GPU box): python tools/ance_step_profile.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa
from cocodr_amd.modeling import BertDotNLL, CocoBertConfig
from cocodr_amd.optim import FlatLamb, clip_grad_norm_
from bench import synth_batch

dev = torch.device("cuda")
cfg = CocoBertConfig.large(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
torch.manual_seed(0)
model = BertDotNLL(cfg).to(dev)
opt = FlatLamb.for_model(model.bert, lr=5e-6, eps=1e-8)
q, qm = synth_batch(0, 32, 64, cfg.vocab_size, dev)
a, am = synth_batch(1, 32, 128, cfg.vocab_size, dev)
b, bm = synth_batch(2, 32, 128, cfg.vocab_size, dev)
flats = [model.bert.flat_decay, model.bert.flat_nodecay]
def step():
    opt.zero_grad(set_to_none=True)
    loss, _a, _l = model(q, qm, a, am, b, bm)
    loss.backward()
    opt.step(clip=clip_grad_norm_(flats, 1.0))
for _ in range(3): step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=32, max_name_column_width=70))
