#!/usr/bin/env python3
"""Kernel-level profile of the iDRO re-weighted ANCE step (config 4 shapes).  Usage (GPU box): python tools/idro_profile.py"""
import os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa
from cocodr_amd.modeling import BertDotNLL, CocoBertConfig
from bench import synth_batch

dev = torch.device("cuda")
cfg = CocoBertConfig.large(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
torch.manual_seed(0)
model = BertDotNLL(cfg).to(dev)
model.add_group_loss(args=types.SimpleNamespace(model_size="large"), n_groups=50, dro_type="idro", alpha=0.25, eps=0.01, ema=0.1, rho=0.05)
rows = 32
q, qm = synth_batch(0, rows, 64, cfg.vocab_size, dev)
a, am = synth_batch(1, rows, 128, cfg.vocab_size, dev)
b, bm = synth_batch(2, rows, 128, cfg.vocab_size, dev)
groups = torch.randint(0, 50, (rows,), generator=torch.Generator().manual_seed(5)).to(dev)
for _ in range(2):
    r, *_ = model(q, qm, a, am, b, bm, group_ids=groups); r.backward(); model.bert.zero_grad(set_to_none=True)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
    r, *_ = model(q, qm, a, am, b, bm, group_ids=groups); r.backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=70))
