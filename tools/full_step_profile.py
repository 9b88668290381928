#!/usr/bin/env python3
"""Kernel-level profile of the full coCondenser step (bench.py full_coco_step shapes).  Usage (GPU box): python tools/full_step_profile.py"""
import os, sys, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa
from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel
from cocodr_amd.optim import FlatAdamW
from bench import synth_batch

dev = torch.device("cuda")
cfg = CocoBertConfig.base(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
torch.manual_seed(0)
bert = CocoBertModel(cfg).to(dev)
model = CoCondenserForPretraining(bert, types.SimpleNamespace(n_head_layers=2, skip_from=6, late_mlm=True)).to(dev)
opt = FlatAdamW.for_model(bert, lr=1e-4, weight_decay=0.01)
ids, mask = synth_batch(0, 64, 128, cfg.vocab_size, dev)
g = torch.Generator().manual_seed(5)
pick = (torch.rand(ids.shape, generator=g) < 0.15).to(dev) & (mask > 0)
pick[:, 0] = False
labels = torch.where(pick, ids, torch.full_like(ids, -100))
batch = {"input_ids": torch.where(pick, torch.full_like(ids, 103), ids), "attention_mask": mask}
def step():
    opt.zero_grad(set_to_none=True)
    loss = model(batch, labels); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
