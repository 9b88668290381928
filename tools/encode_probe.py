"""Corpus-encode leg alone (cocodr-large, packed batches of 1024 x L128, host-known lengths): wall time and, under rocprofv3, the kernel mix.
python tools/encode_probe.py [n_passages [batch]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cocodr_amd  # noqa: F401
from cocodr_amd import retrieval
from cocodr_amd.modeling import BertDotNLL, CocoBertConfig

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = CocoBertConfig.large()
torch.manual_seed(0)
model = BertDotNLL(cfg).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1234)
L = 128
lens = torch.clamp(torch.round(torch.randn(n, generator=g, device=dev) * 30 + 76), 8, L).to(torch.int64)
ids = torch.randint(1000, cfg.vocab_size, (n, L), generator=g, device=dev, dtype=torch.int32)
ids = torch.where(torch.arange(L, device=dev)[None] < lens[:, None], ids, torch.zeros_like(ids))
ids[:, 0] = 101
lens = lens.cpu()
retrieval.encode_corpus(model, ids[:2 * batch], None, batch_size=batch, lengths=lens[:2 * batch])
torch.cuda.synchronize()
t0 = time.perf_counter()
emb, _ = retrieval.encode_corpus(model, ids, None, batch_size=batch, lengths=lens)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"batch {batch}: encode {n} passages: {dt * 1e3:.1f} ms  {n / dt:.0f} passages/s  ({int(lens.sum())} tokens, {lens.sum().item() / dt / 1e6:.2f} M tokens/s)")
