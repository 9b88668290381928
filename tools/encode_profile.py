#!/usr/bin/env python3
"""Corpus encode, padded vs packed, for a rocprofv3 --kernel-trace --stats run: python tools/encode_profile.py [--packed]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402
from cocodr_amd import retrieval  # noqa: E402
from cocodr_amd.modeling import BertDotNLL, CocoBertConfig  # noqa: E402
from bench import synth_batch  # noqa: E402

packed = "--packed" in sys.argv
cfg = CocoBertConfig.base()
model = BertDotNLL(cfg).cuda().eval()
ids, mask = synth_batch(0, 8192, 128, cfg.vocab_size, "cuda")
retrieval.encode_corpus(model, ids[:512], mask[:512], batch_size=512, pack=packed)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    retrieval.encode_corpus(model, ids, mask, batch_size=512, pack=packed)
torch.cuda.synchronize()
print("packed" if packed else "padded", "sequences/s", 3 * 8192 / (time.perf_counter() - t0))
