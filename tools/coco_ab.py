"""A/B switches of the full coCondenser step (BERT-base, 64 x 128, packed): decoder split-K.  ms per step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cocodr_amd.condenser import CondenserHead  # noqa: E402
from cocodr_amd.modeling import CocoBertConfig  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    cfg = CocoBertConfig.base(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    ids, mask, lens = bench.synth_batch_lens(0, 64, 128, cfg.vocab_size, dev)
    for rep in range(2):
        for sk in (2, 4, 8):
            CondenserHead.decoder_split_k = sk
            r = bench.full_coco_step(cfg, dev, ids, mask, lens, steps=20, warmup=5, padded_too=False)
            print(f"split_k {sk}: {r['ms_per_step']:.3f} ms  loss {r['loss']:.3f}", flush=True)
    CondenserHead.decoder_split_k = 8
