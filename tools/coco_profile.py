"""The full coCondenser step of bench.py alone (BERT-base, 64 x 128, 2 head layers, two MLM losses, clip + AdamW), the packed
contrastive step and the corpus encode - for rocprofv3 --kernel-trace --stats.  Usage: coco_profile.py [coco|packed|encode]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "coco"
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from cocodr_amd.modeling import CocoBertConfig
    cfg = CocoBertConfig.base(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    if which == "coco":
        ids, mask, lens = bench.synth_batch_lens(0, 64, 128, cfg.vocab_size, dev)
        print(bench.full_coco_step(cfg, dev, ids, mask, lens, steps=10, warmup=3, padded_too=False))
    elif which == "packed":
        dt, loss, roof, _, _, _ = bench.contrastive_leg("base", 64, 128, 10, 3, dev, 0, 1, False, 2, False, False, packed=True)
        print({"ms_per_step": dt / 10 * 1e3, "loss": loss})
    else:
        print(bench.corpus_encode(cfg, dev, seq_len=128))
