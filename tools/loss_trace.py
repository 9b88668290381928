"""Per-step losses of the BERT-large contrastive step of bench.py (run-to-run determinism / stability check).
Usage: loss_trace.py [seq_per_gpu] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel  # noqa: E402
from cocodr_amd.optim import FlatAdamW, clip_grad_norm_  # noqa: E402

if __name__ == "__main__":
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    model_name = sys.argv[3] if len(sys.argv) > 3 else "large"
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    cfg = CocoBertConfig.large(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1) if model_name == "large" else \
        CocoBertConfig.base(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(1234)
    bert = CocoBertModel(cfg).to(dev)
    model = CoCondenserForPretraining(bert).to(dev)
    opt = FlatAdamW.for_model(bert, lr=1e-4, weight_decay=0.01)
    pool = [bench.synth_batch(10007 * i, nseq, 128, cfg.vocab_size, dev) for i in range(8)]
    flats = [bert.flat_decay, bert.flat_nodecay]
    losses = []
    for s in range(steps):
        ids, mask = pool[s % 8]
        opt.zero_grad(set_to_none=True)
        loss = model({"input_ids": ids, "attention_mask": mask}, None)
        loss.backward()
        opt.step(clip=clip_grad_norm_(flats, 1.0))
        losses.append(round(float(loss.detach()), 4))
    print(losses)
