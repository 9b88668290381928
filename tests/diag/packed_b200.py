"""Gradient differences packed vs padded at 200 sequences of BERT-large width, with and without the split-tail GEMM route."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import cocodr_amd
from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel
from test_gpu_packed import cfg_small, ragged_batch, t, rel_l2, DEV

H, heads, I, B = 1024, 16, 4096, int(os.environ.get("B", 200))
cfgd = cfg_small(hidden_size=H, num_attention_heads=heads, intermediate_size=I, num_hidden_layers=2, vocab_size=3000)
ids, mask, lens = ragged_batch(B, 128, 3000, 77)
res = {}
for packed in (False, True):
    torch.manual_seed(0)
    m = CocoBertModel(CocoBertConfig(**cfgd)).to(DEV)
    with torch.no_grad():
        s_ln = float(np.sqrt(5.0 / H))
        for k in ("weight", "bias"):
            m.hf_view(f"encoder.layer.1.output.LayerNorm.{k}").mul_(s_ln)
        m.flat_nodecay.add_(0.02)
    m.pack_sequences = packed
    model = CoCondenserForPretraining(m)
    batch = {"input_ids": t(ids), "attention_mask": t(mask)}
    if packed:
        batch["lengths"] = torch.from_numpy(lens)
    loss = model(batch, None)
    loss.backward()
    res[packed] = (float(loss.detach()), {k: v.detach().clone() for k, v in m.hf_named_grads()})
print("loss", res[True][0], res[False][0], "T", int(((np.maximum(lens,1)+31)//32*32).sum()))
for name, ref in res[False][1].items():
    d = rel_l2(res[True][1][name], ref)
    if d > 1e-3:
        print(f"{name:60s} {d:.2e}  |ref| {float(ref.norm()):.3e}")
