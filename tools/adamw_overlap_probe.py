"""How much of the AdamW pass hides under the next step's forward?  (cocodr-base 64 x 128 packed, the headline step.)
Times forward + optimizer pass back to back on one stream against the two on two streams without any dependency between them
(an upper bound for an update that runs range by range ahead of the forward)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cocodr_amd  # noqa: F401
import bench
from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel
from cocodr_amd.optim import FlatAdamW, clip_grad_norm_

dev = torch.device("cuda:0")
cfg = CocoBertConfig.base()
torch.manual_seed(0)
bert = CocoBertModel(cfg).to(dev)
model = CoCondenserForPretraining(bert)
opt = FlatAdamW.for_model(bert, lr=1e-5, weight_decay=0.01)
ids, mask, lens = bench.synth_batch_lens(1, 64, 128, cfg.vocab_size, dev, False)
batch = lambda: {"input_ids": ids, "attention_mask": mask}
flats = [bert.flat_decay, bert.flat_nodecay]
for _ in range(3):
    opt.zero_grad(set_to_none=True)
    loss = model(batch(), None); loss.backward()
    opt.step(clip=clip_grad_norm_(flats, 1.0))
torch.cuda.synchronize()
side = torch.cuda.Stream()


def t_ms(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


coef = clip_grad_norm_(flats, 1.0)
torch.cuda.synchronize()
def fwd():
    with torch.no_grad():
        bert.train()
        return model(batch(), None)
def step():
    opt.step(clip=coef)
def both_serial():
    step(); fwd()
def both_parallel():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    fwd()
    torch.cuda.current_stream().wait_stream(side)
print(f"forward alone {t_ms(fwd):.3f} ms   optimizer pass alone {t_ms(step):.3f} ms   one stream {t_ms(both_serial):.3f} ms   two streams {t_ms(both_parallel):.3f} ms")
