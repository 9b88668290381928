"""Host-side time of the phases of the headline step (cocodr-base, 64 x 128, packed, the reference's batch): how long the Python
thread spends in model(batch), loss.backward(), the optimizer - without any synchronisation - next to the GPU's step time.
python tools/host_phase_probe.py [steps] [--host-lengths]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cocodr_amd  # noqa: F401
import bench
from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel
from cocodr_amd.optim import FlatAdamW, clip_grad_norm_

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
host_lengths = "--host-lengths" in sys.argv
dev = torch.device("cuda", 0)
cfg = CocoBertConfig.base()
torch.manual_seed(0)
bert = CocoBertModel(cfg).to(dev)
model = CoCondenserForPretraining(bert)
opt = FlatAdamW.for_model(bert, lr=1e-4, weight_decay=0.01)
pool = [bench.synth_batch_lens(10007 * i, 64, 128, cfg.vocab_size, dev, False) for i in range(8)]
flats = [bert.flat_decay, bert.flat_nodecay]
T = []
def step(i):
    ids_, mask_, lens_ = pool[i % 8]
    b = {"input_ids": ids_, "attention_mask": mask_}
    if host_lengths:
        b["lengths"] = lens_
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = model(b, None)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step(clip=clip_grad_norm_(flats, 1.0))
    t3 = time.perf_counter()
    return t0, t1, t2, t3
for i in range(5):
    step(i)
torch.cuda.synchronize()
w0 = time.perf_counter()
for i in range(steps):
    T.append(step(i))
torch.cuda.synchronize()
w1 = time.perf_counter()
a = np.array(T)
print(f"host_lengths={host_lengths}  GPU-inclusive step {1e3 * (w1 - w0) / steps:.3f} ms; host: forward {1e3 * np.median(a[:, 1] - a[:, 0]):.3f} ms, "
      f"backward() {1e3 * np.median(a[:, 2] - a[:, 1]):.3f} ms, clip + optimizer {1e3 * np.median(a[:, 3] - a[:, 2]):.3f} ms, "
      f"between steps {1e3 * np.median(a[1:, 0] - a[:-1, 3]):.3f} ms")
