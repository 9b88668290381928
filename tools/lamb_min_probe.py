import sys, torch
sys.path.insert(0, '/root/repo')
import cocodr_amd
from cocodr_amd import optim
from cocodr_amd.optim import FlatLamb, clip_grad_norm_
from cocodr_amd.modeling import CocoBertConfig, CocoBertModel
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
m = CocoBertModel(CocoBertConfig.large()).cuda()
m.flat_decay.grad = torch.randn_like(m.flat_decay) * 1e-3
m.flat_nodecay.grad = torch.randn_like(m.flat_nodecay) * 1e-3
for mn in (1 << 18, 2 << 20, 1 << 40):
    optim.LAMB_FUSED_MIN = mn
    import inspect
    opt = FlatLamb.for_model(m, lr=1e-5, weight_decay=0.01)
    # lamb_fused_plan takes min_len default at def time: patch through _plan by monkeypatching the function default
    optim.lamb_fused_plan.__defaults__ = (mn,)
    print(mn, round(timed(lambda: opt.step())), "us", flush=True)
