"""Does a streaming optimizer pass over a slice that the previous launch just touched run faster than one over cold memory?
AdamW (30 B / parameter, non-temporal loads / stores) on slices of n parameters: the SAME slice every launch against a
round-robin over 40 slices (1.3 GB ... 5 GB of state: cold every time).  us per launch, TB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402,F401
from cocodr_amd._native import check, lib, ptr, stream_ptr  # noqa: E402

NS = 40
for n in (2_000_000, 4_000_000, 8_000_000, 16_000_000):
    tot = n * NS
    p = torch.randn(tot, device="cuda")
    g = torch.randn(tot, device="cuda") * 1e-3
    m = torch.zeros(tot, device="cuda")
    v = torch.zeros(tot, device="cuda")

    def step(k, it):
        o = k * n * 4
        check(lib().cocodr_adamw_step(p.data_ptr() + o, g.data_ptr() + o, m.data_ptr() + o, v.data_ptr() + o, None, 0, n, 1e-4, 0.9, 0.999, 1e-8, 0.01,
                                      it + 1, 1.0, None, stream_ptr()), "adamw")

    res = {}
    for mode in ("same slice", "round robin"):
        for it in range(NS):
            step(0 if mode == "same slice" else it % NS, it)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(2 * NS):
            step(0 if mode == "same slice" else it % NS, it)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / (2 * NS) * 1e3
    print(f"{n / 1e6:.0f} M parameters per launch ({n * 16 / 1e6:.0f} MB of p, g, m, v): same slice {res['same slice']:.1f} us ({n * 28 / res['same slice'] / 1e6:.2f} TB/s), "
          f"cold slices {res['round robin']:.1f} us ({n * 28 / res['round robin'] / 1e6:.2f} TB/s)", flush=True)
    del p, g, m, v
