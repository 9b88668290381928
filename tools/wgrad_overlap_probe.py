"""Would the weight-gradient GEMMs fill the idle CUs of the dgrad chain?  One BERT-base backward's GEMMs at a packed row count:
(a) what the product does - the 12 layers' dgrad chains, then ONE merged weight-gradient launch; (b) the chains on the main stream
and each layer's (or each `group` layers') weight gradients as a merged launch on a side stream, released by an event when the
layer's chain is done.  us per backward, best of 3 x 5.
    python tools/wgrad_overlap_probe.py [--hidden 768] [--rows 5664] [--layers 12] [--group 1,2,3]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cocodr_amd  # noqa: E402,F401
from cocodr_amd import _native as N  # noqa: E402
from cocodr_amd._native import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=768)
    ap.add_argument("--rows", default="5664")
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--group", default="1,2,3,4")
    a_ = ap.parse_args()
    H, I, NL = a_.hidden, 4 * a_.hidden, a_.layers
    L = lib()
    g0 = torch.Generator().manual_seed(0)
    mk = lambda *s: (torch.randn(*s, generator=g0) * 0.3).to(torch.bfloat16).cuda()
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    for T in [int(x) for x in a_.rows.split(",")]:
        # dgrad chain of one layer (NN forms): (N, K, epi)
        chain = [(I, H, N.EPI_DGELU), (H, I, N.EPI_ADD), (H, H, N.EPI_NONE), (H, 3 * H, N.EPI_ADD)]
        dg = []
        keep = []
        for Nn, K, epi in chain:
            a, w, out = mk(T, K), mk(K, Nn), torch.empty(T, Nn, dtype=torch.bfloat16, device="cuda")
            r = mk(T, Nn) if epi != N.EPI_NONE else None
            g = N.GemmArgs()
            g.A, g.B, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
            if r is not None:
                g.R, g.ldr = r.data_ptr(), Nn
            g.M, g.N, g.K, g.lda, g.ldb, g.ldc = T, Nn, K, K, Nn, Nn
            g.trans_a, g.trans_b, g.epi, g.batch = 0, 1, epi, 1
            dg.append(g)
            keep += [a, w, out, r]
        shapes = [(3 * H, H), (H, H), (I, H), (H, I)]
        dys = [mk(NL, T, o) for o, _ in shapes]
        xs = [mk(NL, T, i) for _, i in shapes]
        outs = [torch.empty(NL, o, i, dtype=torch.float32, device="cuda") for o, i in shapes]

        def wproblems(l0, nl):
            arr = []
            for (o, i), dy, x, out in zip(shapes, dys, xs, outs):
                p = N.GemmArgs()
                p.A, p.B, p.C = dy[l0].data_ptr(), x[l0].data_ptr(), out[l0].data_ptr()
                p.M, p.N, p.K, p.lda, p.ldb, p.ldc = o, i, T, o, i, i
                p.trans_a = p.trans_b = p.out_f32 = 1
                p.epi, p.batch = N.EPI_NONE, nl
                p.strideA, p.strideB, p.strideC = T * o, T * i, o * i
                arr.append(p)
            return (N.GemmArgs * 4)(*arr)

        nws = L.cocodr_gemm_multi_workspace_floats()
        ws = torch.empty(nws, dtype=torch.float32, device="cuda")
        ws2 = torch.empty(nws, dtype=torch.float32, device="cuda")
        whole = wproblems(0, NL)

        def chain_once(sp):
            for g in dg:
                assert L.cocodr_gemm(C.byref(g), sp) == 0

        def product():
            sp = main_s.cuda_stream
            for _ in range(NL):
                chain_once(sp)
            assert L.cocodr_gemm_multi(whole, 4, ws.data_ptr(), nws, sp) == 0

        def chains_only():
            for _ in range(NL):
                chain_once(main_s.cuda_stream)

        def wgrad_only():
            assert L.cocodr_gemm_multi(whole, 4, ws.data_ptr(), nws, main_s.cuda_stream) == 0

        def overlapped(group):
            probs = [wproblems(l0, min(group, NL - l0)) for l0 in range(0, NL, group)]

            def run():
                sp = main_s.cuda_stream
                side.wait_stream(main_s)
                for l in range(NL):
                    chain_once(sp)
                    if (l + 1) % group == 0 or l == NL - 1:
                        ev = torch.cuda.Event()
                        ev.record(main_s)
                        side.wait_event(ev)
                        assert L.cocodr_gemm_multi(probs[l // group], 4, ws2.data_ptr(), nws, side.cuda_stream) == 0
                main_s.wait_stream(side)
            return run

        def time_us(fn, rounds=3, n=5):
            best = 1e9
            for _ in range(rounds):
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / n)
            return best

        tc, tw, tp = time_us(chains_only), time_us(wgrad_only), time_us(product)
        print(f"rows {T} H {H}: dgrad chains {tc:.0f} us, merged wgrad {tw:.0f} us, product order {tp:.0f} us", flush=True)
        for grp in [int(x) for x in a_.group.split(",")]:
            to = time_us(overlapped(grp))
            print(f"   wgrad per {grp} layer(s) on a side stream: {to:.0f} us ({(tp - to) / tp * 100:+.1f} % against the product order)", flush=True)


if __name__ == "__main__":
    main()
