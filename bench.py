#!/usr/bin/env python3
"""bench.py - contrastive-step throughput of the native MI355X hot path (BASELINE.json metric 1).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full COCO contrastive training step on one batch of synthetic MS MARCO-shaped spans
(BASELINE.json configs[1]: cocodr-base / BERT-base-uncased shape, seq_len 128, 64 sequences per GPU, bf16
activations with fp32 accumulation, in-batch negatives): encoder forward -> last-layer [CLS] ->
(all-gather over ranks) -> span-pair InfoNCE -> encoder backward -> gradient all-reduce -> AdamW + linear
warm-up schedule.  Inputs are resident in HBM before the timed region.  Weak scaling: 64 sequences per GPU.

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events bracketing every launch of the
dominant kernel class (the bf16 MFMA GEMM, coco-dr_amd/csrc/gemm.hip) inside the timed region;
`cpu_baseline` times the numpy oracle (the CPU port of the same step, fwd+loss+bwd) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PROF_EVERY = 5  # timed steps between roofline samples
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
SEQ_PER_GPU = 64
SEQ_LEN = 128


def synth_batch(rank: int, n_seq: int, L: int, vocab: int, device):
    """ids ~ U{1000..V-1}, seed 1234+rank, lengths ~ clip(round(N(76,30)), 8, L), [CLS]=101 first,
    [SEP]=102 last, [PAD]=0 after."""
    rng = np.random.Generator(np.random.PCG64(1234 + rank))
    ids = rng.integers(1000, vocab, (n_seq, L))
    lens = np.clip(np.rint(rng.normal(76, 30, n_seq)), 8, L).astype(np.int64)
    mask = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    ids = ids * mask
    ids[:, 0] = 101
    ids[np.arange(n_seq), lens - 1] = 102
    return torch.from_numpy(ids).to(device), torch.from_numpy(mask).to(device)


def train_flops_per_seq(cfg, L: int) -> float:
    """forward N*(24 H^2 + 4 L H) FLOP per token, training = 3x forward."""
    H, N = cfg.hidden_size, cfg.num_hidden_layers
    return 3.0 * L * N * (24.0 * H * H + 4.0 * L * H)


def cpu_baseline(n_seq: int = 8, L: int = SEQ_LEN, steps: int = 2):
    """The oracle (numpy port of the same contrastive step: encoder fwd + InfoNCE + encoder bwd, fp32) timed on the
    host cores of this box.  Checker/baseline only - never on the product path."""
    import oracle as O
    ocfg = O.OracleConfig()
    P = O.make_params(ocfg, 0)
    rng = np.random.Generator(np.random.PCG64(0))
    ids = rng.integers(1000, ocfg.vocab_size, (n_seq, L))
    mask = np.ones((n_seq, L), np.int64)

    def step():
        hs, cache = O.encoder_fwd(P, ocfg, ids, mask, keep_cache=True)
        _, dE = O.contrastive_loss_grad(O.cls_embedding(hs[-1]).copy(), 1)
        d_last = np.zeros_like(hs[-1])
        d_last[:, 0] = dE
        O.encoder_bwd(P, ocfg, cache, d_last)

    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": n_seq / dt, "unit": "sequences/sec", "cores": int(cores), "kind": "port",
            "sample": f"{steps} steps of {n_seq} sequences x L{L}, BERT-base, numpy fp32 oracle fwd+loss+bwd (no optimizer)"}


def full_coco_step(cfg, args, dev, ids, mask, steps: int = 8, warmup: int = 3):
    """The reference's whole pre-training step (COCO/modeling.py:192-235 with COCO/README.md:49 settings: 2 Condenser
    head layers, skip_from 6, late MLM): backbone + head + two label-sparse MLM losses + contrastive + AdamW.
    Reported next to the headline metric, never instead of it."""
    import types
    from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertModel
    from cocodr_amd.optim import FlatAdamW
    torch.manual_seed(0)
    bert = CocoBertModel(cfg).to(dev)
    margs = types.SimpleNamespace(n_head_layers=2, skip_from=min(6, cfg.num_hidden_layers), late_mlm=True)
    model = CoCondenserForPretraining(bert, margs).to(dev)
    opt = FlatAdamW.for_model(bert, lr=1e-4, weight_decay=0.01)
    opt_h = torch.optim.AdamW(model.c_head.param_groups(0.01), lr=1e-4, fused=True)
    g = torch.Generator().manual_seed(5)
    pick = (torch.rand(ids.shape, generator=g) < 0.15).to(dev) & (mask > 0)
    pick[:, 0] = False
    labels = torch.where(pick, ids, torch.full_like(ids, -100))
    inp = torch.where(pick, torch.full_like(ids, 103), ids)  # [MASK]
    batch = {"input_ids": inp, "attention_mask": mask}

    from cocodr_amd.optim import clip_grad_norm_
    head_params = [p for g_ in model.c_head.param_groups(0.01) for p in g_["params"]]
    all_flats = [bert.flat_decay, bert.flat_nodecay] + head_params

    def step():
        opt.zero_grad(set_to_none=True)
        opt_h.zero_grad(set_to_none=True)
        loss = model(batch, labels)
        loss.backward()
        clip = clip_grad_norm_(all_flats, 1.0)  # HF Trainer default max_grad_norm, over backbone + head, on the device
        opt.step(clip=clip)
        for p in head_params:
            p.grad.mul_(clip[1])
        opt_h.step()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"sequences_per_sec": round(ids.shape[0] / dt, 1), "ms_per_step": round(dt * 1e3, 3), "loss": round(float(loss.detach()), 3),
            "scope": "backbone + 2 Condenser head layers (skip_from 6) + head & late MLM losses (label-sparse, 15 %) + contrastive + clip_grad_norm_(1.0) + AdamW"}


def eval_search(dev, nq: int = 2048, npass: int = 125000, dim: int = 1024, k: int = 1000, iters: int = 3):
    """BASELINE.json's second metric on one shard of config 5 (cocodr-large width, 125 k passages per GPU, k = 1000):
    query x passage dot-products/sec = Nq*Np / wall time of (exact fp32 score + exact top-k), embeddings resident in HBM."""
    from cocodr_amd import ops
    g = torch.Generator().manual_seed(7)
    Q = (torch.randn(nq, dim, generator=g) / dim ** 0.5).to(dev)
    P = (torch.randn(npass, dim, generator=g) / dim ** 0.5).to(dev)
    ws = torch.empty(ops.lib().cocodr_score_topk_workspace_bytes(nq, npass, k), dtype=torch.uint8, device=dev)
    ops.score_topk(Q, P, k, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ops.score_topk(Q, P, k, workspace=ws)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"dot_products_per_sec": round(nq * npass / dt), "ms": round(dt * 1e3, 2),
            "workload": f"{nq} queries x {npass} passages x {dim} fp32, k={k}, exact scores + exact top-k, 1 GPU shard of config 5"}


def ance_step(dev, rows: int = 32, steps: int = 10, warmup: int = 3):
    """BASELINE config 4 (ANCE/drivers/run_ann.py:293-356): BERT-large triplet step, 32 rows/GPU = queries [32,64] +
    positives / negatives [32,128], backward, clip_grad_norm_(1.0), LAMB (the reference's default optimizer), linear
    schedule.  One training row = 3 sequences."""
    from cocodr_amd.modeling import BertDotNLL, CocoBertConfig
    from cocodr_amd.optim import FlatLamb, clip_grad_norm_
    cfg = CocoBertConfig.large()
    torch.manual_seed(0)
    model = BertDotNLL(cfg).to(dev)
    opt = FlatLamb.for_model(model.bert, lr=5e-6, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: max(0.0, 1.0 - s / 1000.0))
    q, qm = synth_batch(0, rows, 64, cfg.vocab_size, dev)
    a, am = synth_batch(1, rows, 128, cfg.vocab_size, dev)
    b, bm = synth_batch(2, rows, 128, cfg.vocab_size, dev)
    flats = [model.bert.flat_decay, model.bert.flat_nodecay]

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _acc, _logits = model(q, qm, a, am, b, bm)
        loss.backward()
        opt.step(clip=clip_grad_norm_(flats, 1.0))  # norm and coefficient stay on the device
        sched.step()
        return loss

    for _ in range(warmup):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"sequences_per_sec": round(3 * rows / dt, 1), "rows_per_sec": round(rows / dt, 1), "ms_per_step": round(dt * 1e3, 3),
           "loss": round(float(loss.detach()), 4),
           "scope": f"cocodr-large triplet step, {rows} rows (q L64 + pos/neg L128), bf16, clip_grad_norm_(1.0) + LAMB; BASELINE configs[3]"}
    # the same step with iDRO re-weighting: 50 query clusters, per-group gradients of the last 2 layers
    import types
    n_groups = 50
    model.add_group_loss(args=types.SimpleNamespace(model_size="large"), n_groups=n_groups, dro_type="idro", alpha=0.25, eps=0.01,
                         ema=0.1, rho=0.05)
    groups = torch.randint(0, n_groups, (rows,), generator=torch.Generator().manual_seed(5)).to(dev)

    def idro_step():
        opt.zero_grad(set_to_none=True)
        robust, _acc, _gl, _gc = model(q, qm, a, am, b, bm, group_ids=groups)
        robust.backward()
        opt.step(clip=clip_grad_norm_(flats, 1.0))
        return robust

    for _ in range(2):
        idro_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(3, steps // 2)):
        idro_step()
    torch.cuda.synchronize()
    dti = (time.perf_counter() - t0) / max(3, steps // 2)
    out["idro"] = {"ms_per_step": round(dti * 1e3, 3), "rows_per_sec": round(rows / dti, 1),
                   "groups_present": int(groups.unique().numel()), "n_groups": n_groups}
    return out


def corpus_encode(cfg, dev, n: int = 8192, seq_len: int = 128, batch: int = 512, iters: int = 3):
    """Inference half of the path (ANCE/drivers/run_ann_data_gen.py:157-212): eval-mode passage embeddings of a token
    cache resident in HBM, kept on device (retrieval.encode_corpus)."""
    from cocodr_amd import retrieval
    from cocodr_amd.modeling import BertDotNLL
    model = BertDotNLL(cfg).to(dev).eval()
    ids, mask = synth_batch(0, n, seq_len, cfg.vocab_size, dev)
    retrieval.encode_corpus(model, ids[:batch], mask[:batch], batch_size=batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        emb, _ = retrieval.encode_corpus(model, ids, mask, batch_size=batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"sequences_per_sec": round(n / dt, 1), "ms": round(dt * 1e3, 2),
            "workload": f"{n} passages x L{seq_len}, batch {batch}, BertDot_NLL_LN body_emb (last-layer [CLS]), bf16 encoder, eval mode"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="base", choices=["base", "large"])
    ap.add_argument("--seq-per-gpu", type=int, default=SEQ_PER_GPU)
    ap.add_argument("--seq-len", type=int, default=SEQ_LEN)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-full-step", action="store_true", help="skip the extra full-coCondenser-step and eval-search measurements")
    ap.add_argument("--dp-chunks", type=int, default=2, help="layer ranges whose gradient all-reduce overlaps the backward")
    args = ap.parse_args()

    import torch.distributed as dist
    import cocodr_amd  # noqa: F401
    from cocodr_amd import ops
    from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or bool(os.environ.get("COCODR_FORCE_DIST"))  # FORCE: exercise the N>1 path on one GPU
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    cfg = CocoBertConfig.base() if args.model == "base" else CocoBertConfig.large()
    torch.manual_seed(0)  # identical random-init weights on every rank
    bert = CocoBertModel(cfg).to(dev)
    model = CoCondenserForPretraining(bert)
    if use_dist:
        bert.enable_grad_allreduce(chunks=args.dp_chunks)  # averaged inside the backward, overlapped with it
    from cocodr_amd.optim import FlatAdamW
    opt = FlatAdamW.for_model(bert, lr=1e-4, weight_decay=0.01)  # torch.optim.AdamW semantics, one native pass per flat
    total = args.steps + args.warmup
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: min(1.0, (s + 1) / max(1, int(0.1 * total))))
    # a small pool of different synthetic batches, resident in HBM before the timed region, visited round-robin (one
    # repeated batch is memorised within a few steps and the loss saturates at 0)
    pool = [synth_batch(rank + 10007 * i, args.seq_per_gpu, args.seq_len, cfg.vocab_size, dev) for i in range(8)]
    ids, mask = pool[0]
    batches = [{"input_ids": i_, "attention_mask": m_} for i_, m_ in pool]
    step_no = [0]

    from cocodr_amd.optim import clip_grad_norm_
    flats = [bert.flat_decay, bert.flat_nodecay]

    def step():
        batch = batches[step_no[0] % len(batches)]
        step_no[0] += 1
        opt.zero_grad(set_to_none=True)
        loss = model(batch, None)
        loss.backward()  # with N > 1 the gradient all-reduce is issued chunk by chunk inside this call
        # HF Trainer clips to max_grad_norm = 1.0 by default before optimizer.step() (COCO/run_coco_pre_training.py drives
        # the stock training loop); norm and coefficient stay on the device, the AdamW pass applies the coefficient
        opt.step(clip=clip_grad_norm_(flats, 1.0))
        sched.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    fence()
    if not args.no_roofline and rank == 0:
        ops.prof_begin(1)  # HIP events around every GEMM launch, on the launch stream
    # the event pairs cost stream time (~7 % of the step when every GEMM launch of every step carries one), so the
    # roofline leg brackets the GEMM launches of every PROF_EVERY-th timed step only
    t0 = time.perf_counter()
    for i in range(args.steps):
        if not args.no_roofline and rank == 0:
            ops.prof_pause(i % PROF_EVERY != 0)
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    roof = None
    if not args.no_roofline and rank == 0:
        n_launch, gemm_ms, gemm_flops = ops.prof_end()
        if n_launch and gemm_ms > 0:
            ach = gemm_flops / (gemm_ms * 1e-3) / 1e12
            traffic = None  # HBM-side bytes per GEMM launch from the committed PMC passes (profiles/r01c_*.md)
            if args.model == "base" and args.seq_per_gpu == SEQ_PER_GPU and args.seq_len == SEQ_LEN:  # the PMC passes ran on this shape
                try:
                    with open(os.path.join(ROOT, "profiles", "r01_gemm_pmc.json")) as f:
                        traffic = round(json.load(f)["hbm_bytes_per_launch"])
                except Exception:
                    pass
            roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
                    "kernel": "gemm_glds_kernel (bf16 MFMA 32x32x16, all NT/NN/TN launches)",
                    "launches_per_step": n_launch // max(1, len(range(0, args.steps, PROF_EVERY))),
                    "sampled": f"every GEMM launch of every {PROF_EVERY}th timed step ({n_launch} launches)",
                    "avg_launch_us": round(gemm_ms * 1e3 / n_launch, 2),
                    "gemm_share_of_step": round(gemm_ms / len(range(0, args.steps, PROF_EVERY)) / (dt / args.steps * 1e3), 3)}
    full = None
    if not args.no_full_step and not use_dist:
        full = full_coco_step(cfg, args, dev, ids, mask)  # second scope: what the reference's step really runs
    search = eval_search(dev) if (not args.no_full_step and not use_dist) else None
    encode = corpus_encode(cfg, dev, seq_len=args.seq_len) if (not args.no_full_step and not use_dist) else None
    ance = ance_step(dev) if (not args.no_full_step and not use_dist and args.model == "base") else None
    if use_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    final_loss = float(loss.detach())

    if rank == 0:
        n_seq = args.seq_per_gpu * world * args.steps
        value = n_seq / dt
        step_tflops = value * train_flops_per_seq(cfg, args.seq_len) / 1e12
        out = {
            "metric": "contrastive-step sequences/sec", "value": round(value, 2), "unit": "sequences/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"cocodr-{args.model} contrastive step (COCO in-batch negatives), seq_len={args.seq_len}, "
                                   f"{args.seq_per_gpu} sequences/GPU, bf16 + fp32 accumulate, clip_grad_norm_(1.0) + AdamW; "
                                   + ("BASELINE configs[1]" if args.model == "base" and world == 1 else
                                      "BASELINE configs[2] shape per GPU" if args.model == "base" else "north_star BERT-large target shape"),
                       "global_batch": args.seq_per_gpu * world, "seq_len": args.seq_len,
                       "batches": "8 pre-generated synthetic batches per rank, resident in HBM, visited round-robin",
                       "parallelism": f"dp{world}" + (" + RCCL all_gather negatives" if world > 1 else "")},
            "loss": round(final_loss, 4),
            "algorithmic_tflops_whole_step": round(step_tflops, 1),
            "whole_step_frac_of_mfma_peak": round(step_tflops / (MFMA_BF16_PEAK_TFLOPS * world), 4),
        }
        if roof is not None:
            out["roofline"] = roof
        if full is not None:
            out["full_coco_step"] = full
        if search is not None:
            out["eval_search"] = search
        if encode is not None:
            out["corpus_encode"] = encode
        if ance is not None:
            out["ance_triplet_step"] = ance
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
