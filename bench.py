#!/usr/bin/env python3
"""bench.py - contrastive-step throughput of the native MI355X hot path (BASELINE.json metric 1).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...                  (re-launches itself under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full COCO contrastive training step on one batch of synthetic MS MARCO-shaped spans
(BASELINE.json configs[1]: cocodr-base / BERT-base-uncased shape, seq_len 128, 64 sequences per GPU, bf16
activations with fp32 accumulation, in-batch negatives): encoder forward -> last-layer [CLS] ->
(all-gather over ranks) -> span-pair InfoNCE -> encoder backward -> gradient all-reduce -> clip + AdamW + linear
warm-up schedule.  Inputs are resident in HBM before the timed region.  Weak scaling: 64 sequences per GPU.

The LAST line rank 0 writes to stdout is the contract object, compact (< 4 KB): metric / value / unit / n_gpus / steps / warmup /
ms_per_step / dtype / data / config, `roofline` (numbers only; measured live with HIP events bracketing every launch of the
dominant kernel class - the bf16 MFMA GEMM, coco-dr_amd/csrc/gemm.hip + gemm_pp.hip - inside the timed region), ONE
`cpu_baseline` block (the same step on the host cores: torch-CPU HuggingFace BertModel + restated loss + AdamW) and `summary`
(one number per side leg).  Every side leg in full - `north_star_large_step` (the BERT-large seq-128 step the north star names),
`host_lengths_contrastive_step`, `padded_contrastive_step`, `full_coco_step`, `ance_triplet_step`, `corpus_encode`, `eval_search`
(BASELINE.json's second metric at the real config-5 shard), `config5_end_to_end`, `multi_gpu`, the other CPU baselines - goes to
`bench_legs.json` next to this file (and to gpurun_out/ when that directory exists) and, one leg per line prefixed `[leg]`, to stderr.
"""
import argparse
import json
import os
import sys
import time
import warnings

# nothing but the contract line may look like output; the known noisy categories are silenced - NOT RuntimeWarning / our own
# warnings (e.g. the DDP-adoption or LAMB fall-back messages of coco-dr_amd), which go to stderr and are worth reading
for _cat in (UserWarning, FutureWarning, DeprecationWarning):
    warnings.filterwarnings("ignore", category=_cat)

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PROF_EVERY = 5  # timed steps between roofline samples
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3    # fp32-input MFMA peak (v_mfma_f32_32x32x2_f32), same guide
SEQ_PER_GPU = 64
SEQ_LEN = 128


def synth_batch_lens(rank: int, n_seq: int, L: int, vocab: int, device, dense: bool = False):
    """ids ~ U{1000..V-1}, seed 1234+rank, lengths ~ clip(round(N(76,30)), 8, L) (MS MARCO-shaped; (24, 8)
    at L <= 64), [CLS]=101 first, [SEP]=102 last, [PAD]=0 after.  ``dense``: every sequence fills L (the variant SURVEY
    8(d) prescribes for roofline fractions).  Returns (ids, mask) on ``device`` and the lengths on the HOST - what a collator
    that pads on the CPU knows (COCO/data.py:135-144)."""
    rng = np.random.Generator(np.random.PCG64(1234 + rank))
    ids = rng.integers(1000, vocab, (n_seq, L))
    mu, sd = (76, 30) if L > 64 else (24, 8)
    lens = np.full(n_seq, L, np.int64) if dense else np.clip(np.rint(rng.normal(mu, sd, n_seq)), 8, L).astype(np.int64)
    mask = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    ids = ids * mask
    ids[:, 0] = 101
    ids[np.arange(n_seq), lens - 1] = 102
    return torch.from_numpy(ids).to(device), torch.from_numpy(mask).to(device), torch.from_numpy(lens)


def synth_batch(rank: int, n_seq: int, L: int, vocab: int, device, dense: bool = False):
    return synth_batch_lens(rank, n_seq, L, vocab, device, dense)[:2]


def attention_train_flops(cfg, extents) -> float:
    """SURVEY 8(d)'s attention term for the rows a step executes: forward 4 ext^2 H per sequence and layer (QK^T and PV), x3."""
    H, N = cfg.hidden_size, cfg.num_hidden_layers
    return 3.0 * N * 4.0 * H * float(np.sum(np.asarray(extents, np.float64) ** 2))


def train_flops_per_seq(cfg, L: int) -> float:
    """forward N*(24 H^2 + 4 L H) FLOP per token, training = 3x forward."""
    H, N = cfg.hidden_size, cfg.num_hidden_layers
    return 3.0 * L * N * (24.0 * H * H + 4.0 * L * H)


def train_gemm_flops_per_seq(cfg, L: int) -> float:
    """The GEMM class's share of it (QKV, attention output, FFN1, FFN2: 24 H^2 FLOP per token and layer, x3 for the step), on the
    L padded tokens of a sequence - SURVEY 8(d) counts padded tokens: the reference computes on padding."""
    H, N = cfg.hidden_size, cfg.num_hidden_layers
    return 3.0 * L * N * 24.0 * H * H


# ---------------------------------------------------------------------------------------------------------- CPU baselines
def _cpu_model_name() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores() -> int:
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_baseline_numpy(n_seq: int = 8, L: int = SEQ_LEN, steps: int = 2):
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
    return {"value": round(n_seq / dt, 3), "unit": "sequences/sec", "cores": int(cores),
            "sample": f"{steps} steps of {n_seq} sequences x L{L}, BERT-base, numpy fp32 oracle fwd+loss+bwd (no optimizer)"}


def _best_thread_count(n_seq: int, L: int) -> int:
    """a 8 x 64-token step does not scale to every core of a 2-socket box: take the fastest of a few thread counts"""
    phys = _physical_cores()
    cands = sorted({t for t in (8, 16, 32, 64, phys) if t <= phys})
    best, best_t = cands[-1], float("inf")
    for t in cands:
        r = cpu_baseline_torch(n_seq, L, warmup=1, steps=1 if n_seq * L > 2048 else 2, threads=t)
        if r["s_per_step"] < best_t:
            best, best_t = t, r["s_per_step"]
    return best


def cpu_baseline_torch(n_seq: int, L: int, warmup: int = 3, steps: int = 10, threads: int = 0):
    """The step the reference runs on a CPU: the encoder arithmetic of the reference IS transformers'
    BertModel (COCO/modeling.py:199, ANCE/model/models.py:226), here in fp32, eager attention, eval mode (COCO keeps the
    backbone in eval, :198), random-init BERT-base; on top of it the restated span-pair InfoNCE (COCO/modeling.py:244-248,
    oracle-checked) and torch AdamW with HF Trainer's clip_grad_norm_(1.0).  Median step time after warm-up."""
    from transformers import BertConfig, BertModel
    threads = threads or _physical_cores()
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(0)
        model = BertModel(BertConfig(attn_implementation="eager"), add_pooling_layer=False).eval()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01)
        ids, mask = synth_batch(0, n_seq, L, 30522, "cpu")
        target = torch.arange(n_seq).view(-1, 2).flip(1).flatten()  # co_target, COCO/modeling.py:172-177
        times = []
        for i in range(warmup + steps):
            t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            hs = model(input_ids=ids, attention_mask=mask, output_hidden_states=True, return_dict=True).hidden_states
            E = hs[-1][:, 0]
            S = E @ E.T
            S.fill_diagonal_(float("-inf"))
            loss = torch.nn.functional.cross_entropy(S, target, reduction="none").mean()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            if i >= warmup:
                times.append(time.perf_counter() - t0)
        med = float(np.median(times))
    finally:
        torch.set_num_threads(old)
    return {"value": round(n_seq / med, 3), "unit": "sequences/sec", "cores": int(threads), "s_per_step": round(med, 4),
            "sample": f"median of {steps} steps (after {warmup} warm-up) of {n_seq} sequences x L{L}: transformers BertModel (BERT-base, "
                      f"fp32, eager attention) fwd + span-pair InfoNCE + bwd + clip_grad_norm_(1.0) + torch AdamW"}


def cpu_baseline():
    """`value`: a bounded sample of the headline workload at its exact shape (configs[1]: 64 sequences x L 128, a few steps) on
    the host cores; `eight_sequences`: the same step on 8 sequences (more steps, the round-2/3 number).  `config1`: BASELINE.json configs[0] at its exact shape (8 sequences x L 64; BASELINE.md measured the reference's
    full coCondenser step at 6.4 sequences/s on 8 threads there - this scope has no Condenser head / MLM decoders).
    `port`: the numpy oracle."""
    out = {"kind": "port", "cpu": _cpu_model_name()}
    try:
        thr = _best_thread_count(SEQ_PER_GPU, SEQ_LEN)  # chosen ON the timed shape (64 x L128)
        # the headline's exact batch (64 sequences x L128) as a bounded sample: 1 warm-up + 5 timed steps (~25 s of CPU work)
        out.update(cpu_baseline_torch(SEQ_PER_GPU, SEQ_LEN, warmup=1, steps=5, threads=thr))
        out["eight_sequences"] = cpu_baseline_torch(8, SEQ_LEN, threads=thr)
        out["config1"] = cpu_baseline_torch(8, 64, threads=thr)
        out["threads_tried"] = "fastest of 8 / 16 / 32 / 64 / all physical cores on the timed shape (64 sequences x L128, 1 warm-up + 1 step each)"
        out["port"] = cpu_baseline_numpy()
    except Exception as e:  # transformers missing on the box: the numpy port alone
        out.update(cpu_baseline_numpy())
        out["note"] = f"torch-CPU HuggingFace baseline unavailable ({type(e).__name__}: {e}); numpy oracle only"
    return out


# ---------------------------------------------------------------------------------------------------------- side legs
def full_coco_step(cfg, dev, ids, mask, lens=None, steps: int = 8, warmup: int = 3, padded_too: bool = True):
    """The reference's whole pre-training step (COCO/modeling.py:192-235 with COCO/README.md:49 settings: 2 Condenser
    head layers, skip_from 6, late MLM): backbone + head + two label-sparse MLM losses + contrastive + AdamW.
    Reported next to the headline metric, never instead of it."""
    import types
    from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertModel
    from cocodr_amd.optim import FlatAdamW, clip_grad_norm_
    torch.manual_seed(0)
    bert = CocoBertModel(cfg).to(dev)
    margs = types.SimpleNamespace(n_head_layers=2, skip_from=min(6, cfg.num_hidden_layers), late_mlm=True)
    model = CoCondenserForPretraining(bert, margs).to(dev)
    opt = FlatAdamW.for_model(model, lr=1e-4, weight_decay=0.01)  # backbone + head flats, all four shadows kept in the pass
    g = torch.Generator().manual_seed(5)
    pick = (torch.rand(ids.shape, generator=g) < 0.15).to(dev) & (mask > 0)
    pick[:, 0] = False
    labels = torch.where(pick, ids, torch.full_like(ids, -100))
    inp = torch.where(pick, torch.full_like(ids, 103), ids)  # [MASK]
    batch = {"input_ids": inp, "attention_mask": mask}
    if lens is not None:
        batch["lengths"] = lens  # host-known lengths: the packed layout (the default execution) is built without a read-back
    all_flats = [bert.flat_decay, bert.flat_nodecay, model.c_head.flat_decay, model.c_head.flat_nodecay]

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model(batch, labels)
        loss.backward()
        opt.step(clip=clip_grad_norm_(all_flats, 1.0))  # HF Trainer default max_grad_norm, over backbone + head, on the device
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pdt = float("nan")
    if padded_too:
        bert.pack_sequences = False  # the same step on the padded layout (all B x L rows)
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        pdt = (time.perf_counter() - t0) / steps
    return {"sequences_per_sec": round(ids.shape[0] / dt, 1), "ms_per_step": round(dt * 1e3, 3), "loss": round(float(loss.detach()), 3),
            "execution": "packed (backbone and Condenser head on the stored rows)", "padded_ms_per_step": None if pdt != pdt else round(pdt * 1e3, 3),
            "scope": "backbone + 2 Condenser head layers (skip_from 6) + head & late MLM losses (label-sparse, 15 %) + contrastive + clip_grad_norm_(1.0) + AdamW"}


def eval_search(dev, nq: int = 10000, npass: int = 125000, dim: int = 1024, k: int = 1000, iters: int = 5):
    """BASELINE.json's second metric on one GPU's shard of config 5 at its real size (cocodr-large width, 10 000 queries x
    125 000 passages per GPU, k = 1000): query x passage dot-products/sec = Nq*Np / wall time of (scores + exact top-k),
    embeddings resident in HBM.  Default pipeline: split-precision scores (two IEEE halves per operand, three partial
    products = 3 executed half-precision MFMA FLOPs per algorithmic FLOP, fp32 accumulation; at least as accurate as an fp32
    dot product, include/cocodr.h) - roofline against the dense 16-bit MFMA peak on the EXECUTED FLOPs.  The exact
    fp32-MFMA pipeline (mode 1) is timed next to it against the fp32-MFMA peak.  The GEMM time is bracketed with HIP events on
    the launch stream in a separate pass."""
    from cocodr_amd import ops
    g = torch.Generator().manual_seed(7)
    Q = (torch.randn(nq, dim, generator=g) / dim ** 0.5).to(dev)
    P = (torch.randn(npass, dim, generator=g) / dim ** 0.5).to(dev)
    ws = torch.empty(ops.lib().cocodr_score_topk_workspace_bytes_dim(nq, npass, dim, k), dtype=torch.uint8, device=dev)

    def timed(mode: int):
        ops.score_set_mode(mode)
        ops.score_topk(Q, P, k, workspace=ws)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ops.score_topk(Q, P, k, workspace=ws)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        ops.prof_begin(3)
        ops.score_topk(Q, P, k, workspace=ws)
        torch.cuda.synchronize()
        n_launch, ms, flops = ops.prof_end()
        return dt, n_launch, ms, flops

    try:
        dt, n_launch, ms, flops = timed(0)
        dte, ne, mse, flopse = timed(1)
        dth, _, _, _ = timed(2)
        os.environ["COCODR_SCORE_NOFILTER"] = "1"  # the exhaustive route (score slab + radix select), for comparison
        dtx, _, _, _ = timed(0)
    finally:
        os.environ.pop("COCODR_SCORE_NOFILTER", None)
        ops.score_set_mode(0)
    # the reference's use of the index - add(P) once, search several times (ANCE/drivers/run_ann_data_gen.py:310-317,390) - through
    # retrieval.FlatIPIndex: the passages' split image and filter sample stay resident, a search rebuilds the query side only
    from cocodr_amd import retrieval
    index = retrieval.FlatIPIndex(dim)
    index.add(P)
    index.search(Q, k)
    index.search(Q, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        index.search(Q, k)
    torch.cuda.synchronize()
    dti = (time.perf_counter() - t0) / iters
    del index
    plan = ops.score_filter_plan(nq, npass, dim, k)
    handed_back = None
    if plan["filtered"]:
        ops.score_topk(Q, P, k, workspace=ws)
        torch.cuda.synchronize()
        o = plan["handed_back_count_offset"]
        handed_back = int(ws[o:o + 4].view(torch.int32).item())
    alg = 2.0 * nq * npass * dim
    dimp = (dim + 63) // 64 * 64
    executed = 3.0 * 2.0 * nq * npass * dimp  # ql.ph + qh.pl + qh.ph
    out = {"dot_products_per_sec": round(nq * npass / dt), "ms": round(dt * 1e3, 2),
           "workload": f"{nq} queries x {npass} passages x {dim} fp32, k={k}, split-precision scores (3 half-precision MFMA products per "
                       f"score, fp32 accumulate) + exact top-k, one GPU's shard of config 5",
           "algorithmic_tflops": round(alg / dt / 1e12, 1)}
    out["resident_index"] = {"dot_products_per_sec": round(nq * npass / dti), "ms": round(dti * 1e3, 2),
                             "note": "the same search through retrieval.FlatIPIndex (cocodr_score_topk_resident): add(P) once, every later "
                                     "search of the same shape reuses the passages' scale, split image and filter sample - the reference's "
                                     "IndexFlatIP usage (run_ann_data_gen.py:310-317,390); identical D / I (tests/test_gpu_search_filter.py)"}
    out["selection"] = {"filtered": bool(plan["filtered"]), "plan": plan, "rows_handed_back_to_the_exhaustive_pass": handed_back,
                        "exhaustive_route_ms": round(dtx * 1e3, 2), "exhaustive_route_dot_products_per_sec": round(nq * npass / dtx),
                        "note": "filtered search (include/cocodr.h): per-row thresholds from a strided passage sample, the score GEMM's "
                                "epilogue keeps the scores at or above them, the k best are selected from those candidates; identical "
                                "D / I to the exhaustive route (COCODR_SCORE_NOFILTER=1: fp32 score slab + radix select), "
                                "tests/test_gpu_search_filter.py"}
    out["roofline"] = {"bound": "mfma", "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "dtype": "f16 operands, f32 accumulate",
                       "achieved": round(executed / dt / 1e12, 1), "frac": round(executed / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                       "algorithmic_achieved": round(alg / dt / 1e12, 1), "algorithmic_frac": round(alg / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                       "note": "whole search (operand split + sample and score GEMMs + selection) / wall time against the dense 16-bit MFMA peak: "
                               "`achieved` / `frac` count the EXECUTED half-precision MFMA FLOPs (3 per algorithmic FLOP: the price of fp32 "
                               "accuracy on the 16-bit pipe), `algorithmic_*` the 2 Nq Np H of the metric; the reference's own arithmetic "
                               "(fp32) is the exact_fp32_mfma_pipeline block, against the fp32-MFMA peak"}
    if n_launch and ms > 0:
        ach = executed / (ms * 1e-3) / 1e12
        out["roofline"].update({"score_kernel_achieved": round(ach, 1), "score_kernel_frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                                "score_kernel_launches": n_launch, "score_kernel_share_of_search": round(ms / (dt * 1e3), 3)})
    exact = {"dot_products_per_sec": round(nq * npass / dte), "ms": round(dte * 1e3, 2),
             "roofline": {"bound": "mfma", "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "dtype": "f32",
                          "achieved": round(alg / dte / 1e12, 2), "frac": round(alg / dte / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}}
    if ne and mse > 0:
        exact["roofline"].update({"score_kernel_achieved": round(flopse / (mse * 1e-3) / 1e12, 2),
                                  "score_kernel_frac": round(flopse / (mse * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                  "score_kernel_share_of_search": round(mse / (dte * 1e3), 3)})
    out["exact_fp32_mfma_pipeline"] = exact
    out["half_precision_scores_opt_in"] = {
        "dot_products_per_sec": round(nq * npass / dth), "ms": round(dth * 1e3, 2),
        "roofline": {"bound": "mfma", "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "dtype": "f16 operands, f32 accumulate",
                     "achieved": round(2.0 * nq * npass * dimp / dth / 1e12, 1), "frac": round(2.0 * nq * npass * dimp / dth / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)},
        "note": "cocodr_score_set_mode(2), NOT the default and not the number of record: one product of the operands rounded to IEEE half "
                "(a faiss fp16 flat index's arithmetic); score error ~1e-5 |q||p|, nDCG@10 within 1e-3 of the exact search "
                "(tests/test_gpu_retrieval.py::test_half_precision_score_mode_is_within_the_stated_tolerances)"}
    return out


def config5_end_to_end(dev, n_pass: int = 1_000_000, nq: int = 10_000, world: int = 8, k: int = 1000, batch: int = 1024, keep: bool = False):
    """BASELINE configs[4] whole, on ONE GPU (the 8-way configuration walked shard by shard - VERDICT r03 item 8): cocodr-large
    encodes `n_pass` synthetic passages (L 128) sharded `world` ways by the reference's rule (record i -> shard i % W,
    ANCE/utils/util.py:390-392) and `nq` queries (L 64), embeddings stay in HBM (ANCE/drivers/run_ann_data_gen.py:157-212); then every
    shard is searched (k = 1000) and the per-shard lists are merged natively into what ONE IndexFlatIP search over the rank-major
    merged corpus returns (evaluate/evaluation/evaluate_beir.py:200-224).  ``keep``: also return (Q, P in merged order, D, I) for
    the parity test (tests/test_gpu_retrieval.py)."""
    from cocodr_amd import retrieval
    from cocodr_amd.modeling import BertDotNLL, CocoBertConfig
    cfg = CocoBertConfig.large()
    torch.manual_seed(0)
    model = BertDotNLL(cfg).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1234)

    def tokens(n, L, mu, sd):  # SURVEY 8(d) synthetic inputs, generated in HBM (the token cache of a corpus this size lives there)
        lens = torch.clamp(torch.round(torch.randn(n, generator=g, device=dev) * sd + mu), 8, L).to(torch.int64)
        ids = torch.randint(1000, cfg.vocab_size, (n, L), generator=g, device=dev, dtype=torch.int32)
        pos = torch.arange(L, device=dev)[None]
        ids = torch.where(pos < lens[:, None], ids, torch.zeros_like(ids))
        ids[:, 0] = 101
        ids.scatter_(1, (lens - 1)[:, None], torch.full((n, 1), 102, dtype=torch.int32, device=dev))
        return ids, lens.cpu()

    t_all = time.perf_counter()
    shards, enc_s = [], 0.0
    for r in range(world):
        n_r = len(range(r, n_pass, world))
        ids, lens = tokens(n_r, SEQ_LEN, 76, 30)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        emb, _ = retrieval.encode_corpus(model, ids, None, batch_size=batch, lengths=lens)
        torch.cuda.synchronize()
        enc_s += time.perf_counter() - t0
        shards.append(emb)
        del ids
    qids, qlens = tokens(nq, 64, 24, 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Q, _ = retrieval.encode_corpus(model, qids, None, batch_size=batch, is_query=True, lengths=qlens)
    torch.cuda.synchronize()
    q_s = time.perf_counter() - t0
    del model
    t0 = time.perf_counter()
    Ds, Is = [], []
    for r in range(world):
        D, I = retrieval.search(Q, shards[r], k)
        Ds.append(D)
        Is.append(I.to(torch.int32))
    offs = torch.tensor([sum(s_.shape[0] for s_ in shards[:w]) for w in range(world)], dtype=torch.int64, device=dev)
    Dm, Im = retrieval.merge_shard_lists(torch.stack(Ds), torch.stack(Is), offs, k)
    torch.cuda.synchronize()
    s_s = time.perf_counter() - t0
    out = {"passages": n_pass, "queries": nq, "shards": world, "k": k,
           "encode_passages_per_sec": round(n_pass / enc_s, 1), "encode_s": round(enc_s, 2), "encode_queries_per_sec": round(nq / q_s, 1),
           "search_dot_products_per_sec": round(nq * n_pass / s_s), "search_ms": round(s_s * 1e3, 2), "wall_s": round(time.perf_counter() - t_all, 2),
           "workload": f"cocodr-large: {n_pass} passages x L{SEQ_LEN} in {world} shards (i % W) + {nq} queries x L64 encoded (packed batches of "
                       f"{batch}, host-known lengths), per-shard split-precision search k = {k} + native {world}-way merge; one GPU walks the "
                       "shards one after the other; BASELINE configs[4]"}
    if keep:
        return out, Q, torch.cat(shards), Dm, Im
    return out


def search_cpu_baseline(nq: int = 1000, npass: int = 125000, dim: int = 1024, k: int = 1000):
    """The reference's search on the host cores (evaluate/evaluation/evaluate_beir.py:220-224 runs faiss IndexFlatIP on the CPU;
    faiss is not in this image - SURVEY 8c - so the stand-in is what it computes: fp32 Q P^T through the CPU BLAS + top-k), on a
    bounded slice of one config-5 shard: `nq` of the 10 000 queries x 125 000 passages."""
    g = torch.Generator().manual_seed(7)
    Q = torch.randn(nq, dim, generator=g) / dim ** 0.5
    P = torch.randn(npass, dim, generator=g) / dim ** 0.5
    threads = _physical_cores()
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.topk(Q[:64] @ P.T, k, dim=1)
        t0 = time.perf_counter()
        D, I = torch.topk(Q @ P.T, k, dim=1)
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(old)
    return {"value": round(nq * npass / dt), "unit": "dot-products/sec", "cores": int(threads), "kind": "port", "cpu": _cpu_model_name(),
            "sample": f"{nq} queries x {npass} passages x {dim} fp32 (a tenth of one config-5 shard's queries), k = {k}: torch fp32 Q @ P.T on the "
                      "CPU BLAS + torch.topk - the arithmetic of faiss IndexFlatIP.search, which the reference calls and this image lacks"}


def ance_step(dev, rows: int = 32, steps: int = 10, warmup: int = 3, rank: int = 0, world: int = 1, fence=None, dp_chunks: int = 2,
              extras: bool = True, host_lengths: bool = False):
    """BASELINE config 4 (ANCE/drivers/run_ann.py:293-356): BERT-large triplet step, 32 rows/GPU = queries [32,64] +
    positives / negatives [32,128], backward, clip_grad_norm_(1.0), LAMB (the reference's default optimizer), linear
    schedule.  One training row = 3 sequences.  ``world > 1``: the data-parallel step - every rank its own rows
    (the reference strides them by rank, ANCE/utils/util.py:390-392), the summed gradient of the query and passage passes
    averaged ONCE over RCCL (DDP in the reference, run_ann.py:177-184), timed between fences, whole-job rows per second."""
    from cocodr_amd.modeling import BertDotNLL, CocoBertConfig
    from cocodr_amd.optim import FlatLamb, clip_grad_norm_
    cfg = CocoBertConfig.large()
    torch.manual_seed(0)
    model = BertDotNLL(cfg).to(dev)
    if world > 1:
        model.bert.enable_grad_allreduce(chunks=dp_chunks)
    opt = FlatLamb.for_model(model.bert, lr=5e-6, eps=1e-8)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: max(0.0, 1.0 - s / 1000.0))
    q, qm, ql = synth_batch_lens(3 * rank, rows, 64, cfg.vocab_size, dev)
    a, am, al = synth_batch_lens(3 * rank + 1, rows, 128, cfg.vocab_size, dev)
    b, bm, bl = synth_batch_lens(3 * rank + 2, rows, 128, cfg.vocab_size, dev)
    flats = [model.bert.flat_decay, model.bert.flat_nodecay]

    def step():
        opt.zero_grad(set_to_none=True)
        # the reference's 6 tensors (ANCE/data/msmarco_data.py:381-382 after .to(device)); host_lengths: + what the CPU-side data function knows
        loss, _acc, _logits = model(q, qm, a, am, b, bm, lengths=(ql, al, bl) if host_lengths else None)
        loss.backward()
        opt.step(clip=clip_grad_norm_(flats, 1.0))  # norm and coefficient stay on the device
        sched.step()
        return loss

    for _ in range(warmup):
        loss = step()
    if world > 1:
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        fence()
        return (time.perf_counter() - t0) / steps, float(loss.detach())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"sequences_per_sec": round(3 * rows / dt, 1), "rows_per_sec": round(rows / dt, 1), "ms_per_step": round(dt * 1e3, 3),
           "loss": round(float(loss.detach()), 4),
           "execution": "BertDotNLL defaults: queries + positives + negatives as ONE packed encoder pass (merge_passes + pack_sequences), "
                        "dropout on (model.train(), ANCE/drivers/run_ann.py:293)",
           "scope": f"cocodr-large triplet step, {rows} rows (q L64 + pos/neg L128), bf16, clip_grad_norm_(1.0) + LAMB; BASELINE configs[3]"}
    # roofline of the step's dominant kernel class (bf16 MFMA GEMMs), HIP events around every GEMM launch of three more steps
    from cocodr_amd import ops
    ops.prof_begin(1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n_launch, gemm_ms, gemm_flops = ops.prof_end()
    if n_launch and gemm_ms > 0:
        ach = gemm_flops / (gemm_ms * 1e-3) / 1e12
        H_, N_ = cfg.hidden_size, cfg.num_hidden_layers
        alg = 3.0 * 24.0 * H_ * H_ * N_ * (rows * 64 + 2 * rows * 128) * 3 / (gemm_ms * 1e-3) / 1e12  # padded tokens of q + pos + neg, 3 steps
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                           "algorithmic_achieved": round(alg, 1), "algorithmic_frac": round(alg / MFMA_BF16_PEAK_TFLOPS, 4),
                           "launches_per_step": n_launch // 3, "avg_launch_us": round(gemm_ms * 1e3 / n_launch, 2),
                           "gemm_share_of_step": round(gemm_ms / 3 / (dt * 1e3), 3),
                           "flops": "achieved / frac: FLOPs the GEMM launches of the step execute (stored rows of the packed pass) / their "
                                    "summed duration; algorithmic_*: 24 H^2 x 3 per PADDED token of the three encoder inputs over the same time"}
    if not extras:  # (tools/ance_profile.py: the default step alone under the profiler)
        return out
    # the same step as the reference lays it out: a padded query pass (side stream) next to a padded passage pass
    model.bert.pack_sequences = model.merge_passes = False
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    mdt = (time.perf_counter() - t0) / steps
    out["padded_two_passes"] = {"rows_per_sec": round(rows / mdt, 1), "ms_per_step": round(mdt * 1e3, 3),
                                "note": "merge_passes = pack_sequences = False: padded query pass on a side stream + padded passage pass"}
    model.bert.pack_sequences = model.merge_passes = True
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
    retrieval.encode_corpus(model, ids[:batch], mask[:batch], batch_size=batch, pack=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        emb, _ = retrieval.encode_corpus(model, ids, mask, batch_size=batch, pack=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    retrieval.encode_corpus(model, ids[:batch], mask[:batch], batch_size=batch, pack=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        emb_p, _ = retrieval.encode_corpus(model, ids, mask, batch_size=batch, pack=True)
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t0) / iters
    return {"sequences_per_sec": round(n / dt, 1), "ms": round(dt * 1e3, 2),
            "packed_sequences_per_sec": round(n / dtp, 1), "packed_equals_padded": bool(torch.equal(emb_p, emb)),
            "workload": f"{n} passages x L{seq_len}, batch {batch}, BertDot_NLL_LN body_emb (last-layer [CLS]), bf16 encoder, eval mode; "
                        "packed = the same batches stored back to back (every sequence on its own length), same embeddings"}


def multi_gpu_legs(dev, rank: int, world: int, fence, tmax, shared: bool, dp_chunks: int):
    """What BASELINE configs[3] / [4] name beyond the contrastive step, on all ranks of the job:
      * sharded corpus encode - every rank encodes its shard (record i -> rank i % W, ANCE/utils/util.py:390-392;
        ANCE/drivers/run_ann_data_gen.py:157-212), embeddings stay in HBM;
      * sharded search - 10 000 queries x (125 000 x W) passages x 1024, k = 1000: query all-gather, per-shard score + top-k,
        query-block exchange, native k-way merge, gather of the merged blocks ALL inside the timed region
        (evaluate/evaluation/evaluate_beir.py:220-224 on the rank-major merged corpus);
      * the data-parallel ANCE triplet step (ANCE/drivers/run_ann.py:293-356), BERT-large, 32 rows per GPU.
    Every number is whole-job work / max-over-ranks time between fences.  ``shared`` (ranks share a GPU over gloo: a code-path
    check) runs reduced sizes."""
    from cocodr_amd import retrieval
    from cocodr_amd.modeling import BertDotNLL, CocoBertConfig
    out = {"note": "reduced sizes: the ranks share GPUs over gloo (code-path check, not a scaling number)" if shared else
                   "whole-job work / max-over-ranks time between barrier + synchronize fences"}
    # ---- sharded corpus encode (cocodr-large, the config-5 encoder)
    cfg = CocoBertConfig.large()
    torch.manual_seed(0)
    enc = BertDotNLL(cfg).to(dev).eval()
    n_local, batch = (1024, 256) if shared else (8192, 512)
    ids, mask = synth_batch(100 + rank, n_local, SEQ_LEN, cfg.vocab_size, dev)
    retrieval.encode_corpus(enc, ids[:batch], mask[:batch], batch_size=batch)
    fence()
    t0 = time.perf_counter()
    emb, _ = retrieval.encode_corpus(enc, ids, mask, batch_size=batch)
    fence()
    dt = tmax(time.perf_counter() - t0)
    out["sharded_corpus_encode"] = {"sequences_per_sec": round(n_local * world / dt, 1), "ms": round(dt * 1e3, 2),
                                    "workload": f"cocodr-large body_emb, {n_local} passages x L{SEQ_LEN} per rank (record i -> rank i % W), batch {batch}, "
                                                "packed execution of the padded batches (layout planned on the device from the masks), embeddings "
                                                "kept in HBM; BASELINE configs[4] encode half"}
    del enc, ids, mask, emb
    torch.cuda.empty_cache()
    # ---- sharded search
    nq, np_local, dim, k = (2000, 20000, 1024, 100) if shared else (10000, 125000, 1024, 1000)
    g = torch.Generator().manual_seed(7 + rank)
    nq_local = len(range(rank, nq, world))
    Ql = (torch.randn(nq_local, dim, generator=g) / dim ** 0.5).to(dev)
    Pl = (torch.randn(np_local, dim, generator=g) / dim ** 0.5).to(dev)
    retrieval.sharded_search(Ql, Pl, k)
    iters = 3
    fence()
    t0 = time.perf_counter()
    for _ in range(iters):
        D, I = retrieval.sharded_search(Ql, Pl, k)
    fence()
    dt = tmax((time.perf_counter() - t0) / iters)
    out["sharded_search"] = {"dot_products_per_sec": round(nq * np_local * world / dt), "ms": round(dt * 1e3, 2),
                             "workload": f"{nq} queries x ({np_local} x {world}) passages x {dim} fp32, k = {k}: query all-gather + per-shard "
                                         "split-precision scores + exact top-k + query-block exchange (fp32 scores, int32 shard-local "
                                         "positions) + native k-way merge + gather of the merged blocks, all timed; BASELINE configs[4] search half",
                             "result_rows": int(D.shape[0])}
    del Ql, Pl, D, I
    torch.cuda.empty_cache()
    # ---- data-parallel ANCE step
    rows = 32
    adt, aloss = ance_step(dev, rows=rows, steps=3 if shared else 8, warmup=2 if shared else 3, rank=rank, world=world, fence=fence,
                           dp_chunks=dp_chunks)
    adt = tmax(adt)
    out["ance_triplet_step"] = {"rows_per_sec": round(rows * world / adt, 1), "sequences_per_sec": round(3 * rows * world / adt, 1),
                                "ms_per_step": round(adt * 1e3, 3), "loss": round(aloss, 4),
                                "scope": f"cocodr-large triplet step, {rows} rows per GPU (q L64 + pos/neg L128), summed gradient of the two "
                                         "passes averaged once over the ranks, clip_grad_norm_(1.0) + LAMB; BASELINE configs[3]"}
    return out


# ---------------------------------------------------------------------------------------------------------- the timed step
def _traffic_for(model_name: str, seq_per_gpu: int, seq_len: int, packed: bool = False):
    """HBM-side bytes per GEMM launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs,
    corrected as MI355X_MICROARCH.md prescribes) of THIS workload shape and execution (packed / padded), with the file it came
    from; None when no pass was taken on it."""
    path = os.path.join(ROOT, "profiles", f"gemm_pmc_{model_name}_{seq_per_gpu}x{seq_len}{'_packed' if packed else ''}.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return round(d["hbm_bytes_per_launch"]), {"file": os.path.relpath(path, ROOT), "commit": d.get("commit"), "launches": d.get("launches")}
    except Exception:
        return None, None


def contrastive_leg(model_name: str, seq_per_gpu: int, seq_len: int, steps: int, warmup: int, dev, rank: int, world: int, use_dist: bool,
                    dp_chunks: int, roofline: bool, dense: bool = False, packed: bool = False, host_lengths: bool = False):
    """Build the model, run `warmup` untimed + exactly `steps` timed contrastive steps between two fences; returns
    (seconds over the timed steps on this rank, final loss, roofline dict or None, cfg, one batch, what the steps executed)."""
    import torch.distributed as dist
    from cocodr_amd import ops
    from cocodr_amd.modeling import CoCondenserForPretraining, CocoBertConfig, CocoBertModel
    from cocodr_amd.optim import FlatAdamW, clip_grad_norm_
    cfg = CocoBertConfig.base() if model_name == "base" else CocoBertConfig.large()
    torch.manual_seed(0)  # identical random-init weights on every rank
    bert = CocoBertModel(cfg).to(dev)
    model = CoCondenserForPretraining(bert)
    if use_dist:
        bert.enable_grad_allreduce(chunks=dp_chunks)  # averaged inside the backward, overlapped with it
    opt = FlatAdamW.for_model(bert, lr=1e-4, weight_decay=0.01)  # torch.optim.AdamW semantics, one native pass per flat
    total = steps + warmup
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: min(1.0, (s + 1) / max(1, int(0.1 * total))))
    # a small pool of different synthetic batches, resident in HBM before the timed region, visited round-robin (one
    # repeated batch is memorised within a few steps and the loss saturates at 0).  A batch is what the reference's collator
    # hands the model - padded ids + attention mask (COCO/data.py:150-154), nothing else (``host_lengths``: plus the B lengths
    # on the host, which a collator that pads on the CPU knows - a boundary extension, measured as a side leg); NOTHING of the
    # packed layout is prebuilt: every timed step builds its own inside model(batch) (coco-dr_amd/modeling.py PackedIndex:
    # planned on the device from the mask, 16 bytes read back; with host lengths one pinned copy of 3B+1 integers)
    pool = [synth_batch_lens(rank + 10007 * i, seq_per_gpu, seq_len, cfg.vocab_size, dev, dense) for i in range(8)]
    bert.pack_sequences = bool(packed)  # (True is the model's default)

    def fresh_batch(i):  # a new dict per step: nothing a previous step attached can survive
        ids_, mask_, lens_ = pool[i % len(pool)]
        b_ = {"input_ids": ids_, "attention_mask": mask_}
        if host_lengths:
            b_["lengths"] = lens_
        return b_

    step_no = [0]
    flats = [bert.flat_decay, bert.flat_nodecay]
    seen, last = [], [None]  # per step: (a dict the previous step did not see, without a prebuilt layout); tests/test_gpu_bench_contract.py

    def step():
        batch = fresh_batch(step_no[0])
        seen.append(batch is not last[0] and "packed_index" not in batch)
        last[0] = batch
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

    loss = None
    for _ in range(warmup):
        loss = step()
    fence()
    prof = roofline and rank == 0
    if prof:
        ops.prof_begin(1)  # HIP events around every GEMM launch, on the launch stream
    # the event pairs cost stream time (~7 % of the step when every GEMM launch of every step carries one), so the
    # roofline leg brackets the GEMM launches of every PROF_EVERY-th timed step only
    t0 = time.perf_counter()
    for i in range(steps):
        if prof:
            ops.prof_pause(i % PROF_EVERY != 0)
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    roof = None
    # rows and attention extents the timed steps executed (host arithmetic on the lengths the batches carry)
    def extents_of(i):  # rows every sequence of the step's batch is stored on (cocodr_amd.modeling.packed_extents)
        if not packed:
            return np.full(seq_per_gpu, seq_len)
        from cocodr_amd.modeling import packed_extents
        lens_ = pool[(warmup + i) % len(pool)][2].numpy()
        host, _, _ = packed_extents(lens_, seq_per_gpu, seq_len)
        return np.diff(host[seq_per_gpu:].astype(np.int64))
    rows_per_step = float(np.mean([extents_of(i).sum() for i in range(steps)]))
    # (the attention kernels walk an extent in 32-row blocks: their executed FLOPs are those of the extents rounded up to 32)
    attn_flops_per_step = float(np.mean([attention_train_flops(cfg, (extents_of(i) + 31) // 32 * 32) for i in range(steps)]))
    exec_info = {"rows_per_step": int(round(rows_per_step)), "rows_per_step_padded": seq_per_gpu * seq_len}
    if prof:
        n_launch, gemm_ms_raw, gemm_flops = ops.prof_end()
        # event-to-event time -> kernel time: one event pair adds a fixed ~2-3 us to the launch it brackets (measured on this box,
        # on this stream, around an empty kernel); rocprofv3's per-kernel durations of the same command (profiles/) are the check
        ev_us = ops.prof_event_overhead_us() if n_launch else 0.0
        gemm_ms = gemm_ms_raw - n_launch * ev_us * 1e-3
        if n_launch and gemm_ms > 0:
            ach = gemm_flops / (gemm_ms * 1e-3) / 1e12  # FLOPs the bracketed launches carried out / their summed kernel time
            sampled = len(range(0, steps, PROF_EVERY))
            traffic, src = _traffic_for(model_name, seq_per_gpu, seq_len, packed)
            lps = n_launch // max(1, sampled)
            gemm_ms_step = gemm_ms / sampled
            alg = train_gemm_flops_per_seq(cfg, seq_len) * seq_per_gpu * sampled / (gemm_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_unit": "HBM-side bytes per GEMM launch (FETCH_SIZE + WRITE_SIZE PMC passes, average over the class)",
                    "traffic_per_step_bytes": None if traffic is None else int(traffic * lps),
                    "traffic_gbps": None if traffic is None else round(traffic * lps / (gemm_ms_step * 1e-3) / 1e9, 1),
                    "traffic_source": src,
                    "algorithmic_achieved": round(alg, 1), "algorithmic_frac": round(alg / MFMA_BF16_PEAK_TFLOPS, 4),
                    "kernel": "bf16 MFMA GEMM class of coco-dr_amd/csrc/gemm.hip + gemm_pp.hip + gemm_a4.hip (all NT / NN / TN launches of the step)",
                    "launches_per_step": lps,
                    "sampled": f"every GEMM launch of every {PROF_EVERY}th timed step ({n_launch} launches)",
                    "avg_launch_us": round(gemm_ms * 1e3 / n_launch, 2),
                    "event_overhead_us": round(ev_us, 2), "avg_launch_us_event_to_event": round(gemm_ms_raw * 1e3 / n_launch, 2),
                    "gemm_share_of_step": round(gemm_ms_step / (dt / steps * 1e3), 3),
                    "executed_gemm_flops_per_step": gemm_flops / sampled,
                    "flops": "achieved / frac: the FLOPs the timed launches EXECUTE (2 M N K of every launch: "
                             + ("stored rows only, " if packed else "all B x L rows, ") +
                             "the last layer's output projection and FFN on the [CLS] rows alone - cocodr_config.cls_tail) / the summed "
                             "duration of those launches.  algorithmic_*: the class's FLOPs on the padded batch as the reference computes "
                             "it (24 H^2 per padded token and layer, x3: SURVEY 8d) over the same time - a throughput figure, not a "
                             "utilisation",
                    "batches": "fully dense (every sequence fills L)" if dense else
                               ("MS MARCO-shaped lengths, stored back to back (every sequence on its own length): no work on padding rows, same loss and "
                                "gradients as the padded execution (tests/test_gpu_packed.py)" if packed else
                                "MS MARCO-shaped lengths, padded to L; every kernel runs over all B x L rows")}
            roof.update(exec_info)
            exec_info["executed_flops_per_step"] = gemm_flops / sampled + attn_flops_per_step
    exec_info["fresh_batches"] = bool(seen and all(seen))
    del opt, model, bert
    torch.cuda.empty_cache()
    return dt, float(loss.detach()), roof, cfg, pool[0], exec_info


def whole_step_fracs(n_seq: int, steps: int, dt: float, cfg, seq_len: int, exec_info: dict, world: int = 1) -> dict:
    """Whole-step rates against the MFMA peak.  executed_*: the FLOPs the step's launches carry out (GEMM class as bracketed by
    the HIP events + the attention term on the executed extents) / step time - the utilisation figure.  algorithmic_*: SURVEY
    8(d)'s count on the padded batch (the reference's arithmetic) / step time - a throughput figure."""
    v = n_seq * steps / dt
    alg = v * train_flops_per_seq(cfg, seq_len) / 1e12
    d = {"algorithmic_tflops_whole_step": round(alg, 1), "algorithmic_whole_step_frac": round(alg / (MFMA_BF16_PEAK_TFLOPS * world), 4)}
    if exec_info.get("executed_flops_per_step"):
        ex = exec_info["executed_flops_per_step"] / (dt / steps) / 1e12
        d["executed_tflops_whole_step"] = round(ex, 1)
        d["executed_whole_step_frac"] = round(ex / MFMA_BF16_PEAK_TFLOPS, 4)
    d["rows_per_step"] = exec_info.get("rows_per_step")
    d["fresh_batches_every_step"] = exec_info.get("fresh_batches")
    return d


def compact_roofline(roof: dict) -> dict:
    """The contract line's roofline block: numbers plus a short kernel name; the prose of the full block stays in bench_legs.json."""
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_per_step_bytes", "traffic_gbps", "algorithmic_achieved",
            "algorithmic_frac", "launches_per_step", "avg_launch_us", "event_overhead_us", "gemm_share_of_step", "rows_per_step", "rows_per_step_padded")
    out = {k: roof[k] for k in keep if k in roof}
    src = roof.get("traffic_source")
    if isinstance(src, dict):  # the traffic figure is NOT taken in this run: file + commit of the PMC passes it was read from
        out["traffic_source"] = {k: src.get(k) for k in ("file", "commit")}
    out["kernel"] = "bf16 MFMA GEMM class (gemm.hip + gemm_pp.hip + gemm_a4.hip), all launches of the step"
    return out


def leg_summary(extras: dict) -> dict:
    """One number per side leg for the contract line."""
    s = {}
    g = lambda d, *ks: (g(d.get(ks[0], {}), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None  # noqa: E731
    ns = extras.get("north_star_large_step", {})
    s["large_256_padded_gemm_frac"] = g(ns, "256_sequences_padded", "roofline", "frac")
    s["large_256_padded_step_frac"] = g(ns, "256_sequences_padded", "executed_whole_step_frac")
    s["large_256_padded_seq_per_sec"] = g(ns, "256_sequences_padded", "sequences_per_sec")
    s["large_256_packed_seq_per_sec"] = g(ns, "256_sequences", "sequences_per_sec")
    s["host_lengths_seq_per_sec"] = g(extras, "host_lengths_contrastive_step", "sequences_per_sec")
    s["reference_batch_seq_per_sec"] = g(extras, "reference_batch_contrastive_step", "sequences_per_sec")
    s["padded_seq_per_sec"] = g(extras, "padded_contrastive_step", "sequences_per_sec")
    s["padded_gemm_frac"] = g(extras, "padded_contrastive_step", "roofline", "frac")
    s["full_coco_seq_per_sec"] = g(extras, "full_coco_step", "sequences_per_sec")
    s["ance_rows_per_sec"] = g(extras, "ance_triplet_step", "rows_per_sec")
    s["ance_gemm_frac"] = g(extras, "ance_triplet_step", "roofline", "frac")
    s["corpus_encode_packed_seq_per_sec"] = g(extras, "corpus_encode", "packed_sequences_per_sec")
    s["search_dot_products_per_sec"] = g(extras, "eval_search", "dot_products_per_sec")
    s["search_resident_index_dot_products_per_sec"] = g(extras, "eval_search", "resident_index", "dot_products_per_sec")
    s["search_half_precision_opt_in_dot_products_per_sec"] = g(extras, "eval_search", "half_precision_scores_opt_in", "dot_products_per_sec")
    s["search_exhaustive_route_dot_products_per_sec"] = g(extras, "eval_search", "selection", "exhaustive_route_dot_products_per_sec")
    s["search_cpu_dot_products_per_sec"] = g(extras, "eval_search", "cpu_baseline", "value")
    s["config5_search_dot_products_per_sec"] = g(extras, "config5_end_to_end", "search_dot_products_per_sec")
    s["config5_encode_passages_per_sec"] = g(extras, "config5_end_to_end", "encode_passages_per_sec")
    s["config3_seq_per_sec"] = g(extras, "config3_global_batch_2048", "sequences_per_sec")
    s["sharded_search_dot_products_per_sec"] = g(extras, "multi_gpu", "sharded_search", "dot_products_per_sec")
    s["sharded_encode_seq_per_sec"] = g(extras, "multi_gpu", "sharded_corpus_encode", "sequences_per_sec")
    s["dp_ance_rows_per_sec"] = g(extras, "multi_gpu", "ance_triplet_step", "rows_per_sec")
    return {k: v for k, v in s.items() if v is not None}


def write_legs(legs: dict) -> None:
    """bench_legs.json next to this file (+ gpurun_out/ when it exists), and one `[leg] name {json}` line per leg on stderr."""
    text = json.dumps(legs, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_legs.json"), "w") as f:
                    f.write(text + "\n")
            except OSError:
                pass
    for k, v in legs.items():
        print(f"[leg] {k} {json.dumps(v)}", file=sys.stderr)


def _self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="base", choices=["base", "large"])
    ap.add_argument("--seq-per-gpu", type=int, default=0,
                    help="sequences per GPU and step; default 64 (BASELINE configs[1]) at every --gpus N (weak scaling); at 8 GPUs "
                         "configs[2]'s global batch 2048 (COCO/README.md:55 NPROC x BATCH_SIZE) runs as a side leg")
    ap.add_argument("--seq-len", type=int, default=SEQ_LEN)
    ap.add_argument("--dense", action="store_true", help="every synthetic sequence fills seq_len (SURVEY 8d's roofline variant)")
    ap.add_argument("--padded", action="store_true",
                    help="headline on the padded execution (every GEMM over all B x L rows, as the reference computes); default: packed "
                         "execution of the same padded batches - identical loss and gradients, no work on padding rows")
    ap.add_argument("--host-lengths", action="store_true",
                    help="batches carry the B sequence lengths on the host next to ids + mask (a boundary extension: nothing is read back); "
                         "default: the reference's batch unchanged, {input_ids, attention_mask}")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-full-step", action="store_true", help="skip the extra legs (north-star large step, full coCondenser step, search, encode, ANCE)")
    ap.add_argument("--dp-chunks", type=int, default=2, help="layer ranges whose gradient all-reduce overlaps the backward")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(_self_launch(args))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # `value` keeps N = 1's per-GPU batch (64 sequences) at every N, so the driver's curve over --gpus 1 / 2 / 4 / 8 is a weak-scaling
    # curve; BASELINE configs[2] (8 GPUs, global batch 2048 = 256 per GPU) runs as a side leg (summary.config3_seq_per_sec)
    config3 = args.seq_per_gpu == 0 and world == 8 and args.model == "base"
    if args.seq_per_gpu == 0:
        args.seq_per_gpu = SEQ_PER_GPU

    import torch.distributed as dist
    import cocodr_amd  # noqa: F401
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    shared = world > n_dev  # fewer GPUs than ranks (a 1-GPU box running the N > 1 code path): ranks share devices
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1
    backend = None
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "gloo" if shared else "nccl"  # RCCL needs one device per rank; gloo moves the tensors through the host
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    solo = not use_dist
    packed = not args.padded
    host_lengths = bool(args.host_lengths)
    dt, final_loss, roof, cfg, (ids, mask, lens), xinfo = contrastive_leg(args.model, args.seq_per_gpu, args.seq_len, args.steps, args.warmup, dev, rank,
                                                            world, use_dist, args.dp_chunks, not args.no_roofline, args.dense, packed=packed,
                                                            host_lengths=host_lengths)
    extras = {}
    # ---- from here on the headline is measured: nothing a side leg does may cost the contract line.  A side leg that raises is
    # recorded as "<leg>_error" (stderr + bench_legs.json); one that HANGS (a collective whose peer died on a multi-GPU node) is
    # cut off by a watchdog: rank 0 prints the line with the legs finished so far and "side_legs_incomplete", every rank exits 0
    state = {"emitted": False, "emit": None}
    deadline = float(os.environ.get("COCODR_BENCH_SIDE_LEGS_DEADLINE_S", "1800" if world == 1 else "900"))

    def watchdog():
        import threading

        def fire():
            if rank == 0 and not state["emitted"] and state["emit"] is not None:
                print(f"[bench] side legs exceeded {deadline:.0f} s: printing the headline without them", file=sys.stderr, flush=True)
                try:
                    state["emit"](incomplete=True)
                except Exception as e:  # pragma: no cover
                    print(f"[bench] watchdog emit failed: {e}", file=sys.stderr, flush=True)
            os._exit(0)

        t = threading.Timer(deadline + (0.0 if rank == 0 else 5.0), fire)
        t.daemon = True
        t.start()
        return t

    def side_leg(name, fn):
        try:
            extras[name] = fn()
        except Exception as e:
            import traceback
            extras[name + "_error"] = f"{type(e).__name__}: {e}"[:300]
            print(f"[bench] side leg {name} failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)

    def leg_north_star():
        # the north-star target shape (BASELINE.json north_star: ">= 50 % MFMA roofline on BERT-large seq128 contrastive step
        # at 1 GPU"): cocodr-large at this line's 64 sequences and at COCO/README.md:59-63's per-GPU batch for the large
        # model (100 documents = 200 spans), each with its own HIP-event roofline block
        large = {}
        # (256 sequences: every GEMM of the padded step fills whole rounds of the 256 CUs - what the kernels reach without tile-count
        # tails).  Each size in both executions: packed (this line's default) and padded (executed FLOPs = the dense count)
        for n_seq, k_steps in ((64, 10), (200, 6), (256, 5)):
            for pk_ in (True, False):
                ldt, lloss, lroof, lcfg, _, linfo = contrastive_leg("large", n_seq, SEQ_LEN, k_steps, 3, dev, 0, 1, False, args.dp_chunks,
                                                                    not args.no_roofline, args.dense, packed=pk_, host_lengths=host_lengths)
                v = n_seq * k_steps / ldt
                entry = {"sequences_per_sec": round(v, 1), "ms_per_step": round(ldt / k_steps * 1e3, 3), "steps": k_steps, "loss": round(lloss, 4)}
                entry.update(whole_step_fracs(n_seq, k_steps, ldt, lcfg, SEQ_LEN, linfo))
                entry.update({"execution": "packed" if pk_ else "padded", "roofline": lroof})
                large[f"{n_seq}_sequences" + ("" if pk_ else "_padded")] = entry
        large["workload"] = "cocodr-large (BERT-large, 24 x 1024) contrastive step, seq_len 128, bf16 + fp32 accumulate, clip_grad_norm_(1.0) + AdamW, 1 GPU"
        return large

    def leg_other_boundary():
        # the same headline step with the batch in the OTHER boundary form: host-known lengths next to ids + mask (a collator that
        # pads on the CPU has them; nothing is read back) when the headline takes the reference's batch unchanged, and vice versa
        # (with the same roofline sampling as the headline leg - its event pairs cost ~1.5 % of the timed region - so that the
        # two numbers differ by the boundary form alone)
        hdt, hloss, _, _, _, _ = contrastive_leg(args.model, args.seq_per_gpu, args.seq_len, args.steps, args.warmup, dev, 0, 1, False,
                                                 args.dp_chunks, not args.no_roofline, args.dense, packed=True, host_lengths=not host_lengths)
        hv = args.seq_per_gpu * args.steps / hdt
        return {"sequences_per_sec": round(hv, 1), "ms_per_step": round(hdt / args.steps * 1e3, 3), "loss": round(hloss, 4),
                "headline_vs_this": round((args.seq_per_gpu * args.steps / dt) / hv, 4),
                "batch": "{input_ids, attention_mask} only (COCO/data.py:150-154): layout planned on the device, 16 bytes read back" if host_lengths
                         else "{input_ids, attention_mask, lengths}: the B lengths on the host, nothing read back"}

    def leg_other_execution():
        # the same headline step in the other execution: padded (every GEMM over all B x L rows, executed FLOPs = the dense count: the
        # line that is comparable kernel for kernel with rounds 1-2) when the headline is packed, and the other way round
        odt, oloss, oroof, _, _, oinfo = contrastive_leg(args.model, args.seq_per_gpu, args.seq_len, args.steps, args.warmup, dev, 0, 1, False,
                                                         args.dp_chunks, not args.no_roofline, args.dense, packed=not packed, host_lengths=host_lengths)
        ov = args.seq_per_gpu * args.steps / odt
        other = {"sequences_per_sec": round(ov, 1), "ms_per_step": round(odt / args.steps * 1e3, 3), "loss": round(oloss, 4),
                 "headline_vs_this": round((args.seq_per_gpu * args.steps / dt) / ov, 3)}
        other.update(whole_step_fracs(args.seq_per_gpu, args.steps, odt, cfg, args.seq_len, oinfo))
        other.update({"roofline": oroof,
                      "note": "same batches, same loss and gradients as the headline step (tests/test_gpu_packed.py); "
                              + ("here every kernel runs over all B x L rows, padding included, as the reference does "
                                 "(CocoBertModel.pack_sequences = False)" if packed else
                                 "here the sequences are stored back to back, every one on its own length (no work on padding rows)")})
        return other

    def leg_search():
        r_ = eval_search(dev)
        if not args.no_cpu_baseline:
            r_["cpu_baseline"] = search_cpu_baseline()
        return r_

    def solo_side_legs():
        side_leg("north_star_large_step", leg_north_star)
        if packed:
            side_leg("reference_batch_contrastive_step" if host_lengths else "host_lengths_contrastive_step", leg_other_boundary)
        side_leg("padded_contrastive_step" if packed else "packed_contrastive_step", leg_other_execution)
        if args.model == "base":
            # second scope: what the reference's step really runs
            side_leg("full_coco_step", lambda: full_coco_step(cfg, dev, ids, mask, lens if host_lengths else None))
            side_leg("ance_triplet_step", lambda: ance_step(dev, host_lengths=host_lengths))
        side_leg("corpus_encode", lambda: corpus_encode(cfg, dev, seq_len=args.seq_len))
        side_leg("eval_search", leg_search)
        if args.model == "base":
            side_leg("config5_end_to_end", lambda: config5_end_to_end(dev))

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def tmax(x: float) -> float:
        if not use_dist:
            return x
        t = torch.tensor([x], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    dt = tmax(dt)

    def release():  # ranks that share a GPU (the one-GPU code-path check) also share its memory: hand the finished leg's cached blocks back
        if shared:
            import gc
            gc.collect()
            torch.cuda.empty_cache()

    def multi_side_legs():
        release()
        if config3:  # BASELINE configs[2]: cocodr-base on 8 GPUs with RCCL all_gather negatives, global batch 2048 (COCO/README.md:55)
            def leg_config3():
                wdt, wloss, _, _, _, _ = contrastive_leg(args.model, 256, args.seq_len, args.steps, args.warmup, dev, rank, world, use_dist,
                                                      args.dp_chunks, False, args.dense, packed=packed, host_lengths=host_lengths)
                wdt = tmax(wdt)
                return {"sequences_per_sec": round(256 * world * args.steps / wdt, 2), "ms_per_step": round(wdt / args.steps * 1e3, 3),
                        "global_batch": 256 * world, "loss": round(wloss, 4),
                        "note": "256 sequences per GPU (BASELINE configs[2]); NOT comparable with the N = 1, 2, 4 "
                                "lines' 64 per GPU - the headline `value` of this line is"}
            side_leg("config3_global_batch_2048", leg_config3)
            release()
        side_leg("multi_gpu", lambda: multi_gpu_legs(dev, rank, world, fence, tmax, shared, args.dp_chunks))

    def emit(incomplete=False):
        if rank != 0 or state["emitted"]:
            return
        state["emitted"] = True
        n_seq = args.seq_per_gpu * world * args.steps
        value = n_seq / dt
        par = f"dp{world}"
        if world > 1:
            par += " + RCCL all_gather negatives" if backend == "nccl" else f" over gloo ({world} ranks sharing {n_dev} GPU(s): code-path check, not a scaling number)"
        which = ("BASELINE configs[1]" if args.model == "base" and world == 1 else
                 "BASELINE configs[1]'s batch per GPU, negatives all-gathered as in configs[2]" if args.model == "base" else
                 "north_star BERT-large target shape")
        boundary = ("{input_ids, attention_mask, lengths}: B lengths on the host" if host_lengths else
                    "{input_ids, attention_mask} as the reference's collator emits them (COCO/data.py:150-154)")
        config = {"workload": f"cocodr-{args.model} contrastive step (COCO in-batch negatives), seq_len={args.seq_len}, {args.seq_per_gpu} sequences/GPU "
                              f"padded to seq_len (MS MARCO-shaped lengths), bf16 + fp32 accumulate, clip_grad_norm_(1.0) + AdamW; {which}",
                  "global_batch": args.seq_per_gpu * world, "seq_len": args.seq_len, "execution": "packed" if packed else "padded",
                  "batch": boundary, "parallelism": par}
        out = {
            "metric": "contrastive-step sequences/sec", "value": round(value, 2), "unit": "sequences/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": config, "loss": round(final_loss, 4),
        }
        fracs = whole_step_fracs(args.seq_per_gpu * world, args.steps, dt, cfg, args.seq_len, xinfo, world)
        out.update(fracs)
        if roof is not None:
            out["roofline"] = compact_roofline(roof)
        cpu = None
        if not args.no_cpu_baseline and world == 1 and not incomplete:
            cpu = cpu_baseline()
            out["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample", "cpu") if k in cpu}
        out["summary"] = leg_summary(extras)
        failed = sorted(k[:-6] for k in extras if k.endswith("_error"))
        if incomplete or failed:  # (absent from a normal line)
            out["side_legs_incomplete"] = {"watchdog": bool(incomplete), "failed": failed}
        out["legs_file"] = "bench_legs.json"
        # ---- every leg in full: side file + stderr, never the contract line
        legs = {"headline": {**out, "roofline": roof, "cpu_baseline": cpu,
                             "config_notes": {
                                 "execution": "packed: sequences stored back to back inside the encoder, no work on padding rows, loss and gradients "
                                              "identical to the padded execution (tests/test_gpu_packed.py)" if packed else "padded: all B x L rows",
                                 "batches": "8 pre-generated synthetic batches per rank (padded ids + attention mask in HBM" +
                                            (", the B lengths on the host" if host_lengths else "") + "), visited round-robin; every timed step "
                                            "gets a fresh batch dict" + (" and builds its packed layout inside the step (nothing prebuilt or hoisted)"
                                                                         if packed else "")}}}
        legs.update(extras)
        write_legs(legs)
        line = json.dumps(out, separators=(",", ":"))
        if len(line) >= 4096:  # the driver's parser gave up on a 20 KB line in round 4: never again
            for k in ("summary", "legs_file", "loss"):
                out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
        sys.stderr.flush()
        print(line, flush=True)
    state["emit"] = emit
    dog = watchdog()
    if not args.no_full_step:
        if solo:
            solo_side_legs()
        else:
            multi_side_legs()
    emit()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    dog.cancel()


if __name__ == "__main__":
    main()
