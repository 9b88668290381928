"""The driver's contract with bench.py: one JSON line on stdout with the agreed keys, the roofline block measured in the run, and
the multi-rank code path (two ranks sharing the GPU fall back to gloo and say so)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # exactly one JSON line
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys_and_a_measured_roofline():
    d = _run("--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-full-step")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "contrastive-step sequences/sec" and d["unit"] == "sequences/sec" and d["n_gpus"] == 1
    assert (d["steps"], d["warmup"]) == (4, 2) and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 64 * 1000.0 / d["ms_per_step"]) < 0.01 * d["value"]  # whole-job sequences per second
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert 0.05 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    assert 5000 < d["value"] < 50000  # an MI355X, not a fallback
    # default execution: packed (no work on padding rows).  `achieved` / `frac` count the FLOPs the timed launches EXECUTE (the same
    # meaning as rounds 1-2), `algorithmic_*` the padded-token count of SURVEY 8d over the same time
    assert d["config"]["execution"] == "packed" and "packed" in d["config"]["workload"]
    assert 0.05 < r["frac"] < r["algorithmic_frac"] and r["rows_per_step"] < r["rows_per_step_padded"] == 64 * 128
    # (executed: the stored rows, and in the last layer - whose [CLS] rows alone are consumed - 18 of the 24 H^2 per token on those
    # rows only: (1 - 0.75 / 12) of the stored-row count for the 12 layers of BERT-base)
    ratio = r["rows_per_step"] / r["rows_per_step_padded"] * (1 - 0.75 / 12)
    assert abs(r["achieved"] / r["algorithmic_achieved"] - ratio) < 0.03
    # nothing hoisted: every timed step received a fresh batch dict without a prebuilt packed layout and built its own
    assert d["fresh_batches_every_step"] is True and "nothing prebuilt" in d["config"]["batches"]
    assert d["executed_whole_step_frac"] < d["algorithmic_whole_step_frac"]
    assert r["traffic"] is None or (r["traffic_per_step_bytes"] == r["traffic"] * r["launches_per_step"] and r["traffic_gbps"] > 0)


def test_bench_padded_execution_executes_the_algorithmic_flops():
    d = _run("--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-full-step", "--padded")
    r = d["roofline"]
    assert d["config"]["execution"] == "padded"
    # every GEMM runs over all B x L rows, but for the last layer's output projection and FFN ([CLS] rows only)
    assert abs(r["achieved"] / r["algorithmic_achieved"] - (1 - 0.75 / 12)) < 0.01


def test_timed_step_receives_fresh_batches_and_packs_inside_the_step():
    """bench.contrastive_leg directly: no batch carries a prebuilt `packed_index`, each step sees a new dict, and the packed
    layout of step i is built inside model(batch) - counted through the native launch's Python entry."""
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from cocodr_amd import modeling
    calls = []
    orig = modeling.PackedIndex.__init__

    def counting(self, *a, **k):
        calls.append(1)
        return orig(self, *a, **k)

    modeling.PackedIndex.__init__ = counting
    try:
        dt, loss, roof, cfg, _, info = bench.contrastive_leg("base", 16, 128, 3, 1, torch.device("cuda", 0), 0, 1, False, 2, False, packed=True)
    finally:
        modeling.PackedIndex.__init__ = orig
    assert len(calls) == 4 and info["fresh_batches"] is True and info["rows_per_step"] < 16 * 128


def test_bench_multi_rank_path_on_one_gpu():
    d = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline")
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["scaling"] == "weak"
    assert "gloo" in d["config"]["parallelism"] or "RCCL" in d["config"]["parallelism"]
    assert d["value"] > 0
    # N > 1 also measures what BASELINE configs[3] / [4] name: sharded encode, sharded search (merge inside the timed region)
    # and the data-parallel ANCE step
    m = d["multi_gpu"]
    assert m["sharded_corpus_encode"]["sequences_per_sec"] > 0
    assert m["sharded_search"]["dot_products_per_sec"] > 0 and m["sharded_search"]["result_rows"] == 2000
    assert m["ance_triplet_step"]["rows_per_sec"] > 0 and m["ance_triplet_step"]["loss"] > 0


def test_bench_eight_rank_path_is_configs2_on_one_gpu():
    """`bench.py --gpus 8` as the driver would launch it on an 8-GPU node, here with the eight ranks sharing the one GPU over gloo:
    the default becomes BASELINE configs[2] (256 sequences per GPU, global batch 2048), the line carries the 64-per-GPU weak-scaling
    point and the multi_gpu legs (sharded encode / search / data-parallel ANCE step)."""
    d = _run("--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline")
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 2048 and d["scaling"] == "weak"
    assert "configs[2]" in d["config"]["workload"] and d["value"] > 0 and d["loss"] > 0
    assert d["same_per_gpu_batch_as_n1"]["global_batch"] == 512 and d["same_per_gpu_batch_as_n1"]["sequences_per_sec"] > 0
    m = d["multi_gpu"]
    assert m["sharded_corpus_encode"]["sequences_per_sec"] > 0 and m["sharded_search"]["result_rows"] == 2000
    assert m["ance_triplet_step"]["rows_per_sec"] > 0
