"""The driver's contract with bench.py: one JSON line on stdout with the agreed keys, the roofline block measured in the run, and
the multi-rank code path (two ranks sharing the GPU fall back to gloo and say so)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, legs: bool = False, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    legs_path = os.path.join(ROOT, "bench_legs.json")
    if os.path.exists(legs_path):
        os.remove(legs_path)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    if p.returncode != 0:  # a failed rank's traceback sits far above torchrun's summary: show the first one, then the tail
        at = p.stderr.find("Traceback")
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_contract_stderr.txt"), "w") as f:
            f.write(p.stderr)
        raise AssertionError((p.stderr[at:at + 3000] if at >= 0 else "") + "\n...\n" + p.stderr[-1500:])
    out = [l for l in p.stdout.splitlines() if l.strip()]
    # the driver's contract: the LAST stdout line is the one compact JSON object (< 4 KB; round 4's 20 KB line was not parsed), and no
    # other stdout line looks like JSON
    assert out and out[-1].startswith('{"metric"') and len(out[-1].encode()) < 4096, (len(out[-1]) if out else 0, p.stdout[-500:])
    assert sum(l.lstrip().startswith("{") for l in out) == 1, p.stdout[-2000:]
    d = json.loads(out[-1])
    for v in d.get("roofline", {}).values():  # numbers + one short kernel name, no prose
        assert not isinstance(v, str) or len(v) <= 100
    if not legs:
        return d
    assert d["legs_file"] == "bench_legs.json"
    with open(legs_path) as f:
        return d, json.load(f)


def test_bench_line_has_the_contract_keys_and_a_measured_roofline():
    d, legs = _run("--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-full-step", legs=True)
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
    # frac prices KERNEL time: the event pairs' fixed cost (measured in the run around an empty kernel) is taken out per launch, and
    # the traffic figure - read from committed PMC passes, not measured in this run - says so in the line itself
    assert 0.0 <= r["event_overhead_us"] < 8.0
    assert r["traffic"] is None or (set(r["traffic_source"]) == {"file", "commit"} and r["traffic_source"]["file"].startswith("profiles/"))
    assert 5000 < d["value"] < 50000  # an MI355X, not a fallback
    # default execution: packed (no work on padding rows).  `achieved` / `frac` count the FLOPs the timed launches EXECUTE (the same
    # meaning as rounds 1-2), `algorithmic_*` the padded-token count of SURVEY 8d over the same time
    assert d["config"]["execution"] == "packed"
    assert 0.05 < r["frac"] < r["algorithmic_frac"] and r["rows_per_step"] < r["rows_per_step_padded"] == 64 * 128
    # (executed: the stored rows, and in the last layer - whose [CLS] rows alone are consumed - 18 of the 24 H^2 per token on those
    # rows only: (1 - 0.75 / 12) of the stored-row count for the 12 layers of BERT-base)
    ratio = r["rows_per_step"] / r["rows_per_step_padded"] * (1 - 0.75 / 12)
    assert abs(r["achieved"] / r["algorithmic_achieved"] - ratio) < 0.03
    # nothing hoisted: every timed step received a fresh batch dict without a prebuilt packed layout and built its own - from the
    # reference's batch unchanged: {input_ids, attention_mask}, no lengths key
    assert d["fresh_batches_every_step"] is True and "nothing prebuilt" in legs["headline"]["config_notes"]["batches"]
    assert "lengths" not in d["config"]["batch"] and "COCO/data.py:150-154" in d["config"]["batch"]
    assert d["executed_whole_step_frac"] < d["algorithmic_whole_step_frac"]
    assert r["traffic"] is None or (r["traffic_per_step_bytes"] == r["traffic"] * r["launches_per_step"] and r["traffic_gbps"] > 0)
    # the full roofline block (with its prose) lives in the side file
    assert legs["headline"]["roofline"]["frac"] == r["frac"] and "flops" in legs["headline"]["roofline"]


def test_default_flags_line_is_compact_and_the_legs_go_to_the_side_file():
    """`python bench.py` exactly as the driver runs it (default flags, every side leg, both CPU baselines): the last stdout line
    parses, is under 4 KB and carries roofline + ONE cpu_baseline block + one-number summaries; the legs are in bench_legs.json."""
    d, legs = _run(legs=True)
    assert (d["steps"], d["warmup"], d["n_gpus"]) == (20, 5, 1)
    r, c, s_ = d["roofline"], d["cpu_baseline"], d["summary"]
    assert 0.05 < r["frac"] < 1.0 and r["bound"] == "mfma" and len(r["kernel"]) <= 100
    assert set(c) <= {"value", "unit", "cores", "kind", "sample", "cpu"} and c["value"] > 0 and c["kind"] == "port" and c["cores"] >= 1
    assert "median of 5 steps" in c["sample"]
    for k in ("large_256_padded_gemm_frac", "large_256_padded_step_frac", "host_lengths_seq_per_sec", "padded_seq_per_sec", "ance_rows_per_sec",
              "full_coco_seq_per_sec", "search_dot_products_per_sec", "search_cpu_dot_products_per_sec", "config5_search_dot_products_per_sec"):
        assert s_[k] > 0, k
    # the reference-shaped batch is the headline; knowing the lengths on the host is worth ~1 % (same-box A/Bs: 0.6-1.0 %; VERDICT r04
    # item 2 asks for <= 2 %; two 20-step legs of one run differ by up to ~1 % on their own, hence 3 % here)
    assert d["value"] > 0.97 * s_["host_lengths_seq_per_sec"], (d["value"], s_["host_lengths_seq_per_sec"])
    for k in ("headline", "north_star_large_step", "host_lengths_contrastive_step", "padded_contrastive_step", "full_coco_step", "ance_triplet_step",
              "corpus_encode", "eval_search", "config5_end_to_end"):
        assert k in legs, k
    assert legs["north_star_large_step"]["256_sequences_padded"]["roofline"]["frac"] == s_["large_256_padded_gemm_frac"]
    assert legs["headline"]["cpu_baseline"]["port"]["value"] > 0 and legs["corpus_encode"]["packed_equals_padded"] is True


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
    dev = torch.device("cuda", 0)
    for host_lengths in (False, True):  # the reference's batch unchanged (layout planned on the device), and with host-known lengths
        calls = []
        orig = modeling.PackedIndex.build

        def counting(*a, **k):
            calls.append(1)
            return orig(*a, **k)

        modeling.PackedIndex.build = staticmethod(counting)
        try:
            dt, loss, roof, cfg, _, info = bench.contrastive_leg("base", 16, 128, 3, 1, dev, 0, 1, False, 2, False, packed=True,
                                                                 host_lengths=host_lengths)
        finally:
            modeling.PackedIndex.build = staticmethod(orig)
        assert len(calls) == 4 and info["fresh_batches"] is True and info["rows_per_step"] < 16 * 128
    torch.cuda.empty_cache()  # the multi-rank tests below start ranks that share this GPU with the test process


def test_bench_multi_rank_path_on_one_gpu():
    d, legs = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", legs=True)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["scaling"] == "weak"
    assert "gloo" in d["config"]["parallelism"] or "RCCL" in d["config"]["parallelism"]
    assert d["value"] > 0
    # N > 1 also measures what BASELINE configs[3] / [4] name: sharded encode, sharded search (merge inside the timed region)
    # and the data-parallel ANCE step - full legs in the side file, one number each in the line
    m = legs["multi_gpu"]
    assert m["sharded_corpus_encode"]["sequences_per_sec"] > 0
    assert m["sharded_search"]["dot_products_per_sec"] > 0 and m["sharded_search"]["result_rows"] == 2000
    assert m["ance_triplet_step"]["rows_per_sec"] > 0 and m["ance_triplet_step"]["loss"] > 0
    s_ = d["summary"]
    assert s_["sharded_search_dot_products_per_sec"] == m["sharded_search"]["dot_products_per_sec"] and s_["dp_ance_rows_per_sec"] > 0


def test_bench_eight_rank_path_keeps_the_per_gpu_batch_on_one_gpu():
    """`bench.py --gpus 8` as the driver would launch it on an 8-GPU node, here with the eight ranks sharing the one GPU over gloo:
    `value` keeps N = 1's 64 sequences per GPU (global batch 512: the driver's curve over N is a weak-scaling curve); BASELINE
    configs[2] (256 per GPU, global batch 2048) and the multi_gpu legs (sharded encode / search / data-parallel ANCE step) are side legs."""
    d, legs = _run("--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", legs=True)
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 512 and d["scaling"] == "weak"
    assert "64 sequences/GPU" in d["config"]["workload"] and d["value"] > 0 and d["loss"] > 0
    assert "gloo" in d["config"]["parallelism"] or "RCCL" in d["config"]["parallelism"]
    assert legs["config3_global_batch_2048"]["global_batch"] == 2048 and d["summary"]["config3_seq_per_sec"] > 0
    m = legs["multi_gpu"]
    assert m["sharded_corpus_encode"]["sequences_per_sec"] > 0 and m["sharded_search"]["result_rows"] == 2000
    assert m["ance_triplet_step"]["rows_per_sec"] > 0


def test_a_side_leg_that_never_returns_cannot_cost_the_contract_line():
    """The headline is measured first; the side legs run under a watchdog (bench.py main): past the deadline rank 0 prints the line with
    the legs finished so far and every rank exits 0 - on a multi-GPU node a side-leg collective whose peer died would otherwise hang
    the run and lose the measured headline.  Here: a 3 s deadline on the single-GPU run, whose side legs take minutes."""
    d = _run("--steps", "4", "--warmup", "2", "--no-cpu-baseline", extra_env={"COCODR_BENCH_SIDE_LEGS_DEADLINE_S": "3"})
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    assert d["side_legs_incomplete"]["watchdog"] is True
    assert "config5_encode_passages_per_sec" not in d.get("summary", {})
