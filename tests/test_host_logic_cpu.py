"""Host logic added in round 5 that needs no GPU: the chunk / one-pass plans of FlatLamb (which tensor goes through which kernel,
nothing covered twice or not at all), bench.py's compact contract line (size, keys, no prose in `roofline`) and its side file."""
import io
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import cocodr_amd  # noqa: E402,F401
from cocodr_amd.optim import LAMB_FUSED_MIN, lamb_fused_plan, lamb_plan  # noqa: E402


def test_lamb_plans_partition_the_flat_between_the_one_pass_and_the_two_pass_kernels():
    sizes = [1024, 3 * 1024 * 1024, 64, 1024 * 1024, LAMB_FUSED_MIN, LAMB_FUSED_MIN - 64, 5 * 1024 * 1024, 4096]
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o = (o + n + 63) // 64 * 64
    cap = 4 * 1024 * 1024
    idx, f_start, f_len, f_seg, wg_begin, wg_count, round_first = lamb_fused_plan(offs, o, cap, workgroups=256, wg_elements=16384)
    # rounds: 3 M + 1 M fill the 256 workgroups exactly (192 + 64); the 2^18 tensor gets a round - and all its workgroups - to itself
    assert round_first.tolist() == [0, 2, 3] and wg_begin.tolist() == [0, 192, 0] and wg_count.tolist() == [192, 64, 256]
    assert all(int(c) * 16384 >= int(n) for c, n in zip(wg_count, f_len))
    assert idx == [1, 3, 4] and f_seg.tolist() == idx                       # >= 2^18 and <= capacity; the 5 M tensor is too large
    assert f_start.tolist() == [offs[i] for i in idx] and all(int(n) % 4 == 0 for n in f_len)
    assert f_len.tolist() == [(offs + [o])[i + 1] - offs[i] for i in idx]   # alignment padding rides with the tensor in front of it
    start, length, seg, seg_begin = lamb_plan(offs, o, skip=idx)
    covered = np.zeros(o, np.int32)
    for a, n in zip(start, length):
        covered[a:a + n] += 1
    for a, n in zip(f_start, f_len):
        covered[a:a + n] += 1
    assert (covered == 1).all()                                              # every element exactly once
    assert len(seg_begin) == len(sizes) + 1
    for s in range(len(sizes)):                                              # skipped tensors have no chunks, the others only their own
        mine = seg[seg_begin[s]:seg_begin[s + 1]]
        assert (len(mine) == 0) == (s in idx) and (mine == s).all()
    # nothing qualifies: empty one-pass plan, the chunk plan covers everything
    idx0, *_ = lamb_fused_plan(offs, o, capacity=1024, workgroups=1, wg_elements=1024)
    assert idx0 == [] and sum(lamb_plan(offs, o)[1]) == o
    # capacity 0 (no co-resident grid on the device) never selects anything
    assert lamb_fused_plan(offs, o, 0)[0] == []


def _fake_roof():
    return {"bound": "mfma", "achieved": 610.2, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.2441, "traffic": 96008284,
            "traffic_unit": "x" * 300, "traffic_per_step_bytes": 9600828400, "traffic_gbps": 2613.2, "traffic_source": {"file": "f", "commit": "c"},
            "algorithmic_achieved": 1136.0, "algorithmic_frac": 0.4545, "kernel": "y" * 200, "launches_per_step": 100, "sampled": "z" * 200,
            "avg_launch_us": 36.5, "gemm_share_of_step": 0.69, "executed_gemm_flops_per_step": 2.3e12, "flops": "w" * 600, "batches": "v" * 300,
            "rows_per_step": 4776, "rows_per_step_padded": 8192}


def test_compact_roofline_keeps_numbers_and_one_short_name():
    import bench
    r = bench.compact_roofline(_fake_roof())
    assert r["frac"] == 0.2441 and r["achieved"] == 610.2 and r["peak"] == 2500.0 and r["bound"] == "mfma" and r["traffic"] == 96008284
    assert all(not isinstance(v, str) or len(v) <= 100 for v in r.values()) and len(r["kernel"]) <= 100
    assert "flops" not in r and "batches" not in r and "sampled" not in r
    assert r["traffic_source"] == {"file": "f", "commit": "c"}                # where the PMC passes behind `traffic` live: stays in the line


def test_leg_summary_and_side_file(tmp_path, monkeypatch, capsys):
    import bench
    extras = {"north_star_large_step": {"256_sequences_padded": {"roofline": _fake_roof(), "executed_whole_step_frac": 0.39, "sequences_per_sec": 4200.0},
                                        "256_sequences": {"sequences_per_sec": 6000.0}, "workload": "text"},
              "host_lengths_contrastive_step": {"sequences_per_sec": 12440.5}, "eval_search": {"dot_products_per_sec": 130e9, "cpu_baseline": {"value": 2.2e8}},
              "ance_triplet_step": {"rows_per_sec": 1843.0, "roofline": {"frac": 0.33}}}
    s = bench.leg_summary(extras)
    assert s["large_256_padded_gemm_frac"] == 0.2441 and s["large_256_padded_step_frac"] == 0.39 and s["host_lengths_seq_per_sec"] == 12440.5
    assert s["search_dot_products_per_sec"] == 130e9 and s["search_cpu_dot_products_per_sec"] == 2.2e8 and s["ance_gemm_frac"] == 0.33
    assert "full_coco_seq_per_sec" not in s and all(isinstance(v, (int, float)) for v in s.values())   # absent legs leave no key
    # a contract line built like main() builds it stays far below the driver's 4 KB
    line = json.dumps({"metric": "contrastive-step sequences/sec", "value": 12413.83, "unit": "sequences/sec", "n_gpus": 1, "steps": 20, "warmup": 5,
                       "ms_per_step": 5.156, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                       "config": {"workload": "w" * 320, "global_batch": 64, "seq_len": 128, "execution": "packed", "batch": "b" * 100, "parallelism": "dp1"},
                       "roofline": bench.compact_roofline(_fake_roof()), "cpu_baseline": {"value": 19.3, "unit": "sequences/sec", "cores": 16, "kind": "port",
                                                                                         "sample": "s" * 220, "cpu": "c" * 40},
                       "summary": s, "legs_file": "bench_legs.json"}, separators=(",", ":"))
    assert len(line) < 4096
    # write_legs: the side file next to bench.py (here: a scratch root) and one "[leg] name {json}" line per leg on stderr, nothing on stdout
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.write_legs({"headline": {"value": 1.0}, **extras})
    out = capsys.readouterr()
    assert out.out == "" and out.err.count("[leg] ") == len(extras) + 1
    assert json.load(open(tmp_path / "bench_legs.json"))["ance_triplet_step"]["rows_per_sec"] == 1843.0
