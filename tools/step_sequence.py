"""Kernel sequence of ONE training step from a rocprofv3 --kernel-trace CSV: name (shortened), duration us, gap to the previous
kernel's end us.  Usage: python tools/step_sequence.py <kernel_trace.csv> [marker kernel substring that starts a step = pack_index]"""
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "mask_lengths_kernel"
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
prev_end = None
tot = gaps = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::|^void |cocodr_gemm_pp::|cocodr_gemm_v2::|at::native::|\(.*$", "", r["Kernel_Name"])[:70]
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    print(f"{name:70s} {(e - s) / 1e3:8.1f} {gap:7.1f}")
    tot += (e - s) / 1e3
    gaps += max(gap, 0.0)
    prev_end = e
print(f"kernels {b - a}  kernel time {tot:.0f} us  gaps {gaps:.0f} us  span {(int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.0f} us")
