#!/usr/bin/env python3
"""One training step as the GPU ran it: every kernel of the LAST step of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py
in launch order - start offset, duration, idle gap in front of it - and the GEMM launches grouped by kernel family.
Usage: python tools/step_timeline.py <kernel_trace.csv> [steps in the run] > profiles/...md"""
import csv
import re
import sys


def main(path, steps=9):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the steps are periodic: find the last occurrence of the step's first kernel (the embedding forward)
    firsts = [i for i, r in enumerate(rows) if "embed_ln_fwd_kernel" in r[2]]
    a = firsts[-1]
    # a step ends with its last optimizer kernel
    b = max(i for i, r in enumerate(rows) if i >= a and ("adamw_kernel" in r[2] or "lamb" in r[2])) + 1
    step = rows[a:b]
    t0 = step[0][0]
    short = lambda n: re.sub(r"\(anonymous namespace\)::|^void |cocodr_gemm_pp::|cocodr_gemm_v2::|cocodr_gemm_a4::|\(.*$", "", n)[:64]
    print(f"kernels of the step: {len(step)}; first start -> last end: {(step[-1][1] - t0) / 1e3:.1f} us; "
          f"sum of kernel durations: {sum(e - s for s, e, _ in step) / 1e3:.1f} us; "
          f"sum of idle gaps between consecutive kernels: {sum(max(0, step[i][0] - step[i - 1][1]) for i in range(1, len(step))) / 1e3:.1f} us\n")
    fam = {}
    for i, (s, e, n) in enumerate(step):
        k = short(n)
        gap = max(0, s - step[i - 1][1]) if i else 0
        d = fam.setdefault(k, [0, 0.0, 0.0])
        d[0] += 1; d[1] += (e - s) / 1e3; d[2] += gap / 1e3
    print("| kernel | launches | total us | avg us | idle in front, total us |\n|---|---:|---:|---:|---:|")
    for k, (n, t, g) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {n} | {t:.1f} | {t / n:.2f} | {g:.1f} |")
    print("\nOne encoder layer of the forward and of the backward, in launch order (layer 6 of 12):\n")
    print("| # | start us | kernel | duration us | idle in front us |\n|---:|---:|---|---:|---:|")
    attn = [i for i, r in enumerate(step) if "attn_fwd_kernel" in r[2]]
    attb = [i for i, r in enumerate(step) if "attn_bwd" in r[2]]
    spans = []
    if len(attn) >= 7:
        spans.append((attn[5] + 1, attn[6] + 1))
    if len(attb) >= 7:
        spans.append((attb[5] + 1, attb[6] + 1))
    for lo, hi in spans:
        for i in range(lo, hi):
            s, e, n = step[i]
            print(f"| {i} | {(s - t0) / 1e3:.1f} | `{short(n)}` | {(e - s) / 1e3:.2f} | {max(0, s - step[i - 1][1]) / 1e3:.2f} |")
        print("| | | | | |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 9)
