#!/usr/bin/env python3
"""Largest idle gaps between consecutive kernels in the last steps of a rocprofv3 kernel-trace csv of bench.py.
python tools/step_gaps.py <kernel_trace.csv>"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
firsts = [i for i, r in enumerate(rows) if "embed_ln_fwd_kernel" in r[2]]
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void |cocodr_gemm_pp::|cocodr_gemm_v2::|cocodr_gemm_a4::|\(.*$", "", n)[:50]
for a in firsts[-3:]:
    cand = [i for i, r in enumerate(rows) if i >= a and i < a + 600 and ("adamw_kernel" in r[2] or "lamb" in r[2])]
    if not cand:
        continue
    b = max(cand) + 1
    step = rows[a:b]
    t0 = step[0][0]
    gaps = sorted(((step[i][0] - step[i - 1][1], i) for i in range(1, len(step))), reverse=True)
    tot = sum(max(0, g) for g, _ in gaps)
    print(f"step of {len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us, idle {tot / 1e3:.1f} us; gap in front of the step's first kernel {(rows[a][0] - rows[a - 1][1]) / 1e3:.1f} us")
    for g, i in gaps[:4]:
        print(f"   {g / 1e3:7.1f} us in front of #{i} {short(step[i][2])} (behind {short(step[i - 1][2])})")
