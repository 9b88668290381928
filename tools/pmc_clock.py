#!/usr/bin/env python3
"""Effective shader clock and matrix-pipe utilisation per kernel from ONE rocprofv3 --pmc pass that holds GRBM_GUI_ACTIVE and
SQ_VALU_MFMA_BUSY_CYCLES (counter_collection.csv carries the dispatch timestamps):
    clock      = GRBM_GUI_ACTIVE / 8 XCDs / (end - start)
    MFMA busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)     (pipe cycles per SIMD cycle, at the ACTUAL clock)
Usage: python tools/pmc_clock.py <x_counter_collection.csv> [substring of the kernel names to keep]"""
import collections
import csv
import re
import sys


def main(path, keep=""):
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if keep and keep not in r["Kernel_Name"]:
            continue
        d = rows[(r["Dispatch_Id"], r["Kernel_Name"], r["Grid_Size"])]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    agg = collections.defaultdict(list)
    for (_, name, grid), d in rows.items():
        if "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d and d["dur"] > 0:
            cyc = d["GRBM_GUI_ACTIVE"] / 8.0
            name = re.sub(r"\(anonymous namespace\)::|^void |cocodr_gemm_pp::|cocodr_gemm_v2::|cocodr_gemm_a4::|\(.*$", "", name)[:60]
            agg[(name, grid)].append((d["dur"], cyc / d["dur"], d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc),
                                      d.get("SQ_WAIT_INST_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1), d.get("SQ_WAIT_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1)))
    print("| kernel | grid | launches | avg us | clock GHz | MFMA pipe busy (at that clock) | x clock / 2.4 = of nominal peak | issue-stalled | parked |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        n = len(v)
        dur, clk, util, wi, wa = (sum(x[i] for x in v) / n for i in range(5))
        print(f"| `{name}` | {grid} | {n} | {dur * 1e6:.1f} | {clk / 1e9:.2f} | {util:.3f} | {util * clk / 2.4e9:.3f} | {wi:.2f} | {wa:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
