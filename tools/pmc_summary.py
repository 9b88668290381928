#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc CSV output (x_counter_collection.csv).
Usage: python tools/pmc_summary.py gpurun_out/pmc2/FETCH_SIZE/x_counter_collection.csv [more.csv ...]"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([\w:]+(<[^>]*>)?)", n)
    return (m.group(1) if m else n)[:70]


def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        for r in csv.DictReader(open(path)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for v in agg.values() for c in v})
    print("| kernel | launches | " + " | ".join(f"avg {c}" for c in counters) + " |")
    print("|---|---:|" + "---:|" * len(counters))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(sum(x) for x in kv[1].values())):
        n = max(len(x) for x in v.values())
        print(f"| `{k}` | {n} | " + " | ".join(f"{sum(v[c]) / len(v[c]):.4g}" if c in v else "-" for c in counters) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
