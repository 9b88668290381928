#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) into the per-kernel stats table that
`--stats` prints: calls, total / average / min / max duration, share of GPU time.
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--grid] > profiles/NAME.md"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main(path: str, by_grid: bool = False):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    key = f"name, {gcol}, grid_y" if (by_grid and gcol and "grid_y" in cols) else "name"
    rows = list(cur.execute(f"select {key}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {key}"))
    nk = len(key.split(","))
    total = sum(r[nk + 1] for r in rows) or 1
    rows.sort(key=lambda r: -r[nk + 1])
    print(f"| kernel | {'grid | ' if nk > 1 else ''}calls | total ms | avg us | min us | max us | % |")
    print("|---|" + ("---|" if nk > 1 else "") + "---:|---:|---:|---:|---:|---:|")
    for r in rows:
        name = short(r[0])
        grid = f"{r[1]}x{r[2]} | " if nk > 1 else ""
        c, tot, avg, mn, mx = r[nk:]
        print(f"| `{name}` | {grid}{c} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} |")
    print(f"\ntotal kernel time: {total / 1e6:.3f} ms over {sum(r[nk] for r in rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], "--grid" in sys.argv)
