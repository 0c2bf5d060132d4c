#!/usr/bin/env python
"""Condense rocprofv3 `--kernel-trace --stats` csv output into a small per-kernel table (what profiles/ keeps).

    python tools/prof_summary.py gpurun_out/prof > profiles/r01_bench_kernel_stats.txt"""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        print("no *kernel_stats.csv under", d)
        return 1
    rows = []
    with open(files[0], newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
    tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    print(f"# source: {os.path.relpath(files[0])}")
    print(f"{'kernel':84s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = r["Name"].replace("void ntts::", "").replace("ntts::", "")
        print(f"{name[:84]:84s} {int(r['Calls']):8d} {float(r['TotalDurationNs']) / 1e6:10.3f} "
              f"{float(r['AverageNs']) / 1e3:10.2f} {float(r['MinNs']) / 1e3:9.2f} {float(r['MaxNs']) / 1e3:9.2f} "
              f"{100 * float(r['TotalDurationNs']) / tot:6.2f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
