#!/usr/bin/env python
"""Summarise a rocprofv3 counter-collection run (csv) per kernel: dispatches, mean counter value per dispatch.

    python tools/pmc_summary.py gpurun_out/pmc [--x2 FETCH_SIZE]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-byte requests of a wide
coalesced streaming read at 64 B, so it reports half the bytes (MI355X_MICROARCH.md, section HBM) -- the `x2` column
applies that correction and is what bench.py's roofline.traffic quotes."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print("no *counter_collection.csv under", d)
        return 1
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))   # kernel -> counter -> [n, sum]
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                k = k.split("(")[0]
                k = k.replace("void ntts::", "").replace("ntts::", "")
                c = row.get("Counter_Name", "?")
                v = float(row.get("Counter_Value", 0) or 0)
                a = acc[k][c]
                a[0] += 1
                a[1] += v
    print(f"{'kernel':70s} {'counter':14s} {'dispatches':>10s} {'mean/dispatch':>16s} {'MB (KiB unit)':>14s} {'MB x2 (gfx950 fetch)':>20s}")
    for k in sorted(acc, key=lambda kk: -max(v[1] for v in acc[kk].values())):
        for c, (n, s) in sorted(acc[k].items()):
            mean = s / max(n, 1)
            mb = mean * 1024 / 1e6
            x2 = mb * 2 if c == "FETCH_SIZE" else mb
            print(f"{k[:70]:70s} {c:14s} {n:10d} {mean:16.1f} {mb:14.3f} {x2:20.3f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
