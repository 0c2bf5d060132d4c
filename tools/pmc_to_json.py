#!/usr/bin/env python
"""Turn the rocprofv3 counter-collection csv of the pmc leg into the small record bench.py quotes as roofline.traffic.

    python tools/pmc_to_json.py gpurun_out/pmc_FETCH_SIZE [gpurun_out/pmc_WRITE_SIZE] [--prefill=605 --decode=40] > profiles/r02_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 counts the 128-B requests of wide coalesced reads at
64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE is left uncorrected (uncalibrated there)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = row.get("Kernel_Name", "?").split("(")[0].replace("void ntts::", "").replace("ntts::", "")
                a = acc[k]
                a[0] += 1
                a[1] += float(row.get("Counter_Value", 0) or 0)
    return {k: (n, s / max(n, 1)) for k, (n, s) in acc.items()}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = {a.split("=")[0][2:]: int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--") and "=" in a}
    S, N, B = opt.get("prefill", 605), opt.get("decode", 40), opt.get("batch", 256)
    lo, hi = S, S + N - 2          # positions of the N - 1 decode steps
    out = {"source": "rocprofv3 --kernel-trace --pmc <counter> (one pass per counter, hipGraph replay off), "
                     f"bench.py --prefill {S} --decode {N} --batch {B}: decode contexts {lo}..{hi}, mean {(lo + hi) / 2:.1f}"
                     " (the roofline leg runs at mean 625)",
           "decode_context_mean": (lo + hi) / 2, "kernels": {}}
    fetch = per_kernel(args[0], "FETCH_SIZE")
    write = per_kernel(args[1], "WRITE_SIZE") if len(args) > 1 else {}
    for k, (n, mean_kib) in fetch.items():
        rec = {"dispatches": n, "fetch_bytes_per_launch": mean_kib * 1024 * 2, "fetch_kib_raw": mean_kib}
        if k in write:
            rec["write_bytes_per_launch_uncorrected"] = write[k][1] * 1024
        out["kernels"][k] = rec
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
