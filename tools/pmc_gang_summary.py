#!/usr/bin/env python
"""Per-kernel HBM counter of a GANG pass (tools/gpu_round.sh pmcgang): mean FETCH_SIZE / WRITE_SIZE per dispatch of every decode-step kernel
while four engines' chains are enqueued side by side, and -- from the kernel trace of the same pass -- how much the dispatches of DIFFERENT
streams overlapped in time (rocprofv3's counter collection may serialise dispatches: then the counters describe the chains one kernel at a time).

    python tools/pmc_gang_summary.py gpurun_out/pmcg_FETCH_SIZE FETCH_SIZE
FETCH_SIZE is in KiB and doubled here (gfx950 counts the 128-B requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM); WRITE_SIZE uncorrected."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, ctr = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != ctr:
                    continue
                k = row.get("Kernel_Name", "?").split("(")[0].replace("void ntts::", "").replace("ntts::", "")
                acc[k][0] += 1
                acc[k][1] += float(row.get("Counter_Value", 0) or 0)
    mul = 2.0 if ctr == "FETCH_SIZE" else 1.0
    print(f"# {ctr} per dispatch, gang of four 256-slot engines, graph replay off ({'x2 gfx950 correction' if mul == 2 else 'uncorrected'})")
    print(f"{'kernel':72s} {'dispatches':>10s} {'MB / dispatch':>14s}")
    for k in sorted(acc, key=lambda kk: -acc[kk][1]):
        n, s = acc[k]
        print(f"{k[:72]:72s} {n:10d} {s / max(n, 1) * 1024 * mul / 1e6:14.3f}")
    # overlap of dispatches from different queues, from the kernel trace
    iv = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                try:
                    iv.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row.get("Queue_Id", row.get("Stream_Id", "?")), row.get("Kernel_Name", "")))
                except (KeyError, ValueError):
                    pass
    iv = [x for x in iv if "attn_decode" in x[3] or "gemm_kernel" in x[3] or "qkv_rope" in x[3] or "add_rmsnorm" in x[3]]
    iv.sort()
    busy = sum(e - s for s, e, _, _ in iv)
    union, cur_s, cur_e = 0, None, None
    for s, e, _, _ in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        union += cur_e - cur_s
    queues = len({q for _, _, q, _ in iv})
    if union:
        print(f"# kernel trace of the same pass: {len(iv)} decode-step dispatches on {queues} queues; sum of their durations / time with at least one running = {busy / union:.2f} "
              f"(1.0 = the profiler ran them one at a time, 4.0 = four chains fully side by side)")


if __name__ == "__main__":
    main()
