#!/usr/bin/env python
"""Per-kernel matrix-core utilisation from a rocprofv3 counter pass:
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 -d DIR -o pmc -- <cmd>
    python tools/mfma_util_summary.py DIR
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 256 CUs x 4 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8: on gfx950 the counter
comes back SUMMED over the 8 XCDs (a 550 us kernel reports 11 M "active" cycles), which the gfx94x formula behind rocprofv3's
derived MfmaUtil does not know -- calibrated here on the prefill gate/up GEMM: SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 = 678 GFLOP
against 675 algorithmic, and busy cycles = 16 per v_mfma_f32_16x16x32_bf16 (662 M for 41.2 M instructions).
bf16 FLOP = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 (one MOPS unit = 512 FLOP); achieved rate = FLOP / (GRBM_GUI_ACTIVE / clock) is not
computed here (the counter pass runs at another clock than the bench): utilisation is the clock-free number."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0].replace("void ntts::", "").replace("ntts::", "")
            a = acc[k][row.get("Counter_Name", "?")]
            a[0] += 1
            a[1] += float(row.get("Counter_Value", 0) or 0)
print(f"{'kernel':62s} {'launches':>8s} {'MFMA busy cyc':>14s} {'GUI act (x8)':>14s} {'MfmaUtil %':>10s} {'bf16 GFLOP':>11s}")
rows = []
for k, c in acc.items():
    n = max(v[0] for v in c.values())
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0])[1] / max(n, 1)
    gui = c.get("GRBM_GUI_ACTIVE", [0, 0.0])[1] / max(n, 1)
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [0, 0.0])[1] / max(n, 1)
    util = 100.0 * busy / (gui / 8 * 256 * 4) if gui > 0 else 0.0
    rows.append((busy * n, k, n, busy, gui, util, mops * 512 / 1e9))
for _, k, n, busy, gui, util, gf in sorted(rows, reverse=True)[:24]:
    print(f"{k[:62]:62s} {n:8d} {busy:14.0f} {gui:14.0f} {util:10.1f} {gf:11.2f}")
