#!/usr/bin/env python
"""Where the GPU's time goes in a rocprofv3 --kernel-trace of bench.py, by PHASE of the hot path rather than by kernel symbol:
prompt pass / decode step / codec / copies -- sum of kernel durations per class, the time with at least one kernel running (GPU idle =
span - that), and the mean number of kernels in flight.  Used to compare the static and the continuous schedule over the same window
(tools/gpu_round.sh profcont): what the ragged scheduler loses is either idle time or extra kernel time in one of the classes.

    python tools/trace_classes.py gpurun_out/profcont [lo hi]      # window = fractions of the trace span, default 0.25 0.75
"""
import csv
import glob
import os
import sys
from collections import defaultdict

# (decode-step kernels are the ones a step graph holds; tile parameters tell the prompt pass's GEMMs from the step's)
DECODE = ("attn_decode_kernel", "qkv_rope_kernel", "add_rmsnorm_row_kernel", "sample_", "embed_norm_meta_kernel", "activate_slots_kernel",
          "gemv_", "step_meta", "bt_update_kernel", "snapshot")
PREFILL = ("attn_prefill", "rope_kv_write", "rope_norm_kv_write", "prefill_init_kernel", "pack_rows_kernel", "gather_rows_kernel", "add_rmsnorm_kernel")
CODEC = ("attn_full", "rownorm_kernel", "groupnorm", "ola_kernel", "codec_embed_kernel", "istft_prep_kernel", "export_codes_kernel", "snake", "conv")


def classify(name):
    n = name.replace("void ntts::", "").replace("ntts::", "")
    if n.startswith("gemm_kernel<"):
        a = [x.strip() for x in n[len("gemm_kernel<"):n.index(">")].split(",")]
        epi, ns = int(a[3]), int(a[4])
        if a[-1] == "true" or epi in (4, 5, 7):          # fp16 operands / codec-only epilogues
            return "codec"
        if epi in (2, 3) or (epi == 1 and ns == 3):       # split-K slabs, argmax head, the step's gate/up tile (3-slot ring)
            return "decode"
        return "prefill"
    for keys, c in ((DECODE, "decode"), (PREFILL, "prefill"), (CODEC, "codec")):
        if any(k in n for k in keys):
            return c
    if "copyBuffer" in n or "fillBuffer" in n:
        return "copies"
    return "other"


def main():
    d = sys.argv[1]
    lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.25, 0.75)
    iv = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                try:
                    iv.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row.get("Kernel_Name", "")))
                except (KeyError, ValueError):
                    pass
    if not iv:
        print("no kernel trace under", d)
        return 1
    iv.sort()
    t0, t1 = iv[0][0], max(e for _, e, _ in iv)
    w0, w1 = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
    cls = defaultdict(lambda: [0, 0.0])
    names = defaultdict(lambda: [0, 0.0])
    union, cur_s, cur_e, busy = 0.0, None, None, 0.0
    for s, e, n in iv:
        if e <= w0 or s >= w1:
            continue
        s2, e2 = max(s, w0), min(e, w1)
        c = classify(n)
        cls[c][0] += 1
        cls[c][1] += e2 - s2
        k = c + ": " + n.replace("void ntts::", "").replace("ntts::", "").split("(")[0][:70]
        names[k][0] += 1
        names[k][1] += e2 - s2
        busy += e2 - s2
        if cur_e is None or s2 > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s2, e2
        else:
            cur_e = max(cur_e, e2)
    if cur_e is not None:
        union += cur_e - cur_s
    span = w1 - w0
    print(f"# window {lo:.2f}..{hi:.2f} of a {1e-6 * (t1 - t0):.1f} ms trace = {1e-6 * span:.1f} ms; at least one kernel running {100 * union / span:.2f} % of it "
          f"(idle {1e-6 * (span - union):.1f} ms); mean kernels in flight while busy {busy / max(union, 1):.2f}")
    print(f"{'class':10s} {'dispatches':>10s} {'sum of durations ms':>20s} {'share of the sum':>18s} {'per ms of window':>18s}")
    for c in sorted(cls, key=lambda c: -cls[c][1]):
        print(f"{c:10s} {cls[c][0]:10d} {1e-6 * cls[c][1]:20.1f} {100 * cls[c][1] / busy:17.2f}% {cls[c][1] / span:18.3f}")
    print("# top symbols")
    for k in sorted(names, key=lambda k: -names[k][1])[:24]:
        print(f"{k:84s} {names[k][0]:9d} {1e-6 * names[k][1]:10.1f} ms {1e-3 * names[k][1] / max(1, names[k][0]):9.2f} us")
    return 0


if __name__ == "__main__":
    sys.exit(main())
