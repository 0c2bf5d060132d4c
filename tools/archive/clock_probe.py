#!/usr/bin/env python
"""What engine clock does the chip sustain under the prefill GEMM?  Runs the 256 x 256 tile on the prefill gate/up shape for ~2 s per
setting (constant operands, random operands, random operands with the LDS-DMA loads ablated) while a second process samples
`rocm-smi --showclocks --showpower`; prints the samples next to the measured TFLOP/s.  The matrix-core peak quoted everywhere
(2.5 PFLOP/s dense bf16) assumes the 2.4 GHz boost clock.
    python tools/clock_probe.py         (through gpurun)"""
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
from neutts import _hip  # noqa: E402

lib = _hip.load_library()


def sample(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:  # noqa: BLE001
            out.append(str(e))
            break
        sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", t)
        pw = re.search(r"Power \(W\): ([\d.]+)", t) or re.search(r"Graphics Package Power \(W\): ([\d.]+)", t)
        out.append((int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None))


def main():
    M, N, K = 32000, 9728, 896
    fl = 2.0 * M * N * K
    for label, abl in (("constant operands", 0), ("random operands", 32), ("random operands, loads ablated", 34)):
        us = C.c_double()
        lib.ntts_k_gemm_probe(M, N, K, 42, abl, 1, 20, C.byref(us))          # warm
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        th.start()
        t0 = time.time()
        rc = lib.ntts_k_gemm_probe(M, N, K, 42, abl, 1, 4000, C.byref(us))
        dt = time.time() - t0
        stop.set()
        th.join()
        clk = [s[0] for s in out if isinstance(s, tuple) and s[0]]
        pw = [s[1] for s in out if isinstance(s, tuple) and s[1]]
        print(f"{label:32s} rc={rc} {us.value:7.1f} us/launch = {fl / us.value / 1e6:5.0f} TFLOP/s over {dt:.1f} s; sclk samples (MHz): {clk}; power (W): {pw}", flush=True)


if __name__ == "__main__":
    main()
