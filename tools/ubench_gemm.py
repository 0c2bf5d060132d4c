#!/usr/bin/env python
"""GEMM tile micro-benchmark / ablation on one MI355X (through ntts_k_gemm_probe).
    python tools/ubench_gemm.py          -> gpurun_out/ubench_gemm.txt"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
from neutts import _hip  # noqa: E402

lib = _hip.load_library()
SHAPES = {"qkv": (256, 1152, 896), "o": (256, 896, 896), "gate_up": (256, 9728, 896), "down": (256, 896, 4864)}
CONFIGS = {10: "64x64 4w NS2", 11: "64x64 4w NS3", 12: "64x64 4w NS4", 13: "64x64 4w NS6", 20: "32x64 2w NS4",
           21: "64x64 2w NS4", 22: "64x64 1w NS4", 23: "64x128 4w NS3", 24: "64x128 8w NS3", 25: "128x64 8w NS3",
           26: "128x64 4w NS3", 30: "128x128 4w NS2", 31: "128x128 4w NS3", 40: "256x128 8w NS2", 41: "128x256 8w NS2",
           42: "256x256 16w NS2", 43: "256x128 8w NS3", 44: "256x128 4w NS2", 45: "256x256 8w NS2", 46: "256x256 16w 4xK32", 47: "256x256 16w 3xK32", 48: "256x256 8w 4xK32", 49: "128x128 4w 4xK32", 60: "256x256 16w persist", 50: "256x64 4w NS3", 51: "256x64 8w NS3",
           52: "256x64 8w NS2", 53: "256x128 16w NS2", 54: "128x128 8w NS3", 55: "320x256 16w NS2", 56: "384x256 16w NS2", 57: "320x256 8w NS2", 58: "384x256 8w NS2"}


def probe(M, N, K, cfg, abl, copies, iters=200):
    us = C.c_double()
    rc = lib.ntts_k_gemm_probe(M, N, K, cfg, abl, copies, iters, C.byref(us))
    return us.value if rc == 0 else float("nan")


PREFILL = {"pf_qkv": (32000, 1152, 896), "pf_gate_up": (32000, 9728, 896), "pf_down": (32000, 896, 4864),
           "codec_fc1": (65536, 4096, 1024)}


def prefill():
    for name, (M, N, K) in PREFILL.items():
        fl = 2.0 * M * N * K
        print(f"== {name}  M={M} N={N} K={K}  {fl / 1e12:.2f} TFLOP")
        print(f"{'config':18s} {'us':>9s} {'TF/s':>7s} | ablations us: {'noMFMA':>9s} {'noDMA':>9s} {'neither':>9s} {'noStore':>9s}")
        for cfg in (30, 42):
            t = probe(M, N, K, cfg, 0, 1, iters=5)
            ab = [probe(M, N, K, cfg, a, 1, iters=5) for a in (1, 2, 3, 4)]
            print(f"{CONFIGS[cfg]:18s} {t:9.1f} {fl / t / 1e6:7.0f} | {'':14s} {ab[0]:9.1f} {ab[1]:9.1f} {ab[2]:9.1f} {ab[3]:9.1f}", flush=True)


def prefill_big():
    """round 5: tiles with MORE rows than 256 x 256 (fewer LDS-DMA bytes per FLOP: the prompt-pass GEMMs run at the per-CU DMA rate,
    profiles/r01e_ubench_prefill_ring_variants.txt ablations), random operands, sustained."""
    for name, (M, N, K) in PREFILL.items():
        fl = 2.0 * M * N * K
        print(f"== {name}  M={M} N={N} K={K}  {fl / 1e12:.2f} TFLOP, random operands")
        print(f"{'config':18s} {'us':>9s} {'TF/s':>7s} | ablations us: {'noMFMA':>9s} {'noDMA':>9s} {'noStore':>9s}")
        for cfg in (42, 45, 55, 56, 57, 58, 42):
            t = probe(M, N, K, cfg, 32, 1, iters=60)
            ab = [probe(M, N, K, cfg, 32 | a_, 1, iters=20) for a_ in (1, 2, 4)]
            print(f"{CONFIGS[cfg]:18s} {t:9.1f} {fl / t / 1e6:7.0f} | {'':14s} {ab[0]:9.1f} {ab[1]:9.1f} {ab[2]:9.1f}", flush=True)


def x_layout():
    """round 5: does the LAYOUT of the activation operand bound the LDS-DMA rate of the big GEMMs?  X row-major (a block's 256 x 64 tile = 256 lines of
    128 B, 2 K bytes apart) vs X k-tile-major (the tile = 32 KB contiguous), W row-major / tile-major, random operands; noMFMA = the DMA stream alone."""
    for name, (M, N, K) in PREFILL.items():
        fl = 2.0 * M * N * K
        print(f"== {name}  M={M} N={N} K={K}  {fl / 1e12:.2f} TFLOP, 256x256 16w NS2, random operands")
        for label, bits in (("X rows, W rows", 0), ("X k-tile-major, W rows", 64), ("X rows, W tile-major", 16), ("X k-tile-major, W tile-major", 80), ("X rows, W rows", 0)):
            t = probe(M, N, K, 42, 32 | bits, 1, iters=60)
            t1 = probe(M, N, K, 42, 32 | bits | 1, 1, iters=20)
            print(f"  {label:30s} {t:8.1f} us {fl / t / 1e6:6.0f} TF/s   noMFMA {t1:8.1f} us", flush=True)


def power():
    """The same prefill / codec GEMMs on constant-filled operands (hipMemset) and on operands with the bit statistics of real
    data: how much of the distance to the 2.5 PFLOP/s matrix-core peak is the clock the chip sustains under real toggling."""
    for name, (M, N, K) in PREFILL.items():
        fl = 2.0 * M * N * K
        print(f"== {name}  M={M} N={N} K={K}  {fl / 1e12:.2f} TFLOP   (256x256 16w NS2)")
        for label, extra in (("constant operands", 0), ("random operands", 32)):
            t = [probe(M, N, K, 42, extra, 1, iters=it) for it in (5, 50)]           # 5 launches: cold chip; 50: sustained
            t2 = probe(M, N, K, 42, extra | 2, 1, iters=20)                         # no LDS-DMA: matrix cores + LDS reads + epilogue
            print(f"  {label:18s} {t[0]:8.1f} us ({fl / t[0] / 1e6:5.0f} TF/s) over 5 launches, {t[1]:8.1f} us ({fl / t[1] / 1e6:5.0f} TF/s) over 50;"
                  f"  loads ablated: {t2:8.1f} us ({fl / t2 / 1e6:5.0f} TF/s)", flush=True)


def power_tiles():
    """Ranking of the big-M tile variants at the board's power limit (random operands, sustained): with real data the prefill GEMMs
    run against the 1400 W cap (tools/clock_probe.py), so the variant that moves the fewest bytes per FLOP may win where it lost on
    constant operands."""
    for name, (M, N, K) in PREFILL.items():
        fl = 2.0 * M * N * K
        print(f"== {name}  M={M} N={N} K={K}  {fl / 1e12:.2f} TFLOP, random operands, 300 launches each (constant operands, 50 launches, in brackets)")
        for cfg in (30, 40, 41, 42, 43, 44, 45, 46, 53):
            t = probe(M, N, K, cfg, 32, 1, iters=300)
            t0 = probe(M, N, K, cfg, 0, 1, iters=50)
            print(f"  {CONFIGS[cfg]:20s} {t:8.1f} us {fl / t / 1e6:6.0f} TF/s   [{t0:8.1f} us {fl / t0 / 1e6:6.0f} TF/s]", flush=True)


def mall():
    """Does touching the weights one kernel ahead (Infinity-Cache resident instead of HBM-cold) speed the decode GEMMs up?"""
    for name, (M, N, K) in SHAPES.items():
        wbytes = N * K * 2
        cold = max(2, int(400e6 // wbytes))
        pf_alone = probe(M, N, K, 0, 8, cold)
        print(f"== {name}: prefetch kernel alone {pf_alone:.2f} us (incl. one empty launch)")
        for cfg in (11, 12):
            c0 = probe(M, N, K, cfg, 0, cold)
            c1 = probe(M, N, K, cfg, 8, cold)
            w0 = probe(M, N, K, cfg, 0, 1)
            print(f"  {CONFIGS[cfg]:16s} cold {c0:7.2f}  warm {w0:7.2f}  [prefetch(next) + gemm] {c1:7.2f}  -> gemm on prefetched ~ {c1 - pf_alone + 2.2:7.2f}", flush=True)


def tall():
    """decode shapes with tiles that hold all 256 rows (W passes through LDS once)"""
    for name, (M, N, K) in SHAPES.items():
        wbytes = N * K * 2
        cold = max(2, int(400e6 // wbytes))
        print(f"== {name}  M={M} N={N} K={K}")
        for cfg in (11, 12, 50, 51, 52, 53, 54):
            print(f"  {CONFIGS[cfg]:18s} warm {probe(M, N, K, cfg, 0, 1):7.2f}  cold {probe(M, N, K, cfg, 0, cold):7.2f}", flush=True)


def head():
    M, N, K = 256, 217472, 896
    print(f"== lm_head M={M} N={N} K={K} W=390 MB (always HBM-cold); row-major W vs tile-major W (abl bit 16)")
    for cfg in (30, 42, 12, 46, 47, 43, 53):
        t = probe(M, N, K, cfg, 0, 1, iters=20)
        t2 = probe(M, N, K, cfg, 16, 1, iters=20)
        print(f"  {CONFIGS[cfg]:18s} {t:8.1f} us  {N * K * 2 / t / 1e6:6.2f} TB/s | tile-major {t2:8.1f} us  {N * K * 2 / t2 / 1e6:6.2f} TB/s", flush=True)


def layout():
    """decode GEMMs, HBM-cold weights: row-major vs tile-major weight addressing"""
    for name, (M, N, K) in SHAPES.items():
        wbytes = N * K * 2
        cold = max(2, int(400e6 // wbytes))
        print(f"== {name}  M={M} N={N} K={K}  cold copies={cold}")
        for cfg in (11, 12, 54):
            c0 = probe(M, N, K, cfg, 0, cold)
            c1 = probe(M, N, K, cfg, 16, cold)
            print(f"  {CONFIGS[cfg]:18s} row-major {c0:7.2f}  tile-major {c1:7.2f}", flush=True)


def main():
    if "--head" in sys.argv:
        return head()
    if "--layout" in sys.argv:
        head()
        return layout()
    if "--tall" in sys.argv:
        return tall()
    if "--x-layout" in sys.argv:
        return x_layout()
    if "--prefill-big" in sys.argv:
        return prefill_big()
    if "--prefill" in sys.argv:
        return prefill()
    if "--mall" in sys.argv:
        return mall()
    if "--power" in sys.argv:
        return power()
    if "--power-tiles" in sys.argv:
        return power_tiles()
    print(f"empty kernel: {probe(64, 64, 64, 0, 0, 1):.2f} us/launch")
    for name, (M, N, K) in SHAPES.items():
        wbytes = N * K * 2
        cold = max(2, int(400e6 // wbytes))
        print(f"== {name}  M={M} N={N} K={K}  W={wbytes / 1e6:.2f} MB  cold copies={cold}")
        print(f"{'config':18s} {'warm':>8s} {'cold':>8s} | cold ablations: {'noMFMA':>8s} {'noDMA':>8s} {'neither':>8s} {'noStore':>8s}")
        for cfg, label in CONFIGS.items():
            if cfg >= 40:
                continue
            warm = probe(M, N, K, cfg, 0, 1)
            c0 = probe(M, N, K, cfg, 0, cold)
            ab = [probe(M, N, K, cfg, a, cold) for a in (1, 2, 3, 4)]
            print(f"{label:18s} {warm:8.2f} {c0:8.2f} | {'':16s} {ab[0]:8.2f} {ab[1]:8.2f} {ab[2]:8.2f} {ab[3]:8.2f}", flush=True)


if __name__ == "__main__":
    main()
