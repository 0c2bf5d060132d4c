#!/usr/bin/env python
"""Tile choice for decode-step GEMMs that carry 512 prompt rows beside the 256 decode rows (M = 768): every tile family of
ntts_k_gemm_probe on the four layer GEMM shapes, HBM-cold tile-major weights of random data, next to M = 256 on today's tiles.
    python tools/probe_tiles_m768.py   (through gpurun)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
from neutts import _hip  # noqa: E402

lib = _hip.load_library()
NAMES = {12: "64x64 4w NS4", 11: "64x64 4w NS3", 23: "64x128 4w NS3", 24: "64x128 8w NS3", 25: "128x64 8w NS3", 26: "128x64 4w NS3", 30: "128x128 4w NS2",
         31: "128x128 4w NS3", 54: "128x128 8w NS3", 40: "256x128 8w NS2", 43: "256x128 8w NS3", 41: "128x256 8w NS2", 44: "256x128 4w NS2",
         42: "256x256 16w NS2", 53: "256x128 16w NS2", 50: "256x64 4w NS3", 51: "256x64 8w NS3"}


def probe(M, N, K, cfg, ks=1, iters=96):
    us = C.c_double()
    rc = lib.ntts_k_gemm_probe(M, N, K, ks * 1000 + cfg, 16 | 32, 24, iters, C.byref(us))
    return us.value if rc == 0 else float("nan")


def main():
    shapes = {"qkv": (1152, 896), "o_proj": (896, 896), "gate_up": (9728, 896), "down": (896, 4864)}
    for name, (N, K) in shapes.items():
        for M in (256, 768):
            fl = 2.0 * M * N * K
            print(f"== {name} M={M} N={N} K={K} ({fl / 1e9:.1f} GFLOP)")
            for cfg in NAMES:
                splits = (1,) if name == "gate_up" else (1, 2, 4) if K <= 1024 else (1, 2, 4, 6, 8, 12)
                row = []
                for ks in splits:
                    t = probe(M, N, K, cfg, ks)
                    row.append(f"ks{ks} {t:6.1f}")
                print(f"  {NAMES[cfg]:18s} " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
