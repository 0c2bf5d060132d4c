#!/usr/bin/env python
"""Reference-encoding path (encode_reference) on one MI355X at NeuCodec geometry: GPU time per clip length,
algorithmic FLOPs / achieved TFLOP/s, and -- with --cpu -- the oracle restatement timed on the host cores for one short clip.

    python tools/encode_probe.py [--secs 3,10,30] [--cpu]       (through gpurun; prints one JSON line per clip)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import synthetic as syn  # noqa: E402
from neutts import _hip  # noqa: E402


def flops(cfg, T):
    """Algorithmic multiply-add FLOPs of one encode call producing T frames (GEMM / conv terms only)."""
    H, I, L = cfg.sem_hidden, cfg.sem_ffn, T * cfg.hop
    per_layer = 2 * T * (4 * H * I + 4 * H * H + 2 * H * H + H * H) + 4 * T * T * H      # two FFNs, qkv + out, pw1 (2H) + pw2, attention
    sem = cfg.sem_layers * per_layer + 2 * T * 160 * H + 4 * 2 * T * 3 * H * H
    ac, length, ch = 2.0 * L * 7 * cfg.ac_hidden, L, cfg.ac_hidden
    for r in cfg.ratios:
        ac += 3 * 2.0 * length * (7 * ch * ch + ch * ch) + 2.0 * (length // r) * 2 * r * ch * 2 * ch
        length //= r
        ch *= 2
    ac += 2.0 * T * 3 * ch * cfg.codec_hidden
    tail = 2.0 * T * cfg.cat_dim * (cfg.cat_dim + len(cfg.levels))
    return sem + ac + tail


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--secs", default="3,10,30")
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    cfg = syn.EncoderConfig.neucodec()
    t0 = time.time()
    w = syn.make_encoder_weights(cfg, 7)
    d = cfg.to_dict()
    d["max_samples"] = 31 * 16000
    eng = _hip.EncoderEngine(d, 0)
    eng.load_state_dict({k: v.numpy() for k, v in w.items()})
    print(f"[encode_probe] weights + engine ready in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
    for secs in [float(x) for x in a.secs.split(",")]:
        wav = syn.synthetic_speech(int(secs * 16000), 3)
        eng.encode(wav)
        best, wall = 1e9, 1e9
        for _ in range(3):
            t1 = time.perf_counter()
            codes = eng.encode(wav)
            wall = min(wall, (time.perf_counter() - t1) * 1e3)
            best = min(best, eng.last_timing())
        fl = flops(cfg, codes.size)
        print(json.dumps({"clip_s": secs, "codes": int(codes.size), "gpu_ms": round(best, 2), "wall_ms": round(wall, 2),
                          "x_real_time": round(secs * 1e3 / best, 1), "alg_gflop": round(fl / 1e9, 1),
                          "achieved_tflops_fp32": round(fl / best / 1e9, 2)}), flush=True)
    if a.cpu:
        import torch
        from oracle import encoder_ref as er          # tools/: test infrastructure may time the oracle
        wav = syn.synthetic_speech(3 * 16000, 3)
        t1 = time.perf_counter()
        c = er.encode(cfg, w, wav)
        print(json.dumps({"cpu_oracle_clip_s": 3.0, "cpu_ms": round((time.perf_counter() - t1) * 1e3, 1), "threads": torch.get_num_threads(),
                          "codes": int(c.size)}), flush=True)


if __name__ == "__main__":
    main()
