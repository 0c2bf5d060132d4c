#!/usr/bin/env python
"""Prompt-pass cost against its size on one MI355X: for n prompts of --prefill tokens in one ntts_backbone_prefill call, the time of the
pass on one engine alone and of four engines' passes enqueued at once on their lanes (the admission wave of continuous mode,
EngineGang.generate(admit="wave")), in us per prompt token.  What a small admission group costs over a 64-prompt chunk.

    python tools/probe_prefill_size.py          (run through gpurun; prints one JSON line per size)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as br  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prefill", type=int, default=500)
    ap.add_argument("--sizes", type=int, nargs="*", default=[4, 8, 12, 16, 24, 32, 48, 64])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--spread", type=float, default=0.0, help="prompt lengths uniform in prefill x (1 -+ spread) instead of all equal (what raggedness costs)")
    a = ap.parse_args()
    cfg = br.BackboneConfig.neutts_air(217488)
    w = br.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    S, B = a.prefill, 256
    import numpy as np
    lens = np.random.default_rng(5).integers(int(S * (1 - a.spread)), int(S * (1 + a.spread)) + 1, size=64) if a.spread > 0 else np.full(64, S)
    prompts = [br.synthetic_prompt(cfg, i, int(lens[i])) for i in range(64)]
    e0 = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                  num_layers=cfg.num_layers, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                  max_context=1024, max_batch=B, max_prefill_tokens=int(64 * S * (1 + a.spread)), weight_dtype="bf16"), 0)
    e0.load_state_dict(wd, inv_freq=br.rope_inv_freq(cfg).numpy())
    gang = _hip.EngineGang(e0, 4)
    samp = _hip.Sampling(max_length=int(S * (1 + a.spread)) + 250, min_new_tokens=250, eos_token_id=cfg.vocab_size - 1, do_sample=False)

    def run(engs, n):
        best = 1e9
        for _ in range(a.reps + 1):
            gang.sync()
            t0 = time.perf_counter()
            for e in engs:
                e.prefill(prompts[:n], list(range(n)), [samp] * n)
            gang.sync()
            best = min(best, time.perf_counter() - t0)
            for e in engs:
                e.release_many(list(range(n)))
        return best * 1e3

    for n in a.sizes:
        one, four = run(gang.engines[:1], n), run(gang.engines, n)
        T = int(lens[:n].sum())
        print(json.dumps({"prompts": n, "tokens": T, "spread": a.spread, "one_engine_ms": round(one, 3), "one_engine_us_per_token": round(one * 1e3 / T, 3),
                          "four_engines_at_once_ms": round(four, 3), "four_us_per_token": round(four * 1e3 / (4 * T), 3)}), flush=True)
    gang.close()


if __name__ == "__main__":
    main()
