#!/usr/bin/env python
"""Probe (one MI355X, through gpurun): do TWO decode chains -- two 256-row engines, each replaying its own step graph on its own
stream -- hide each other's launch gaps and first round trips?  Times `steps` decode steps at context ~625 on one engine alone, then
the same number on each of two engines enqueued alternately in groups of `--group` replays from one thread.

    python tools/probe_two_chains.py [--group 1 2 4 8]      (writes gpurun_out/two_chains.jsonl)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as br  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--prefill", type=int, default=500)
    ap.add_argument("--mid", type=int, default=60)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--vocab", type=int, default=217488)
    ap.add_argument("--group", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--engines", type=int, default=2)
    a = ap.parse_args()
    cfg = br.BackboneConfig.neutts_air(a.vocab)
    w = br.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    inv = br.rope_inv_freq(cfg).numpy()
    B, S = a.batch, a.prefill
    prompts = [br.synthetic_prompt(cfg, i, S) for i in range(B)]
    e0 = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                  num_layers=cfg.num_layers, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                  max_context=768, max_batch=B, max_prefill_tokens=64 * S), 0)
    e0.load_state_dict(wd, inv_freq=inv)
    engs = [e0] + [e0.twin() for _ in range(a.engines - 1)]
    samp = _hip.Sampling(max_length=S + 250, min_new_tokens=250, eos_token_id=cfg.vocab_size - 1, do_sample=False)
    for e in engs:
        for c in range(0, B, 64):
            n = min(64, B - c)
            e.prefill(prompts[c:c + n], list(range(c, c + n)), [samp] * n)
        e.decode(a.mid)
        e.sync()
    out = open(os.path.join(ROOT, "gpurun_out", "two_chains.jsonl"), "a")

    def wall(fn):
        for e in engs:
            e.sync()
        t0 = time.perf_counter()
        fn()
        th = time.perf_counter()
        for e in engs:
            e.sync()
        return (time.perf_counter() - t0) * 1e3, (th - t0) * 1e3

    K = a.steps
    # every leg runs K steps per engine at (nearly) the same context: K * legs << 250 - mid keeps the slots running
    ms, host = wall(lambda: e0.decode(K))
    rec = {"leg": "one engine alone", "steps": K, "ms_per_step": round(ms / K, 4), "host_ms": round(host, 2)}
    print(json.dumps(rec), flush=True)
    out.write(json.dumps(rec) + "\n")
    ms, host = wall(lambda: [e.decode(K) for e in engs[1:]])
    rec = {"leg": "other engines, whole bursts back to back", "steps": K, "ms_per_step_per_engine": round(ms / K / max(1, len(engs) - 1), 4)}
    print(json.dumps(rec), flush=True)
    out.write(json.dumps(rec) + "\n")
    for g in a.group:
        def run():
            for _ in range(0, K, g):
                for e in engs:
                    e.decode(g)
        ms, host = wall(run)
        rec = {"leg": f"{len(engs)} engines alternating, groups of {g}", "steps_each": K, "ms_total": round(ms, 3),
               "ms_per_256_row_step": round(ms / K / len(engs), 4), "host_ms": round(host, 2)}
        print(json.dumps(rec), flush=True)
        out.write(json.dumps(rec) + "\n")
    ms, host = wall(lambda: e0.decode(K))
    rec = {"leg": "one engine alone, again (longer context)", "steps": K, "ms_per_step": round(ms / K, 4)}
    print(json.dumps(rec), flush=True)
    out.write(json.dumps(rec) + "\n")
    ids = [e.read(0)[0][:6] for e in engs]
    print(json.dumps({"ids_slot0": ids}), flush=True)


if __name__ == "__main__":
    main()
