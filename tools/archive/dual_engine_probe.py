#!/usr/bin/env python
"""Does splitting the decode batch over several engines (one HIP stream each, same GPU) hide the launch gaps?

Each engine owns batch/n slots and its own hipGraph; the n step chains are independent, so the GPU can run one chain's
attention (HBM-bound) under another chain's skinny GEMMs (latency-bound).  Cost: the weights are streamed n times per step.

    python tools/dual_engine_probe.py [--batch 256] [--ways 1,2,4]     (through gpurun)
Prints one JSON line per setting: wall ms per step of the whole batch, codec-tokens/s of the decode loop alone."""
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
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--prefill", type=int, default=500)
    ap.add_argument("--mid", type=int, default=125)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--vocab", type=int, default=217488)
    ap.add_argument("--ways", type=str, default="1,2,4")
    a = ap.parse_args()
    cfg = br.BackboneConfig.neutts_air(a.vocab)
    w = br.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    inv = br.rope_inv_freq(cfg).numpy()
    S = a.prefill
    prompts = [br.synthetic_prompt(cfg, i, S) for i in range(a.batch)]
    samp = _hip.Sampling(max_length=S + 250, min_new_tokens=250, eos_token_id=cfg.vocab_size - 1, do_sample=False)
    for ways in [int(x) for x in a.ways.split(",")]:
        b = a.batch // ways
        engs = []
        for k in range(ways):
            e = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                                         intermediate_size=cfg.intermediate_size, num_layers=cfg.num_layers,
                                         num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                         max_context=768, max_batch=b, max_prefill_tokens=64 * S), 0)
            e.load_state_dict(wd, inv_freq=inv)
            for c in range(0, b, 64):
                n = min(64, b - c)
                e.prefill(prompts[k * b + c:k * b + c + n], list(range(c, c + n)), [samp] * n)
            e.decode(a.mid)
            e.sync()
            engs.append(e)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _s in range(a.steps):          # interleave the enqueues so that no chain runs ahead of the others
                for e in engs:
                    e.decode(1)
            for e in engs:
                e.sync()
            best = min(best, (time.perf_counter() - t0) * 1e3 / a.steps)
        own = [e.last_timing()[1] for e in engs]
        ids = engs[0].read(0)[0][:6]
        print(json.dumps({"ways": ways, "batch_per_engine": b, "wall_ms_per_step": round(best, 4),
                          "decode_tokens_per_s": round(a.batch / best * 1e3), "last_step_ms_per_engine": [round(x, 4) for x in own],
                          "ids": ids}), flush=True)
        for e in engs:
            e.close()


if __name__ == "__main__":
    main()
