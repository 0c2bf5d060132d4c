#!/usr/bin/env python
"""Where does the one-off ~40 ms host stall of the second full benchmark step come from?  Replays the static step (4 prompt passes,
249 decode steps, poll, 256 releases) a few times with per-call host timings, under variants selected by PROBE_VARIANT:
  base       as bench.py does it (backbone only)
  norelease  slots released through one engine rebuild instead of 256 release calls (not possible: so: release only, no poll)
  warm       BackboneEngine.warm_up(249) first
  sleep      1 s of idle time between the steps
    python tools/stall_probe.py        (through gpurun)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as syn  # noqa: E402


def main():
    variant = os.environ.get("PROBE_VARIANT", "base")
    B, S, N = 256, 500, 250
    cfg = syn.BackboneConfig.neutts_air()
    w = syn.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    eng = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                   num_layers=cfg.num_layers, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                   max_context=768, max_batch=B, max_prefill_tokens=64 * S), 0)
    eng.load_state_dict(wd, inv_freq=syn.rope_inv_freq(cfg).numpy())
    prompts = [syn.synthetic_prompt(cfg, i, S) for i in range(B)]
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=cfg.vocab_size - 1, do_sample=False)
    if variant == "warm":
        eng.warm_up(N - 1)
    for step in range(4):
        t = {}
        each = []
        for c in range(0, B, 64):
            t0 = time.time()
            eng.prefill(prompts[c:c + 64], list(range(c, c + 64)), [samp] * 64)
            each.append(round((time.time() - t0) * 1e3, 1))
        t["prefill_calls_ms"] = each
        t0 = time.time(); eng.decode(N - 1); t["decode_enqueue_ms"] = round((time.time() - t0) * 1e3, 1)
        t0 = time.time(); eng.poll(); t["poll_ms"] = round((time.time() - t0) * 1e3, 1)
        t0 = time.time()
        for s in range(B):
            eng.release(s)
        t["release_ms"] = round((time.time() - t0) * 1e3, 1)
        if variant == "sync":
            t0 = time.time(); eng.sync(); torch.cuda.synchronize(); t["sync_ms"] = round((time.time() - t0) * 1e3, 1)
        if variant == "sleep":
            time.sleep(1.0)
        print(json.dumps({"variant": variant, "step": step, **t}), flush=True)


if __name__ == "__main__":
    main()
