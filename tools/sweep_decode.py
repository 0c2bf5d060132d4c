#!/usr/bin/env python
"""Decode-step tuning sweep on one MI355X: for each setting of the engine's env knobs, build an engine at the
BASELINE batch (256 x 500-token prompts), advance to mid-generation (context ~625) and time the hipGraph decode step.

    python tools/sweep_decode.py [--batch 256] [--quick]      (run through gpurun; writes gpurun_out/sweep.jsonl)
Coordinate descent: each knob is swept around the best setting found so far."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as br  # noqa: E402  (model geometry, seeded random weights / prompts: plain data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--prefill", type=int, default=500)
    ap.add_argument("--mid", type=int, default=125)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--vocab", type=int, default=217488)
    ap.add_argument("--knobs", type=str, default="", help="JSON list of [name, [values...]] overriding the default plan")
    ap.add_argument("--base", type=str, default="{}", help="JSON dict of settings every measurement starts from")
    a = ap.parse_args()
    cfg = br.BackboneConfig.neutts_air(a.vocab)
    t0 = time.time()
    w = br.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    inv = br.rope_inv_freq(cfg).numpy()
    B, S = a.batch, a.prefill
    prompts = [br.synthetic_prompt(cfg, i, S) for i in range(B)]
    print(f"[sweep] weights + prompts ready in {time.time() - t0:.1f}s", flush=True)
    out = open(os.path.join(ROOT, "gpurun_out", "sweep.jsonl"), "a")

    def measure(env):
        for k in list(os.environ):
            if k.startswith("NTTS_") and k not in ("NTTS_FORCE_BUILD",):
                del os.environ[k]
        for k, v in env.items():
            os.environ[k] = str(v)
        eng = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                                       intermediate_size=cfg.intermediate_size, num_layers=cfg.num_layers,
                                       num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                       max_context=((S + 250 + 31) // 32) * 32 if S > 500 else 768, max_batch=B, max_prefill_tokens=64 * S), 0)
        eng.load_state_dict(wd, inv_freq=inv)
        samp = _hip.Sampling(max_length=S + 250, min_new_tokens=250, eos_token_id=cfg.vocab_size - 1, do_sample=False)
        for c in range(0, B, 64):
            n = min(64, B - c)
            eng.prefill(prompts[c:c + n], list(range(c, c + n)), [samp] * n)
        eng.sync()
        pf_ms = eng.last_timing()[0]
        eng.decode(a.mid)
        eng.sync()
        best = 1e9
        for _ in range(3):
            eng.decode(a.steps)
            eng.sync()
            best = min(best, eng.last_timing()[1] / a.steps)
        ids = eng.read(0)[0][:8]
        kern = {}
        for k, name in enumerate(_hip.BackboneEngine.KERNELS):
            ms, nb, nl = eng.time_kernel(k, 20)
            kern[name] = round(ms * 1e3, 2)
        eng.close()
        rec = {"env": env, "step_ms": round(best, 4), "last_prefill_chunk_ms": round(pf_ms, 2), "isolated_us": kern, "ids": ids}
        print(json.dumps(rec), flush=True)
        out.write(json.dumps(rec) + "\n")
        out.flush()
        return best

    # the engine keeps few environment knobs; an experiment adds its own temporary one and names it here
    plan = json.loads(a.knobs) if a.knobs else [["NTTS_HEAD_TILE", [2, 4]], ["NTTS_XCD_AFFINE", [0, 7]]]
    cur = json.loads(a.base)
    base = measure(cur)
    for name, values in plan:
        best_v, best_t = None, base
        for v in values:
            t = measure({**cur, name: v})
            if t < best_t:
                best_v, best_t = v, t
        if best_v is not None:
            cur[name] = best_v
            base = best_t
        print(f"[sweep] after {name}: best {cur} -> {base:.4f} ms/step", flush=True)
    print("[sweep] FINAL", json.dumps(cur), f"{base:.4f} ms/step", flush=True)


if __name__ == "__main__":
    main()
