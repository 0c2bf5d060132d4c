#!/usr/bin/env python
"""Do concurrent half-batches beat one full batch?  N engines (own stream each) x 256/N decode slots, decode enqueued
back to back on all engines, wall time per decode step of the whole 256-sequence batch.
    python tools/multi_engine_probe.py      (through gpurun)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as br  # noqa: E402  (model geometry, seeded random weights / prompts: plain data)


def main():
    cfg = br.BackboneConfig.neutts_air()
    w = br.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    inv = br.rope_inv_freq(cfg).numpy()
    S, total = 500, 256
    samp = _hip.Sampling(max_length=S + 250, min_new_tokens=250, eos_token_id=cfg.vocab_size - 1, do_sample=False)
    for n_eng in (1, 2, 4):
        B = total // n_eng
        engs = []
        for k in range(n_eng):
            e = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                                         intermediate_size=cfg.intermediate_size, num_layers=cfg.num_layers,
                                         num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                         max_context=768, max_batch=B, max_prefill_tokens=64 * S), 0)
            e.load_state_dict(wd, inv_freq=inv)
            prompts = [br.synthetic_prompt(cfg, k * B + i, S) for i in range(B)]
            for c in range(0, B, 64):
                n = min(64, B - c)
                e.prefill(prompts[c:c + n], list(range(c, c + n)), [samp] * n)
            e.decode(100)
            e.sync()
            engs.append(e)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            for rep in range(4):           # interleave the enqueues so every stream always has work queued
                for e in engs:
                    e.decode(8)
            for e in engs:
                e.sync()
            best = min(best, (time.time() - t0) / 32)
        solo = engs[0].last_timing()[1] / 8
        print(f"{n_eng} engine(s) x {B} slots: {best * 1e3:.3f} ms per step of all {total} sequences "
              f"(last 8-step burst of engine 0 alone on its stream: {solo:.3f} ms/step)", flush=True)
        for e in engs:
            e.close()


if __name__ == "__main__":
    main()
