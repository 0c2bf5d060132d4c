#!/usr/bin/env python
"""Where does the decode attention kernel spend its time?  One launch at the BASELINE batch (256 x ~625 tokens) with the
kernel's phase timestamps (ntts_backbone_attn_timeline); prints, per phase, the time since the earliest kernel entry
(min / median / max over the 2048 waves).      python tools/attn_timeline.py      (through gpurun)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as br  # noqa: E402  (model geometry, seeded random weights / prompts: plain data)

PHASES = ["entry", "state known", "prologue done", "first K page done", "scores done", "softmax merged", "PV done", "exit"]


def main():
    cfg = br.BackboneConfig.neutts_air()
    w = br.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    B, S = int(os.environ.get("TL_BATCH", "256")), 500
    eng = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                                   intermediate_size=cfg.intermediate_size, num_layers=cfg.num_layers,
                                   num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                   max_context=768, max_batch=B, max_prefill_tokens=64 * S), 0)
    eng.load_state_dict(wd, inv_freq=br.rope_inv_freq(cfg).numpy())
    samp = _hip.Sampling(max_length=S + 250, min_new_tokens=250, eos_token_id=cfg.vocab_size - 1, do_sample=False)
    prompts = [br.synthetic_prompt(cfg, i, S) for i in range(B)]
    for c in range(0, B, 64):
        n = min(64, B - c)
        eng.prefill(prompts[c:c + n], list(range(c, c + n)), [samp] * n)
    eng.decode(125)
    eng.sync()
    for rep in range(3):
        t = eng.attn_timeline(3 + rep).astype(np.float64)          # [B, kvh, wave, phase]
        t0 = t[..., 0].min()
        rel = (t - t0) * 0.01                                       # us (100 MHz ticks)
        print(f"-- launch {rep}: us since the earliest wave entered the kernel (min / median / max over {t[..., 0].size} waves)")
        for ph, name in enumerate(PHASES):
            x = rel[..., ph].ravel()
            print(f"   {ph} {name:20s} {x.min():7.2f} {np.median(x):7.2f} {x.max():7.2f}")
        d = np.diff(rel, axis=-1)
        print("   phase durations (median):", " ".join(f"{np.median(d[..., k]):.2f}" for k in range(7)))
    ms, nb, nl = eng.time_kernel(0, 48)
    print(f"time_kernel: {ms * 1e3:.2f} us per launch")


if __name__ == "__main__":
    main()
