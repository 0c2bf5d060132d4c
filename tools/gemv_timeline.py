#!/usr/bin/env python
"""Where does a small-batch GEMV kernel spend its time?  One launch of the o_proj / gate-up / down_proj GEMV of a mid-generation batch-1
engine with the kernel's phase timestamps (ntts_backbone_gemv_timeline); prints, per phase, the time since the earliest wave entered the
kernel (min / median / max over the workgroups).      python tools/gemv_timeline.py      (through gpurun)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as br  # noqa: E402  (model geometry, seeded random weights / prompts: plain data)

F = ["entry", "weights requested", "weights landed", "X panel complete", "MFMA chain done", "stores done / K slices met", "(qkv) stores done"]
H = ["helper entry", "panel written", "past the barrier"]
T = ["entry", "first ring slots requested", "first tile landed", "k-loop done", "epilogue issued / K slices met", "stores drained"]   # tile kernels (batch > 8)


def main():
    cfg = br.BackboneConfig.neutts_air()
    w = br.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    B, S = int(os.environ.get("TL_BATCH", "1")), 500
    eng = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                   num_layers=cfg.num_layers, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                   max_context=768, max_batch=B, max_prefill_tokens=64 * S), 0)
    eng.load_state_dict(wd, inv_freq=br.rope_inv_freq(cfg).numpy())
    samp = _hip.Sampling(max_length=S + 250, min_new_tokens=250, eos_token_id=cfg.vocab_size - 1, do_sample=False)
    prompts = [br.synthetic_prompt(cfg, i, S) for i in range(B)]
    for c in range(0, B, 64):
        n = min(64, B - c)
        eng.prefill(prompts[c:c + n], list(range(c, c + n)), [samp] * n)
    eng.decode(125)
    eng.sync()
    for which, name in [(3, "gate/up")] if os.environ.get("TL_ONLY_GU") else ((1, "qkv + rope"), (2, "o_proj"), (3, "gate/up"), (4, "down_proj")):
        for rep in range(2):
            t = eng.gemv_timeline(which, 3 + rep).astype(np.float64)
            t[t == 0] = np.nan
            t0 = np.nanmin(t)
            rel = (t - t0) * 0.01                                  # us
            print(f"-- {name} launch {rep}: {len(t)} workgroups; us since the earliest wave entered the kernel (min / median / max)")
            names = list(enumerate(F)) + [(8 + k, nm) for k, nm in enumerate(H)] if B <= 8 else list(enumerate(T))
            for k, nm in names:
                x = rel[:, k]
                print(f"   {nm:20s} {np.nanmin(x):7.2f} {np.nanmedian(x):7.2f} {np.nanmax(x):7.2f}")
        ms, nb, nl = eng.time_kernel(which, 48)
        print(f"   time_kernel: {ms * 1e3:.2f} us per launch (back to back, includes the launch gap)")


if __name__ == "__main__":
    main()
