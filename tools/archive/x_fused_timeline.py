import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")): sys.path.insert(0, p)
import numpy as np, torch
from neutts import _hip
import synthetic as br
cfg = br.BackboneConfig.neutts_air()
w = br.make_weights(cfg, 0)
wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
inv = br.rope_inv_freq(cfg).numpy()
B, S = 256, 500
eng = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_layers=cfg.num_layers,
                               num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps, max_context=768, max_batch=B, max_prefill_tokens=64 * S), 0)
eng.load_state_dict(wd, inv_freq=inv)
samp = _hip.Sampling(max_length=S + 250, min_new_tokens=250, eos_token_id=cfg.vocab_size - 1, do_sample=False)
prompts = [br.synthetic_prompt(cfg, i, S) for i in range(B)]
for c in range(0, B, 64): eng.prefill(prompts[c:c + 64], list(range(c, c + 64)), [samp] * 64)
eng.decode(125); eng.sync()
for rep in range(2):
    tl = eng.gemv_timeline(5, 7).astype(np.int64)
    t0 = tl[:, 0][tl[:, 0] > 0].min()
    us = lambda a: (a - t0) / 100.0
    prod, cons = tl[:144], tl[144:656]
    print(f"rep {rep}: producers entry {us(prod[:,0]).min():.1f}..{us(prod[:,0]).max():.1f}  k-loop done {np.median(us(prod[:,1])):.1f}  exit {np.median(us(prod[:,2])):.1f} (max {us(prod[:,2]).max():.1f})")
    print("   deferred consumers:", int(cons[:, 7].sum()), "of", len(cons))
    for name, i in (("entry", 0), ("requests out", 1), ("hand-over done", 2), ("pass1 done", 3), ("merge done", 4), ("pass2 done", 5), ("exit", 6)):
        v = us(cons[:, i]); print(f"   consumers {name:15s} min {v.min():6.1f} p25 {np.percentile(v,25):6.1f} median {np.median(v):6.1f} p75 {np.percentile(v,75):6.1f} max {v.max():6.1f}")
