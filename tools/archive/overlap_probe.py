#!/usr/bin/env python
"""Focused overlap test: engine A decodes 249 steps while engine B runs a CU-masked prompt pass.  Alone vs together."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as syn  # noqa: E402
from pipeline_probe import cu_mask  # noqa: E402


def main():
    B, S, N = 256, 500, 250
    ncu = int(os.environ.get("PP_CUS", "96"))
    layout = int(os.environ.get("PP_LAYOUT", "0"))
    cfg = syn.BackboneConfig.neutts_air()
    w = syn.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    inv = syn.rope_inv_freq(cfg).numpy()
    engs = []
    for _ in range(2):
        e = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                     num_layers=cfg.num_layers, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                     max_context=768, max_batch=B, max_prefill_tokens=64 * S), 0)
        e.load_state_dict(wd, inv_freq=inv)
        engs.append(e)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=cfg.vocab_size - 1, do_sample=False)
    prompts = [syn.synthetic_prompt(cfg, i, S) for i in range(B)]
    A, Bn = engs

    def prefill(e):
        for c in range(0, B, 64):
            e.prefill(prompts[c:c + 64], list(range(c, c + 64)), [samp] * 64)
        e.sync()

    def release(e):
        for s in range(B):
            e.release(s)
        e.sync()

    def timed(fn):
        t0 = time.time(); fn(); return (time.time() - t0) * 1e3

    for mask_on in (False, True):
        Bn.set_prefill_cu_mask(cu_mask(ncu, layout) if mask_on else None)
        # alone
        prefill(A)
        t_dec = timed(lambda: (A.decode(N - 1), A.sync()))
        release(A)
        t_pf = timed(lambda: prefill(Bn))
        release(Bn)
        # together
        prefill(A)
        marks = {}

        def side():
            t0 = time.time()
            prefill(Bn)
            marks["pf"] = (time.time() - t0) * 1e3

        th = threading.Thread(target=side)
        t0 = time.time()
        th.start()
        A.decode(N - 1)
        marks["dec_enq"] = (time.time() - t0) * 1e3
        A.sync()
        marks["dec"] = (time.time() - t0) * 1e3
        th.join()
        marks["both"] = (time.time() - t0) * 1e3
        release(A); release(Bn)
        print(json.dumps({"side_cus": ncu if mask_on else 256, "layout": layout, "hwq": os.environ.get("GPU_MAX_HW_QUEUES", "default"),
                          "alone_decode_ms": round(t_dec, 1), "alone_prefill_ms": round(t_pf, 1),
                          "together": {k: round(v, 1) for k, v in marks.items()}}), flush=True)


if __name__ == "__main__":
    main()
