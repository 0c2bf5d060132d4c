#!/usr/bin/env python
"""Decode-step tuning sweep UNDER AN ENGINE GANG on one MI355X: for each setting of the engine's env knobs build a gang of `--engines`
256-row engines on one arena (lane stream per engine), bring every engine to mid-generation (context ~ prefill + mid) and time
`--steps` decode steps per engine, the step graphs replayed alternately from one thread (the schedule of bench.py's static mode).
Also times one engine of the gang alone, and prints a checksum of engine 0's ids so that bit-neutral settings can be told from
settings that move the arithmetic.

    python tools/sweep_gang.py --settings '[{}, {"NTTS_TALL": 3}, {"NTTS_TALL": 3, "NTTS_GU_TILE": 1}]'
(run through gpurun; appends to gpurun_out/sweep_gang.jsonl)"""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as br  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--prefill", type=int, default=500)
    ap.add_argument("--mid", type=int, default=100)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--vocab", type=int, default=217488)
    ap.add_argument("--engines", type=int, default=4)
    ap.add_argument("--settings", type=str, default="[{}]", help="JSON list of {env knob: value} dicts, measured in order")
    ap.add_argument("--config", choices=["air", "nano-fp8"], default="air")
    ap.add_argument("--stagger", type=float, nargs="*", default=[], help="extra legs: engine k's chain starts k x this many microseconds late (a spin kernel on its lane)")
    ap.add_argument("--kernels", action="store_true", help="also time every decode kernel on engine 0 alone (ntts_backbone_time_kernel)")
    a = ap.parse_args()
    fp8 = a.config == "nano-fp8"
    cfg = br.BackboneConfig.neutts_nano_like(142080) if fp8 else br.BackboneConfig.neutts_air(a.vocab)
    scales = br.default_fp8_input_scales(cfg) if fp8 else None
    w = br.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    inv = br.rope_inv_freq(cfg).numpy()
    B, S = a.batch, a.prefill
    prompts = [br.synthetic_prompt(cfg, i, S) for i in range(B)]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    out = open(os.path.join(ROOT, "gpurun_out", "sweep_gang.jsonl"), "a")
    samp = _hip.Sampling(max_length=S + 250, min_new_tokens=250, eos_token_id=cfg.vocab_size - 1, do_sample=False)
    names = {0: "attn", 1: "qkv", 2: "o_proj", 3: "gate_up", 4: "down", 5: "lm_head", 6: "add_norm"}

    def measure(env):
        for k in list(os.environ):
            if k.startswith("NTTS_") and k not in ("NTTS_FORCE_BUILD",):
                del os.environ[k]
        for k, v in env.items():
            os.environ[k] = str(v)
        e0 = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                      num_layers=cfg.num_layers, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                      max_context=768, max_batch=B, max_prefill_tokens=64 * S, weight_dtype="fp8" if fp8 else "bf16"), 0)
        e0.load_state_dict(wd, inv_freq=inv, input_scales=scales)
        gang = _hip.EngineGang(e0, a.engines)
        engs = gang.engines
        for e in engs:
            for c in range(0, B, 64):
                n = min(64, B - c)
                e.prefill(prompts[c:c + n], list(range(c, c + n)), [samp] * n)
            e.decode(a.mid)
        gang.sync()

        def wall(fn):
            gang.sync()
            t0 = time.perf_counter()
            fn()
            gang.sync()
            return (time.perf_counter() - t0) * 1e3

        K = a.steps

        def alternately():
            for _ in range(K):
                for e in engs:
                    e.decode(1)
        gang_ms = [wall(alternately) / K / len(engs) for _ in range(a.reps)]
        stag = {}
        if a.stagger and len(engs) > 1 and gang.lane(0):
            lanes = [torch.cuda.ExternalStream(gang.lane(k)) for k in range(len(engs))]
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(lanes[0]):                     # spin-kernel calibration: cycles per microsecond
                ev0.record(); torch.cuda._sleep(2_000_000); ev1.record()
            gang.sync()
            cyc_per_us = 2_000_000 / (ev0.elapsed_time(ev1) * 1e3)
            for d in a.stagger:
                def staggered():
                    for k in range(1, len(engs)):
                        with torch.cuda.stream(lanes[k]):
                            torch.cuda._sleep(int(k * d * cyc_per_us))
                    alternately()
                ms = [wall(staggered) for _ in range(2)]
                stag[str(d)] = [round((x - (len(engs) - 1) * d * 1e-3) / K / len(engs), 4) for x in ms]    # the last chain's head start taken off
        one_ms = [wall(lambda: e0.decode(K)) / K for _ in range(2)]
        rec = {"env": env, "engines": len(engs), "ms_per_256_row_step_gang": [round(x, 4) for x in gang_ms], "gang_min": round(min(gang_ms), 4),
               "one_engine_ms_per_step": [round(x, 4) for x in one_ms]}
        if stag:
            rec["ms_per_step_gang_staggered_us"] = stag
        if a.kernels:
            ker = {}
            for which, nm in names.items():
                try:
                    ms, by, per = e0.time_kernel(which, 48)
                    ker[nm] = round(ms * 1e3, 2)
                except Exception as ex:  # noqa: BLE001
                    ker[nm] = str(ex)[:60]
            rec["kernel_us_alone"] = ker
        toks = [e.read(0)[0] for e in engs]
        rec["ids_crc_slot0"] = [zlib.crc32(np.asarray(t, dtype=np.int32).tobytes()) & 0xffffffff for t in toks]
        rec["n_ids_slot0"] = [len(t) for t in toks]
        print(json.dumps(rec), flush=True)
        out.write(json.dumps(rec) + "\n")
        out.flush()
        gang.close()
        e0.close()

    settings = json.load(open(a.settings[1:])) if a.settings.startswith("@") else json.loads(a.settings)
    for env in settings:
        try:
            measure(env)
        except Exception as ex:  # noqa: BLE001
            print(json.dumps({"env": env, "error": repr(ex)[:300]}), flush=True)


if __name__ == "__main__":
    main()
