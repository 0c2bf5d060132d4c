#!/usr/bin/env python
"""Engine clock and socket power during the phases of the benchmark step (prompt pass, decode loop): `rocm-smi` sampled from a
second thread while the engine runs each phase for a few seconds.  The decode step is a chain of 171 short latency-bound kernels:
if the power manager does not hold the boost clock under that load, every one of them is slower than it need be.
    python tools/clock_probe_decode.py        (through gpurun)"""
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as syn  # noqa: E402


def sampler(stop, out):
    while not stop.is_set():
        t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        sclk = re.search(r"sclk clock level: \S+: \((\d+)Mhz\)", t)
        mclk = re.search(r"mclk clock level: \S+: \((\d+)Mhz\)", t)
        fclk = re.search(r"fclk clock level: \S+: \((\d+)Mhz\)", t)
        pw = re.search(r"Power \(W\): ([\d.]+)", t)
        out.append((int(sclk.group(1)) if sclk else None, int(mclk.group(1)) if mclk else None, int(fclk.group(1)) if fclk else None,
                    float(pw.group(1)) if pw else None))


def sampled(fn):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sampler, args=(stop, out))
    th.start()
    t0 = time.time()
    r = fn()
    dt = time.time() - t0
    stop.set()
    th.join()
    return r, dt, out


def main():
    B, S, N = 256, 500, 250
    cfg = syn.BackboneConfig.neutts_air()
    w = syn.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    eng = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                   num_layers=cfg.num_layers, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                   max_context=2048, max_batch=B, max_prefill_tokens=64 * S), 0)
    eng.load_state_dict(wd, inv_freq=syn.rope_inv_freq(cfg).numpy())
    prompts = [syn.synthetic_prompt(cfg, i, S) for i in range(B)]
    samp = _hip.Sampling(max_length=2040, min_new_tokens=1500, eos_token_id=cfg.vocab_size - 1, do_sample=False)

    def prefill():
        for c in range(0, B, 64):
            eng.prefill(prompts[c:c + 64], list(range(c, c + 64)), [samp] * 64)
        eng.sync()
        return eng.last_timing()[0]

    def decode(n):
        eng.decode(n)
        eng.sync()
        return eng.last_timing()[1] / n

    pf, dt, s = sampled(prefill)
    print(f"prompt pass: last chunk {pf:.1f} ms, {dt:.2f} s wall; (sclk, mclk, fclk MHz, W): {s}", flush=True)
    decode(10)
    for k in range(3):
        ms, dt, s = sampled(lambda: decode(400))
        print(f"decode x400 (context {500 + 10 + 400 * k}..): {ms:.4f} ms per step, {dt:.2f} s wall; (sclk, mclk, fclk MHz, W): {s}", flush=True)
    st = os.environ.get("PROBE_SETPERF")
    if st:
        print(subprocess.run(["rocm-smi"] + st.split(), capture_output=True, text=True).stdout[-600:], flush=True)
        ms, dt, s = sampled(lambda: decode(200))
        print(f"after rocm-smi {st}: decode x200: {ms:.4f} ms per step; samples: {s}", flush=True)
        print(subprocess.run(["rocm-smi", "--setperflevel", "auto"], capture_output=True, text=True).stdout[-300:], flush=True)


if __name__ == "__main__":
    main()
