#!/usr/bin/env python
"""What do the decode step's GEMM / norm kernels cost when the launch carries MORE ROWS than the decode batch?  (Round 4: prompt
chunks riding in the decode step's launches.)  Engines of 256 / 512 / 768 / 1024 rows at NeuTTS-Air geometry, every kernel of the
step timed in isolation (ntts_backbone_time_kernel: HBM-cold replays over the layers), plus the hipGraph step itself.

    python tools/probe_rows.py [--rows 256,512,768,1024]        (through gpurun; one JSON line per engine)"""
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
    ap.add_argument("--rows", default="256,512,768,1024")
    ap.add_argument("--prompt", type=int, default=96)
    ap.add_argument("--running", type=int, default=0, help="slots that hold a running sequence (0 = all rows); the other rows ride along masked")
    ap.add_argument("--mid", type=int, default=8)
    ap.add_argument("--envs", default="[{}]", help="JSON list of env dicts: one measurement per entry and row count")
    a = ap.parse_args()
    cfg = br.BackboneConfig.neutts_air()
    w = br.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    inv = br.rope_inv_freq(cfg).numpy()
    for B, env in [(int(x), ev) for x in a.rows.split(",") for ev in json.loads(a.envs)]:
        for k in list(os.environ):
            if k.startswith("NTTS_X_"):
                del os.environ[k]
        for k, v in env.items():
            os.environ[k] = str(v)
        S = a.prompt
        R = a.running if a.running > 0 else B
        eng = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                       num_layers=cfg.num_layers, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                       max_context=((S + 160 + 31) // 32) * 32, max_batch=B, max_prefill_tokens=64 * S), 0)
        eng.load_state_dict(wd, inv_freq=inv)
        samp = _hip.Sampling(max_length=S + 100, min_new_tokens=100, eos_token_id=cfg.vocab_size - 1, do_sample=False)
        prompts = [br.synthetic_prompt(cfg, i, S) for i in range(64)]
        for c in range(0, R, 32):
            eng.prefill(prompts[:32], list(range(c, c + 32)), [samp] * 32)
        eng.decode(a.mid)
        eng.sync()
        best = 1e9
        for _ in range(3):
            eng.decode(16)
            eng.sync()
            best = min(best, eng.last_timing()[1] / 16)
        kern = {}
        for k, name in enumerate(_hip.BackboneEngine.KERNELS):
            ms, nb, nl = eng.time_kernel(k, 48)
            kern[name] = round(ms * 1e3, 2)
        eng.close()
        print(json.dumps({"rows": B, "running": R, "env": env, "context": S + a.mid + 24, "step_ms": round(best, 4), "isolated_us": kern}), flush=True)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"[probe_rows] {time.time() - t0:.0f}s", file=sys.stderr)
