#!/usr/bin/env python
"""Probe (one MI355X, through gpurun): which kernels of the decode step scale when four chains run side by side?
rocprofv3's kernel trace serialises the chains, so this replays ONE kernel of the step (ntts_backbone_time_kernel: the step's own launch,
back to back over the layers, HIP events on the engine's stream) on 1 / 2 / 4 engines of a gang AT THE SAME TIME (one host thread per
engine; ctypes releases the GIL) and prints its time per launch alone and under self-contention.  A kernel whose time does not move
gains 4x in throughput from four chains; one that takes 4x as long gains nothing.

    python tools/probe_kernel_contention.py      (writes gpurun_out/kernel_contention.jsonl)"""
import argparse
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as br  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--prefill", type=int, default=500)
    ap.add_argument("--mid", type=int, default=125)
    ap.add_argument("--iters", type=int, default=480)
    ap.add_argument("--vocab", type=int, default=217488)
    a = ap.parse_args()
    cfg = br.BackboneConfig.neutts_air(a.vocab)
    w = br.make_weights(cfg, 0)
    wd = {k: v.to(torch.bfloat16).cuda() for k, v in w.items()}
    del w
    B, S = a.batch, a.prefill
    prompts = [br.synthetic_prompt(cfg, i, S) for i in range(B)]
    e0 = _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                  num_layers=cfg.num_layers, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                  max_context=768, max_batch=B, max_prefill_tokens=64 * S), 0)
    e0.load_state_dict(wd, inv_freq=br.rope_inv_freq(cfg).numpy())
    gang = _hip.EngineGang(e0, 4)
    samp = _hip.Sampling(max_length=S + 250, min_new_tokens=250, eos_token_id=cfg.vocab_size - 1, do_sample=False)
    for e in gang.engines:
        for c in range(0, B, 64):
            n = min(64, B - c)
            e.prefill(prompts[c:c + n], list(range(c, c + n)), [samp] * n)
        e.decode(a.mid)
        e.sync()
    out = open(os.path.join(ROOT, "gpurun_out", "kernel_contention.jsonl"), "a")
    for k, name in enumerate(_hip.BackboneEngine.KERNELS):
        iters = a.iters if name != "gemm_lm_head_argmax" else max(20, a.iters // 12)
        rec = {"kernel": name}
        for n in (1, 2, 4):
            res = [None] * n
            bar = threading.Barrier(n)

            def run(j):
                bar.wait()
                res[j] = gang.engines[j].time_kernel(k, iters)
            th = [threading.Thread(target=run, args=(j,)) for j in range(n)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            us = [r[0] * 1e3 for r in res]
            rec[f"us_per_launch_x{n}"] = [round(u, 2) for u in us]
            rec[f"GBps_total_x{n}"] = round(sum(res[j][1] / (us[j] * 1e-6) / 1e9 for j in range(n)), 0)
            rec["launches_per_step"] = res[0][2]
        rec["slowdown_x4"] = round(float(np.mean(rec["us_per_launch_x4"])) / rec["us_per_launch_x1"][0], 2)
        print(json.dumps(rec), flush=True)
        out.write(json.dumps(rec) + "\n")
    # Mixed: ONE engine replays the attention launch while the other three replay a GEMM of the step -- what a chain's attention costs
    # when the other chains are elsewhere in the layer (and what their GEMMs cost beside it).  The kernel given FEWER launches runs
    # entirely under the other's, so its figure is the one to read; each pair is run both ways.
    K = {n: i for i, n in enumerate(_hip.BackboneEngine.KERNELS)}
    for other in ("gemm_gate_up_silu", "gemm_down_splitk", "gemm_o_proj_splitk", "gemm_qkv", "add_rmsnorm_kernel"):
        rec = {"mixed": f"attn_decode_kernel on engine 0 | {other} on engines 1-3"}
        for short in ("attn", "other"):
            it_a, it_o = (300, 6000) if short == "attn" else (3000, 400)
            res = [None] * 4
            bar = threading.Barrier(4)

            def run(j):
                bar.wait()
                res[j] = gang.engines[j].time_kernel(K["attn_decode_kernel"] if j == 0 else K[other], it_a if j == 0 else it_o)
            th = [threading.Thread(target=run, args=(j,)) for j in range(4)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if short == "attn":
                rec["attn_us_under_3_others"] = round(res[0][0] * 1e3, 2)
            else:
                rec["other_us_x3_beside_1_attn"] = [round(r[0] * 1e3, 2) for r in res[1:]]
        print(json.dumps(rec), flush=True)
        out.write(json.dumps(rec) + "\n")
    gang.close()


if __name__ == "__main__":
    main()
