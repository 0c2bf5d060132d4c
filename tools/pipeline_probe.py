#!/usr/bin/env python
"""Does the GPU overlap batch k's decode with batch k+1's prefill and batch k-1's codec pass (three HIP streams, two backbone
engines + the codec engine)?  Times K batches serially and pipelined.   python tools/pipeline_probe.py   (through gpurun)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as syn  # noqa: E402


def main():
    B, S, N, K = 256, 500, 250, int(os.environ.get("PP_BATCHES", "4"))
    cfg, ccfg = syn.BackboneConfig.neutts_air(), syn.CodecConfig.neucodec()
    n_codes = int(np.prod(ccfg.levels))
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
    cod = _hip.CodecEngine(dict(hidden_size=ccfg.hidden_size, intermediate_size=ccfg.intermediate_size, num_layers=ccfg.num_layers,
                                num_heads=ccfg.num_heads, quantization_dim=ccfg.quantization_dim, levels=list(ccfg.levels),
                                hop_length=ccfg.hop_length, rms_eps=ccfg.rms_eps, max_frames=N, max_rows=B * (N + 6)), 0)
    cod.load_state_dict({k: v.numpy() for k, v in syn.make_codec_weights(ccfg, 0).items()})
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=cfg.vocab_size - 1, do_sample=False)
    prompts = [syn.synthetic_prompt(cfg, i, S) for i in range(B)]
    codes = [torch.zeros((B, N), dtype=torch.int32, device="cuda") for _ in range(2)]
    lens = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(2)]
    full = np.full(B, N, dtype=np.int32)

    def prefill(e):
        for c in range(0, B, 64):
            e.prefill(prompts[c:c + 64], list(range(c, c + 64)), [samp] * 64)

    def finish(i):      # decode done -> export codes, free the slots
        e = engs[i]
        e.export_codes(list(range(B)), 0, n_codes, codes[i].data_ptr(), N, lens[i].data_ptr(), modulo=True)
        for s in range(B):
            e.release(s)

    def serial(k):
        t0 = time.time()
        for b in range(k):
            e = engs[0]
            prefill(e)
            e.decode(N - 1)
            finish(0)
            cod.decode_device(codes[0].data_ptr(), N, full, producer_stream=e.stream())
            cod.sync()
        return time.time() - t0

    def pipelined(k):
        t0 = time.time()
        prefill(engs[0])                                   # fill
        for b in range(k):
            cur, nxt = engs[b % 2], engs[(b + 1) % 2]
            cur.decode(N - 1)                              # async: 249 graph replays on cur's stream
            if b + 1 < k:
                prefill(nxt)                               # next batch's prompt pass on the other engine's stream
            finish(b % 2)                                  # (blocks until cur's decode is done)
            cod.decode_device(codes[b % 2].data_ptr(), N, full, producer_stream=cur.stream())   # async on the codec stream
        cod.sync()
        return time.time() - t0

    for name, fn in (("serial", serial), ("pipelined", pipelined), ("serial", serial), ("pipelined", pipelined)):
        dt = fn(K)
        print(f"{name:10s}: {K} batches in {dt * 1e3:8.1f} ms = {dt / K * 1e3:7.1f} ms per batch = {K * B * N / dt:9.0f} codec-tokens/s", flush=True)


if __name__ == "__main__":
    main()
