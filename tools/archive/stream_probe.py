#!/usr/bin/env python
"""Streaming leg (SURVEY 8f-1 / BASELINE config 5's "streaming NeuCodec decode overlapped on a side stream"):
time-to-first-audio and chunk cadence of NeuTTS.infer_stream at NeuTTS-Air geometry, synthetic weights, one
utterance, 500-token prompt, EOS masked for 250 new tokens; with and without backbone/codec overlap.
    python tools/stream_probe.py [--tokens 250]      (through gpurun; prints one JSON line per variant)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neutts import NeuTTS  # noqa: E402
import synthetic as br  # noqa: E402  (model geometry, seeded random weights / prompts: plain data)
import synthetic as cr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=250)
    ap.add_argument("--prompt", type=int, default=500)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="utterances of the infer_stream_batch leg (0 = skip)")
    a = ap.parse_args()
    cfg = br.BackboneConfig.neutts_air()
    ccfg = cr.CodecConfig.neucodec()
    w = br.make_weights(cfg, 0)
    cw = cr.make_codec_weights(ccfg, 0)
    eos = cfg.vocab_size - 1
    tts = NeuTTS(
        backbone_repo={"config": dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                                      intermediate_size=cfg.intermediate_size, num_layers=cfg.num_layers,
                                      num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                      max_context=1024, max_prefill_tokens=8192),
                       "state_dict": {k: v.numpy() for k, v in w.items()}, "inv_freq": br.rope_inv_freq(cfg).numpy(),
                       "tokenizer": None, "speech_base": 0, "eos_token_id": eos},
        codec_repo={"config": dict(hidden_size=ccfg.hidden_size, intermediate_size=ccfg.intermediate_size,
                                   num_layers=ccfg.num_layers, num_heads=ccfg.num_heads,
                                   quantization_dim=ccfg.quantization_dim, levels=list(ccfg.levels),
                                   hop_length=ccfg.hop_length, rms_eps=ccfg.rms_eps, max_frames=1024,
                                   max_rows=max(2048, a.batch * 96)),
                    "state_dict": {k: v.numpy() for k, v in cw.items()}},
        do_sample=False, max_batch=max(1, a.batch))
    tts.watermarker = None
    tts._ids_to_codes = lambda ids: [i % 65536 for i in ids]     # random weights do not stay in the speech range
    tts.min_new_tokens = a.tokens
    tts.max_context = a.prompt + a.tokens
    prompt = br.synthetic_prompt(cfg, 0, a.prompt)
    ref_codes = [int(t) % 65536 for t in prompt[-372:]]           # as long as ref:samples/dave.pt
    for overlap in (True, False):
        tts.streaming_overlap_compute = overlap
        best = None
        for _ in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.time()
            stamps, n = [], 0
            for chunk in tts._infer_stream_hip(list(prompt), ref_codes):
                stamps.append(time.time() - t0)
                n += len(chunk)
            rec = {"overlap": overlap, "ttfa_ms": stamps[0] * 1e3, "total_ms": stamps[-1] * 1e3, "chunks": len(stamps),
                   "mean_chunk_period_ms": (stamps[-2] - stamps[0]) / max(len(stamps) - 2, 1) * 1e3,
                   "audio_s": n / 24000.0, "rtf": stamps[-1] / (n / 24000.0)}
            if best is None or rec["total_ms"] < best["total_ms"]:
                best = rec
        print(json.dumps(best), flush=True)

    if a.batch > 1:    # many utterances streamed from one decode batch (BASELINE config 5's shape)
        tts.streaming_overlap_compute = True
        prompts = [list(br.synthetic_prompt(cfg, i, a.prompt)) for i in range(a.batch)]
        refs = [[int(t) % 65536 for t in p[-372:]] for p in prompts]
        best = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            first, last, n = {}, 0.0, 0
            for i, chunk in tts._infer_stream_batch_hip(prompts, refs):
                now = time.time() - t0
                first.setdefault(i, now)
                last = now
                n += len(chunk)
            rec = {"stream_batch": a.batch, "ttfa_ms_first": min(first.values()) * 1e3, "ttfa_ms_last": max(first.values()) * 1e3,
                   "total_ms": last * 1e3, "audio_s": n / 24000.0, "rtf": last / (n / 24000.0),
                   "codec_tokens_per_s": a.batch * a.tokens / last}
            if best is None or rec["total_ms"] < best["total_ms"]:
                best = rec
        print(json.dumps(best), flush=True)


if __name__ == "__main__":
    main()
