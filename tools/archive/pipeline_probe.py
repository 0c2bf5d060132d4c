#!/usr/bin/env python
"""Does the GPU overlap batch k's decode with batch k+1's prefill and batch k-1's codec pass?  Two backbone engines + the codec
engine; the prompt pass and the codec pass run on CU-MASKED side streams (ntts_backbone_set_prefill_cu_mask /
ntts_codec_set_cu_mask) so that their 1024-thread workgroups leave CUs to the other engine's decode steps; two host threads
(the ctypes calls release the GIL) keep both queues fed.  Times K batches serially and pipelined.
    PP_BATCHES=6 PP_CUS=96 PP_LAYOUT=0 python tools/pipeline_probe.py   (through gpurun)"""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neutts import _hip  # noqa: E402
import synthetic as syn  # noqa: E402


def cu_mask(n_cus, layout, total=256):
    """layout 0: CUs 0 .. n-1; 1: the first n/8 CUs of every group of 32; 2: every (total/n)-th CU"""
    bits = [0] * total
    if layout == 0:
        for i in range(n_cus):
            bits[i] = 1
    elif layout == 1:
        per = n_cus // 8
        for g in range(8):
            for i in range(per):
                bits[g * 32 + i] = 1
    else:
        step = total / n_cus
        for i in range(n_cus):
            bits[int(i * step)] = 1
    return [sum(bits[w * 32 + b] << b for b in range(32)) for w in range(total // 32)]


def main():
    B, S, N, K = 256, 500, 250, int(os.environ.get("PP_BATCHES", "6"))
    ncu = int(os.environ.get("PP_CUS", "96"))
    layout = int(os.environ.get("PP_LAYOUT", "0"))
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

    def set_masks(on):
        m = cu_mask(ncu, layout) if on else None
        for e in engs:
            e.set_prefill_cu_mask(m)
        cod.set_cu_mask(m)

    def prefill(e):
        for c in range(0, B, 64):
            e.prefill(prompts[c:c + 64], list(range(c, c + 64)), [samp] * 64)

    def finish(i):      # decode done -> export codes, free the slots
        e = engs[i]
        e.export_codes(list(range(B)), 0, n_codes, codes[i].data_ptr(), N, lens[i].data_ptr(), modulo=True)
        for s in range(B):
            e.release(s)

    wav_keep = {}

    def serial(k):
        t0 = time.time()
        for b in range(k):
            e = engs[0]
            prefill(e)
            e.decode(N - 1)
            finish(0)
            wv = cod.decode_device(codes[0].data_ptr(), N, full, producer_stream=e.stream())
            cod.sync()
            wav_keep["serial"] = wv[:4, :2000].copy()
        return time.time() - t0

    def pipelined(k):
        pf_done = [threading.Event() for _ in range(k)]
        free = [threading.Event() for _ in range(k + 2)]      # free[b] = the engine of batch b is free (batch b-2 exported)
        free[0].set(); free[1].set()

        t0 = time.time()
        log = []
        stamp = lambda tag: log.append((tag, round((time.time() - t0) * 1e3)))

        def feeder():                                       # thread B: prompt passes, one batch ahead
            for b in range(k):
                free[b].wait()
                stamp(f"pf{b}+")
                prefill(engs[b % 2])
                stamp(f"pf{b}-")
                pf_done[b].set()

        th = threading.Thread(target=feeder)
        th.start()
        for b in range(k):                                  # thread A: decode, hand-off, codec
            cur = engs[b % 2]
            pf_done[b].wait()
            stamp(f"dec{b}+")
            cur.decode(N - 1)                               # 249 graph replays, behind the prompt pass (event-ordered)
            stamp(f"dec{b}enq")
            finish(b % 2)                                   # blocks until the decode is done
            stamp(f"dec{b}-")
            free[b + 2].set()
            cod.sync()                                      # the previous batch's waveforms have left the pinned buffer
            stamp(f"codsync{b}")
            wv = cod.decode_device(codes[b % 2].data_ptr(), N, full, producer_stream=cur.stream())
            stamp(f"codenq{b}")
        cod.sync()
        th.join()
        if os.environ.get("PP_LOG"):
            print(sorted(log, key=lambda x: x[1]), flush=True)
        wav_keep["pipelined"] = wv[:4, :2000].copy()
        return time.time() - t0

    def pipelined_1t(k):
        """ONE launching thread: the prompt pass of batch b + 1 is enqueued (asynchronous calls, the other engine's stream) BEFORE
        the 249 graph replays of batch b, so both queues hold work while the host sits in the decode enqueue's back-pressure."""
        t0 = time.time()
        prefill(engs[0])
        for b in range(k):
            cur = engs[b % 2]
            if b + 1 < k:
                prefill(engs[(b + 1) % 2])                  # engine (b + 1) % 2 was exported and released in iteration b - 1
            cur.decode(N - 1)
            cur.export_codes(list(range(B)), 0, n_codes, codes[b % 2].data_ptr(), N, lens[b % 2].data_ptr(), modulo=True)
            st, n_new = cur.poll()                          # blocking, as in bench.py: batch b's decode and export are done
            assert (n_new == N).all()
            for s_ in range(B):
                cur.release(s_)
            cod.sync()
            wv = cod.decode_device(codes[b % 2].data_ptr(), N, full, producer_stream=cur.stream())
        cod.sync()
        wav_keep["pipelined"] = wv[:4, :2000].copy()
        return time.time() - t0

    res = []
    modes = {"serial": (serial, False), "pipelined-mask": (pipelined, True), "pipelined-1t": (pipelined_1t, False),
             "pipelined-1t-mask": (pipelined_1t, True)}
    for name in os.environ.get("PP_MODES", "serial,pipelined-1t,pipelined-1t-mask").split(","):
        fn, masks = modes[name]
        set_masks(masks)
        fn(2)                                               # (untimed: first use of a mode)
        dt = fn(K)
        rec = {"mode": name, "batches": K, "side_cus": ncu if masks else 256, "layout": layout, "ms_per_batch": dt / K * 1e3,
               "codec_tokens_per_s": K * B * N / dt}
        res.append(rec)
        print(json.dumps(rec), flush=True)
    same = bool(np.array_equal(wav_keep["serial"], wav_keep["pipelined"]))
    print(json.dumps({"waveforms_identical_serial_vs_pipelined": same}), flush=True)


if __name__ == "__main__":
    main()
