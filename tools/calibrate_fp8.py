#!/usr/bin/env python
"""fp8 activation-scale calibration (VERDICT r4 missing 4): from a BF16 backbone checkpoint to the static `input_scale` tensors the fp8
model needs (weight_dtype="fp8": e4m3 weights are quantised on upload, the activation scales must come from data).

    python tools/calibrate_fp8.py <checkpoint dir> [--prompts prompts.txt | --synthetic 64] [--margin 1.15] [--out input_scales.json]

<checkpoint dir> = a Hugging Face directory (config.json + *.safetensors [+ tokenizer]) of a Qwen2 / Llama-style decoder, loaded the way
NeuTTS(backbone_repo=dir) loads it.  Calibration data: one prompt per line of --prompts (token ids separated by blanks, or text when the
directory has a tokenizer), or --synthetic N seeded random prompts of 500 tokens (a smoke run: real scales need real prompts).  The
engine runs its prompt passes in calibration mode (ntts_backbone_calibrate: max |x| of every GEMM input), scale = amax / 448 x margin.
Output: {"<module>.input_scale": value} -- what NeuTTS(backbone_repo={... "input_scales": ...}) / load_state_dict(input_scales=...) take,
named like the tensors of a static-fp8 checkpoint (model.layers.N.{self_attn.q_proj, self_attn.o_proj, mlp.gate_proj, mlp.down_proj},
lm_head)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    sys.path.insert(0, p)


def calibrate(engine, prompts, max_new=1, chunk_tokens=None):
    """Run `prompts` (lists of token ids) through `engine` (a bf16 neutts._hip.BackboneEngine) in calibration mode; returns the scales."""
    from neutts import _hip
    budget = chunk_tokens or engine.cfg.get("max_prefill_tokens", 0) or 16384
    engine.calibrate(True)
    try:
        i = 0
        while i < len(prompts):
            j, used = i, 0
            while j < len(prompts) and j - i < engine.max_batch and (j == i or used + len(prompts[j]) <= budget):
                used += len(prompts[j])
                j += 1
            slots = [engine.acquire_slot() for _ in range(i, j)]
            samp = [_hip.Sampling(max_length=len(p) + max_new, min_new_tokens=0, eos_token_id=0, do_sample=False) for p in prompts[i:j]]
            engine.prefill(prompts[i:j], slots, samp)
            engine.sync()
            engine.release_many(slots)
            i = j
        return engine.fp8_input_scales()
    finally:
        engine.calibrate(False)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkpoint")
    ap.add_argument("--prompts")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--margin", type=float, default=1.15,
                    help="multiply every scale: head-room for activations larger than the calibration set's (values above amax x margin saturate in e4m3; "
                         "the record covers every position of the prompt passes, the decode steps' activations are not observed)")
    ap.add_argument("--out", default="input_scales.json")
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args()
    import numpy as np
    from neutts import NeuTTS
    t = NeuTTS.__new__(NeuTTS)          # the class's own backbone loader (HF config dispatch, safetensors streaming), without the codec half
    t.max_context, t._max_batch, t._lib_path = 2048, 16, None
    t._load_backbone(a.checkpoint, a.device)
    eng, tok = t.backbone, t.tokenizer
    V = eng.vocab_size
    if a.prompts:
        prompts = []
        for ln in open(a.prompts):
            ln = ln.strip()
            if not ln:
                continue
            parts = ln.split()
            prompts.append([int(x) for x in parts] if all(x.lstrip("-").isdigit() for x in parts) else tok.encode(ln))
    else:
        n = a.synthetic or 64
        prompts = [np.random.default_rng(1234 + i).integers(0, V, size=500).tolist() for i in range(n)]
        print(f"[calibrate_fp8] no --prompts: {n} synthetic prompts (a smoke run; real scales need real prompts)", file=sys.stderr)
    scales = calibrate(eng, prompts)
    scales = {k: v * a.margin for k, v in scales.items()}
    with open(a.out, "w") as fh:
        json.dump(scales, fh, indent=1)
    print(f"[calibrate_fp8] {len(prompts)} prompts, {sum(len(p) for p in prompts)} tokens -> {len(scales)} input scales in {a.out}")


if __name__ == "__main__":
    main()
