#!/usr/bin/env python
"""How far is the fp8 engine from the fp8 oracle, compared with how far the fp8 model is from the bf16 model?
    python tools/fp8_probe.py [lib]      (through gpurun; with tests/simt_emu/libneutts_emu.so: the emulator)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import backbone_ref as br  # noqa: E402  (a diagnostic tool: the oracle is the yardstick here)
from neutts import _hip  # noqa: E402
from common import engine_cfg  # noqa: E402


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else None
    for name, cfg, S in (("tiny-2L", br.BackboneConfig(vocab_size=512, hidden_size=384, intermediate_size=1024, num_layers=2, num_heads=6, num_kv_heads=2), 40),
                         ("nano-width-4L", br.BackboneConfig(vocab_size=8192, hidden_size=768, intermediate_size=2048, num_layers=4, num_heads=12, num_kv_heads=4), 70)):
        w = br.make_weights(cfg, 23, peak_sigma=0.5)
        scales = br.default_fp8_input_scales(cfg)
        wb = br.cast_weights(w, torch.bfloat16)
        wq = br.fp8_quantize_weights(wb, scales)
        p = br.synthetic_prompt(cfg, 0, S)
        eos = cfg.vocab_size - 1
        o8 = br.generate(cfg, wq, p, S + 1, eos, min_new_tokens=1, keep_logits=True).logits[0].numpy()[:-1]
        o16 = br.generate(cfg, wb, p, S + 1, eos, min_new_tokens=1, keep_logits=True).logits[0].numpy()[:-1]
        eng = _hip.BackboneEngine(engine_cfg(cfg, max_batch=2, max_context=128, max_prefill_tokens=256, weight_dtype="fp8"), 0, lib)
        eng.load_state_dict({k: v.numpy() for k, v in w.items()}, inv_freq=br.rope_inv_freq(cfg).numpy(), input_scales=scales)
        eng.set_debug(True)
        eng.prefill([p], [0], [_hip.Sampling(max_length=S + 1, min_new_tokens=1, eos_token_id=eos, do_sample=False)])
        g8 = eng.read_logits(0)[:-1]
        rms = lambda x: float(np.sqrt(np.mean(np.square(x))))
        print(f"{name}: logits rms {rms(o8):.3f} | engine-fp8 vs oracle-fp8: rel rms {rms(g8 - o8) / rms(o8):.4f} corr {np.corrcoef(g8, o8)[0, 1]:.5f} "
              f"argmax {int(g8.argmax())}/{int(o8.argmax())} | oracle-fp8 vs oracle-bf16: rel rms {rms(o8 - o16) / rms(o16):.4f} corr {np.corrcoef(o8, o16)[0, 1]:.5f}", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
