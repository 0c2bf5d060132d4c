"""Codec golden fixtures from the live transformers xcodec2 modules at NeuCodec geometry (hop 480, n_fft 1920).
    python -m oracle.gen_golden_codec
Fixtures: codes in, waveform out (fp32) -- weights are regenerated from the seed (oracle.codec_ref.make_weights)."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import codec_ref as cr  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def hf_decoder(cfg: cr.CodecConfig, w):
    """Xcodec2Quantizer + Xcodec2Decoder with OUR weights loaded (strict)."""
    from transformers import Xcodec2Config
    from transformers.models.xcodec2.modeling_xcodec2 import Xcodec2Decoder, Xcodec2Quantizer
    # factor hop into downsampling ratios (only their product matters to the decoder: hop_length / n_fft)
    hop, ratios = cfg.hop_length, []
    for p in (2, 2, 2, 2, 2, 3, 3, 5, 5, 7):
        if hop % p == 0 and len(ratios) < 5:
            ratios.append(p); hop //= p
    ratios[-1] *= hop
    while len(ratios) < 5:
        ratios.append(1)
    hc = Xcodec2Config(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                       num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads,
                       num_key_value_heads=cfg.num_heads, head_dim=cfg.head_dim, quantization_dim=cfg.quantization_dim,
                       quantization_levels=tuple(cfg.levels), downsampling_ratios=tuple(ratios), rms_norm_eps=cfg.rms_eps,
                       rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta},
                       semantic_model_config={"hidden_size": cfg.quantization_dim - cfg.hidden_size, "num_hidden_layers": 1},
                       attn_implementation="eager")
    assert hc.hop_length == cfg.hop_length and hc.n_fft == cfg.n_fft
    q, d = Xcodec2Quantizer(hc).eval(), Xcodec2Decoder(hc).eval()
    d.config = hc
    for lyr in d.layers:
        lyr.self_attn.config._attn_implementation = "eager"
    qsd = {k[len("quantizer."):]: v for k, v in w.items() if k.startswith("quantizer.")}
    qsd.update({k: v for k, v in q.state_dict().items() if k not in qsd})   # project_in (encoder side) untouched
    q.load_state_dict(qsd, strict=True)
    d.load_state_dict({k[len("decoder."):]: v for k, v in w.items() if k.startswith("decoder.")}, strict=True)
    return q, d


def hf_decode(q, d, codes):
    with torch.no_grad():
        return d(q.from_codes(codes.transpose(1, 2)))


def fixture(name, cfg, seed, code_sets):
    w = cr.make_weights(cfg, seed)
    q, d = hf_decoder(cfg, w)
    rec = dict(cfg=np.array(list(cfg.to_dict().items()), dtype=object), seed=seed, n=len(code_sets))
    for i, codes in enumerate(code_sets):
        t = time.time()
        wav = hf_decode(q, d, codes)
        print(f"[{name}] set {i}: codes {tuple(codes.shape)} -> wav {tuple(wav.shape)} rms {float(wav.pow(2).mean().sqrt()):.4f} "
              f"in {time.time() - t:.2f}s")
        rec[f"codes_{i}"] = codes.numpy().astype(np.int32)
        rec[f"wav_{i}"] = wav.numpy().astype(np.float32)
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **rec)


def main():
    g = torch.Generator().manual_seed(0)
    tiny = cr.CodecConfig.tiny()
    ncode = int(np.prod(tiny.levels))
    fixture("codec_tiny", tiny, 0, [torch.randint(0, ncode, (2, 1, 37), generator=g), torch.randint(0, ncode, (1, 1, 5), generator=g)])
    full = cr.CodecConfig.neucodec()
    sets = [torch.randint(0, 65536, (1, 1, 50), generator=g)]
    ref = "/root/reference/samples/dave.pt"
    if os.path.exists(ref):   # realistic code sequence: the reference's own sample voice (ref:samples/dave.pt, int32[372])
        sets.append(torch.load(ref).long()[None, None, :100])
    fixture("codec_neucodec", full, 0, sets)


if __name__ == "__main__":
    main()
