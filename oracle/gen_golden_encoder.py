"""Encoder golden fixtures from the LIVE transformers models: Xcodec2Model.encode (w2v-BERT conformer + adapter + acoustic
encoder + fc + FSQ) fed with the live SeamlessM4TFeatureExtractor's features -- what neucodec's `encode_code` runs.
    python -m oracle.gen_golden_encoder
Fixtures: clip length + seed in (the clip is synthetic.synthetic_speech, the weights synthetic.make_encoder_weights, both
regenerated from their seeds), codes + twice-bounded FSQ latents + fbank features out."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synthetic as syn  # noqa: E402
from oracle import encoder_ref as er  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def hf_encode(cfg, w, wav):
    from transformers import SeamlessM4TFeatureExtractor
    m = er.hf_model(cfg, w)
    wp = er.pad_audio(wav, cfg.hop)
    feats = SeamlessM4TFeatureExtractor()(np.pad(wp, (160, 160)), sampling_rate=cfg.sample_rate, return_tensors="np").input_features[0]
    with torch.no_grad():
        out = m.encode(torch.from_numpy(wp).view(1, 1, -1), torch.from_numpy(feats).unsqueeze(0))
        # the twice-bounded latents are not an output of the live model: take them from its own sub-modules
        sem = m.semantic_adapter(m.semantic_encoder(torch.from_numpy(feats).unsqueeze(0)).last_hidden_state.transpose(1, 2))
        ac = m.acoustic_encoder(torch.from_numpy(wp).view(1, 1, -1))
        h = m.fc_encoder(torch.cat([sem, ac], dim=1).transpose(1, 2))
        z = m.quantizer.quantizer.bound(m.quantizer.quantizer.bound(m.quantizer.project_in(h).float()))
    return out.audio_codes[0, 0].numpy().astype(np.int32), z[0].numpy(), feats


def fixture(name, cfg, wseed, clips):
    w = syn.make_encoder_weights(cfg, wseed)
    rec = dict(cfg=np.array(list(cfg.to_dict().items()), dtype=object), seed=wseed, n=len(clips))
    for i, (n_samples, cseed) in enumerate(clips):
        wav = syn.synthetic_speech(n_samples, cseed)
        codes, z, feats = hf_encode(cfg, w, wav)
        assert codes.shape[0] == n_samples // cfg.hop + 1
        rec[f"n_samples_{i}"], rec[f"clip_seed_{i}"] = n_samples, cseed
        rec[f"codes_{i}"], rec[f"latents_{i}"], rec[f"features_{i}"] = codes, z.astype(np.float32), feats.astype(np.float32)
        print(name, i, n_samples, "->", codes.shape, codes[:6])
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **rec)


if __name__ == "__main__":
    # tiny geometry (the CPU SIMT emulator's size): a ragged clip, a clip that is an exact hop multiple, a sub-hop clip
    fixture("encoder_tiny", syn.EncoderConfig.tiny(), 3, [(8037, 1), (6400, 2), (200, 5)])
    # NeuCodec geometry (w2v-BERT 16 x 1024, acoustic 48 .. 1536): two seconds
    if "--full" in sys.argv:
        fixture("encoder_neucodec", syn.EncoderConfig.neucodec(), 7, [(32000 + 123, 11)])
