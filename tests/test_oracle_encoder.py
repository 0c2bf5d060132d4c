"""Encoder oracle (oracle/encoder_ref.py): the restatement against the committed golden vectors produced by the LIVE
transformers models (oracle/gen_golden_encoder.py: Xcodec2Model.encode fed by SeamlessM4TFeatureExtractor), and -- where
transformers is importable -- against the live models on a fresh clip."""
import numpy as np
import pytest
import torch

import synthetic as syn
from oracle import encoder_ref as er
from common import load_encoder_fixture


@pytest.mark.parametrize("i", [0, 1, 2])
def test_restatement_vs_live_golden_tiny(i):
    z, cfg, w = load_encoder_fixture("encoder_tiny")
    wav = syn.synthetic_speech(int(z[f"n_samples_{i}"]), int(z[f"clip_seed_{i}"]))
    codes, parts = er.encode(cfg, w, wav, return_parts=True)
    assert np.array_equal(codes, z[f"codes_{i}"])                                  # integer output: identical
    assert np.abs(parts["latents"] - z[f"latents_{i}"]).max() <= 2e-5
    assert np.abs(parts["features"] - z[f"features_{i}"]).max() <= 1e-4            # float64 numpy fbank both sides


def test_pad_rule_and_front_end_shapes():
    """hf:models/xcodec2/feature_extraction_xcodec2.py:149-158: one zero, then up to the next hop multiple."""
    for L, T in ((1, 1), (319, 1), (320, 2), (321, 2), (6400, 21), (8037, 26)):
        assert er.pad_audio(np.zeros(L, np.float32)).size == T * 320
        assert er.fbank_features(er.pad_audio(np.ones(L, np.float32) * 0.01)).shape == (T, 160)
    f = er.kaldi_mel_filters()
    assert f.shape == (257, 80) and (f >= 0).all() and (f.max(0) > 0).all()
    k = er.kaiser_sinc_filter(0.25, 0.3, 12)
    assert abs(float(k.sum()) - 1.0) < 1e-6 and torch.allclose(k, k.flip(0), atol=1e-7)


def test_restatement_vs_live_hf_fresh_clip():
    pytest.importorskip("transformers")
    from oracle.gen_golden_encoder import hf_encode
    cfg = syn.EncoderConfig.tiny()
    w = syn.make_encoder_weights(cfg, 21)
    wav = syn.synthetic_speech(4321, 4)
    codes, z, feats = hf_encode(cfg, w, wav)
    got, parts = er.encode(cfg, w, wav, return_parts=True)
    assert np.array_equal(got, codes)
    assert np.abs(parts["latents"] - z).max() <= 2e-5 and np.abs(parts["features"] - feats).max() <= 1e-4
