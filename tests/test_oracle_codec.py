"""Codec oracle (oracle/codec_ref.py) vs the committed golden vectors produced from transformers' xcodec2 modules at
NeuCodec geometry (oracle/gen_golden_codec.py), vs the live modules where importable, and the reference's own
_linear_overlap_add (restated) against a direct evaluation."""
import numpy as np
import pytest
import torch

from oracle import codec_ref as cr
from common import load_codec_fixture


@pytest.mark.parametrize("name", ["codec_tiny", "codec_neucodec"])
def test_codec_oracle_matches_golden(name):
    z, cfg, w = load_codec_fixture(name)
    for i in range(int(z["n"])):
        wav = cr.decode_code(cfg, w, torch.from_numpy(z[f"codes_{i}"]).long()).numpy()
        assert wav.shape == z[f"wav_{i}"].shape and wav.shape[-1] == cfg.hop_length * z[f"codes_{i}"].shape[-1]
        assert np.array_equal(wav, z[f"wav_{i}"])      # same torch ops, same order: bit-identical


def test_codec_oracle_vs_live_xcodec2():
    pytest.importorskip("transformers")
    from oracle.gen_golden_codec import hf_decoder, hf_decode
    cfg = cr.CodecConfig(hidden_size=192, intermediate_size=320, num_layers=1, num_heads=3, quantization_dim=128,
                         levels=(4, 4, 4), hop_length=40)
    w = cr.make_weights(cfg, 9)
    q, d = hf_decoder(cfg, w)
    codes = torch.randint(0, 64, (2, 1, 19), generator=torch.Generator().manual_seed(1))
    assert torch.equal(hf_decode(q, d, codes), cr.decode_code(cfg, w, codes))


def test_fsq_codebook_digits():
    """ref:examples/finetune_config.yaml:7 -- 65 536 codes = 8 base-4 digits; value (digit - 2) / 2."""
    cfg = cr.CodecConfig.neucodec()
    cb = cr.fsq_codebook(cfg)
    assert cb.shape == (65536, 8)
    assert sorted(set(cb.flatten().tolist())) == [-1.0, -0.5, 0.0, 0.5]
    i = 0b11_10_01_00_11_10_01_00            # digits, least significant first: 0,1,2,3,0,1,2,3
    assert cb[i].tolist() == [-1.0, -0.5, 0.0, 0.5, -1.0, -0.5, 0.0, 0.5]


def test_linear_overlap_add_properties():
    """ref:neutts/neutts.py:46-70: a constant signal cut into overlapping chunks is reconstructed exactly."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal(12000 * 3 + 960).astype(np.float32)
    frames = [x[i * 12000: i * 12000 + 12960] for i in range(3)] + [x[36000:]]
    y = cr.linear_overlap_add(frames, 12000)
    assert y.shape == x.shape and np.allclose(y, x, atol=1e-5)
