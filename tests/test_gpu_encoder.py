"""Reference-encoding path (ref:neutts/neutts.py:266-271) on a real MI355X through the C-ABI: golden codes
from the live transformers models (Xcodec2Model.encode + SeamlessM4TFeatureExtractor, oracle/gen_golden_encoder.py) at the
emulator's tiny geometry and at NeuCodec geometry (w2v-BERT 16 x 1024, acoustic encoder 48 .. 1536, FSQ 4^8)."""
import numpy as np
import pytest
import torch

import synthetic as syn
from neutts import _hip
from common import check_encoder_codes, load_encoder_fixture, make_encoder_engine

pytestmark = pytest.mark.gpu

LAT_TOL = 2e-4      # fp32 MFMA pipeline vs the fp32 CPU reference (bound = ~10x the emulator's 1e-5 .. 3e-5)


@pytest.fixture(scope="module")
def lib(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _hip.load_library(hip_lib)
    return hip_lib


def test_encoder_tiny_vs_live_hf_golden(lib):
    z, cfg, w = load_encoder_fixture("encoder_tiny")
    eng = make_encoder_engine(cfg, w, lib)
    for i in range(int(z["n"])):
        wav = syn.synthetic_speech(int(z[f"n_samples_{i}"]), int(z[f"clip_seed_{i}"]))
        codes = eng.encode(wav)
        assert np.abs(eng.read_stage("features") - z[f"features_{i}"]).max() <= 2e-4
        check_encoder_codes(cfg, codes, eng.read_stage("latents"), z[f"codes_{i}"], z[f"latents_{i}"], LAT_TOL, f" tiny[{i}]")
        assert np.array_equal(eng.encode(wav), codes)


def test_encoder_neucodec_geometry_vs_live_hf_golden(lib):
    z, cfg, w = load_encoder_fixture("encoder_neucodec")
    assert (cfg.sem_hidden, cfg.sem_layers, cfg.ac_hidden, cfg.codec_hidden, cfg.hop) == (1024, 16, 48, 1024, 320)
    eng = make_encoder_engine(cfg, w, lib, max_samples=12 * 16000)
    del w
    wav = syn.synthetic_speech(int(z["n_samples_0"]), int(z["clip_seed_0"]))
    codes = eng.encode(wav)
    print(f"encoder neucodec geometry: {wav.size / 16000:.2f} s clip -> {codes.size} codes in {eng.last_timing():.1f} ms GPU time")
    assert np.abs(eng.read_stage("features") - z["features_0"]).max() <= 2e-4
    check_encoder_codes(cfg, codes, eng.read_stage("latents"), z["codes_0"], z["latents_0"], LAT_TOL, " neucodec")
    # a 10 s reference clip (the reference recommends 3-15 s): shape, determinism, timing
    long = syn.synthetic_speech(10 * 16000, 12)
    c1 = eng.encode(long)
    ms = eng.last_timing()
    assert c1.shape == (long.size // 320 + 1,) and c1.min() >= 0 and c1.max() < 65536
    assert np.array_equal(eng.encode(long), c1)
    assert len(np.unique(c1)) > 50
    print(f"encoder neucodec geometry: 10 s clip -> {c1.size} codes in {ms:.1f} ms GPU time ({10e3 / ms:.0f}x real time)")
