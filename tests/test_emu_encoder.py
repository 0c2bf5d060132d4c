"""Reference-encoding path (ref:neutts/neutts.py:266-271) on the CPU SIMT emulator: the product's encoder
engine (csrc/encoder.cpp + kernels/enc.h, compiled unchanged) against golden codes from the live transformers models, stage
by stage against the oracle restatement, plus the engine's error paths."""
import numpy as np
import pytest

import synthetic as syn
from oracle import encoder_ref as er
from neutts import _hip
from common import check_encoder_codes, load_encoder_fixture, make_encoder_engine

LAT_TOL = 2e-4      # fp32 pipeline vs the fp32 reference; measured on the emulator: 1e-5 .. 3e-5


@pytest.fixture(scope="module")
def tiny(emu_lib):
    z, cfg, w = load_encoder_fixture("encoder_tiny")
    return z, cfg, w, make_encoder_engine(cfg, w, emu_lib)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_tiny_codes_vs_live_hf_golden(tiny, i):
    """Ragged clip / exact hop multiple (the reference still appends a whole hop) / shorter than one hop."""
    z, cfg, w, eng = tiny
    wav = syn.synthetic_speech(int(z[f"n_samples_{i}"]), int(z[f"clip_seed_{i}"]))
    codes = eng.encode(wav)
    assert codes.shape[0] == wav.size // cfg.hop + 1
    f = eng.read_stage("features")
    assert np.abs(f - z[f"features_{i}"]).max() <= 2e-4          # log-mel after per-bin normalisation, O(1) values
    check_encoder_codes(cfg, codes, eng.read_stage("latents"), z[f"codes_{i}"], z[f"latents_{i}"], LAT_TOL, f" tiny[{i}]")


def test_each_stage_vs_oracle(tiny):
    """Every fused stage against its own restated stage at its own scale (a whole-pipeline tolerance would hide an O(1)
    error in one branch behind the tanh of the quantiser)."""
    z, cfg, w, eng = tiny
    wav = syn.synthetic_speech(5000, 9)
    codes, parts = er.encode(cfg, w, wav, return_parts=True)
    got = eng.encode(wav)
    H = cfg.sem_hidden
    cat = eng.read_stage("concat")
    for name, a, b in (("features", eng.read_stage("features"), parts["features"]), ("semantic", cat[:, :H], parts["semantic"]),
                       ("acoustic", cat[:, H:], parts["acoustic"]), ("fc", eng.read_stage("fc"), parts["fc"])):
        rel = np.abs(a - b).max() / np.abs(b).max()
        print(f"stage {name}: max |err| / max |ref| = {rel:.2e}")
        assert rel <= 1e-4, (name, rel)
    check_encoder_codes(cfg, got, eng.read_stage("latents"), codes, parts["latents"], LAT_TOL, " stage test")
    assert np.array_equal(eng.encode(wav), got)                   # deterministic, state-free between calls


def test_error_paths(emu_lib):
    cfg = syn.EncoderConfig.tiny()
    w = syn.make_encoder_weights(cfg, 0)
    d = cfg.to_dict()
    d["max_samples"] = 4000
    eng = _hip.EncoderEngine(d, 0, emu_lib)
    with pytest.raises(_hip.NeuTTSHipError, match="not finalised"):
        eng.encode(np.zeros(400, np.float32))
    sd = {k: v.numpy() for k, v in w.items()}
    missing = {"fc_encoder.bias", "acoustic_encoder.block.2.res_unit2.snake1.act.beta", "semantic_encoder.encoder.layers.1.self_attn.linear_k.weight"}
    with pytest.raises(_hip.NeuTTSHipError) as ei:
        eng.load_state_dict({k: v for k, v in sd.items() if k not in missing})
    for k in missing:
        assert k in str(ei.value)                                  # every missing tensor is named, not just the first
    bad = dict(sd)
    bad["semantic_adapter.conv2.weight"] = bad["semantic_adapter.conv2.weight"][:, :, :2]
    with pytest.raises(_hip.NeuTTSHipError, match="unexpected shape.*semantic_adapter.conv2.weight"):
        eng.load_state_dict(bad)
    eng.load_state_dict(sd)
    with pytest.raises(ValueError, match="exceeds max_samples"):
        eng.encode(np.zeros(4001, np.float32))
    with pytest.raises(ValueError, match="empty"):
        eng.encode(np.zeros(0, np.float32))
    assert eng.encode(np.zeros(4000, np.float32)).shape == (4000 // cfg.hop + 1,)
    d2 = dict(d)
    d2["ratios"] = [2, 2, 4, 4, 4]
    with pytest.raises(_hip.NeuTTSHipError, match="ratios must multiply"):
        _hip.EncoderEngine(d2, 0, emu_lib)
