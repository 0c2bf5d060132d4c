"""NeuCodec decoder engine on the CPU SIMT emulator vs the golden waveforms / the oracle."""
import numpy as np
import pytest
import torch

from oracle import codec_ref as cr
from neutts import _hip
from common import load_codec_fixture, make_codec_engine, rms


# relative rms bound of the waveform by GEMM-operand format (fp32 golden waveforms; tools/codec_operand_sim.py predicts 8e-4 / 7e-3 at NeuCodec geometry)
REL = {"fp16": 2.5e-3, "bf16": 0.02}


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("resident", ["1", "0"])
def test_codec_tiny_vs_golden_ragged_batch(emu_lib, resident, precision, monkeypatch):
    """Both attention kernels of the decoder layers: K / V^T resident in LDS with ONE online-softmax sweep (utterances of up to
    256 frames: the default), and the paged two-sweep kernel behind the V^T transpose pass (longer utterances; forced here); both 16-bit
    operand formats: IEEE half (ABI 9 default: v_mfma_f32_16x16x32_f16, 11 significant bits) and bf16 (rounds 1-5)."""
    monkeypatch.setenv("NTTS_CODEC_ATTN_RESIDENT", resident)
    monkeypatch.setenv("NTTS_CODEC_GN_REG", resident)          # likewise GroupNorm: utterance slice in registers / two-pass kernel
    z, cfg, w = load_codec_fixture("codec_tiny")
    eng = make_codec_engine(cfg, w, emu_lib, precision=precision)
    codes = [z["codes_0"][0, 0].tolist(), z["codes_1"][0, 0].tolist(), z["codes_0"][1, 0].tolist()]
    gold = [z["wav_0"][0, 0], z["wav_1"][0, 0], z["wav_0"][1, 0]]
    wavs = eng.decode(codes)                          # 37-, 5- and 37-frame utterances in ONE call
    for wv, g in zip(wavs, gold):
        assert wv.dtype == np.float32 and wv.shape == g.shape and not np.isnan(wv).any()
        print(f"codec tiny, {precision} operands: rms error {rms(wv - g):.2e}, relative {rms(wv - g) / rms(g):.2e}")
        assert rms(wv - g) <= 1e-3, rms(wv - g)       # BASELINE.json: waveform RMS within 1e-3 of the fp32 reference
        assert rms(wv - g) <= REL[precision] * rms(g)
    # batch invariance: decoding an utterance alone gives the same samples as inside the ragged batch
    alone = eng.decode([codes[1]])[0]
    assert np.array_equal(alone, wavs[1])


def test_codec_fp16_weight_out_of_range_is_refused(emu_lib):
    """precision = fp16 checks every GEMM weight against the half range at finalize (NTTS_EINVAL names the alternative); bf16 takes the same dict."""
    z, cfg, w = load_codec_fixture("codec_tiny")
    w = dict(w)
    big = w["decoder.layers.0.mlp.fc1.weight"].clone()
    big[0, 0] = 1.0e5
    w["decoder.layers.0.mlp.fc1.weight"] = big
    with pytest.raises(_hip.NeuTTSHipError, match="fp16"):
        make_codec_engine(cfg, w, emu_lib)
    make_codec_engine(cfg, w, emu_lib, precision="bf16").close()


def test_codec_splits_calls_when_rows_exceed_workspace(emu_lib):
    z, cfg, w = load_codec_fixture("codec_tiny")
    eng = make_codec_engine(cfg, w, emu_lib, max_frames=40, max_rows=50)   # room for one 37-frame utterance per call
    codes = [z["codes_0"][0, 0].tolist(), z["codes_0"][1, 0].tolist()]
    wavs = eng.decode(codes)
    assert rms(wavs[0] - z["wav_0"][0, 0]) <= 1e-3 and rms(wavs[1] - z["wav_0"][1, 0]) <= 1e-3


def test_codec_error_paths(emu_lib):
    z, cfg, w = load_codec_fixture("codec_tiny")
    eng = _hip.CodecEngine(dict(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_layers=cfg.num_layers,
                                num_heads=cfg.num_heads, quantization_dim=cfg.quantization_dim, levels=list(cfg.levels),
                                hop_length=cfg.hop_length, max_frames=16, max_rows=64), 0, emu_lib)
    with pytest.raises(_hip.NeuTTSHipError):             # not finalised
        eng.decode([[1, 2, 3]])
    sd = {k: v.numpy() for k, v in w.items()}
    missing = dict(sd)
    missing.pop("decoder.norm.bias")
    with pytest.raises(_hip.NeuTTSHipError):             # a tensor is missing -> finalize fails loudly
        eng.load_state_dict(missing)
    eng.load_state_dict(sd)
    with pytest.raises(_hip.NeuTTSHipError):             # code out of range
        eng.decode([[1, 2, 10 ** 6]])
    with pytest.raises(_hip.NeuTTSHipError):             # too many frames
        eng.decode([list(range(17))])
    assert eng.decode([[1, 2, 3]])[0].shape == (3 * cfg.hop_length,)


def test_codec_pinned_output_views(emu_lib):
    """reuse_output=True: same samples, returned as views into the engine's page-locked staging buffer."""
    z, cfg, w = load_codec_fixture("codec_tiny")
    eng = make_codec_engine(cfg, w, emu_lib)
    codes = [z["codes_0"][0, 0].tolist(), z["codes_1"][0, 0].tolist()]
    a = eng.decode(codes)
    b = eng.decode(codes, reuse_output=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert not b[0].flags["OWNDATA"]


def check_long_utterance(lib):
    """300 frames (> 256: the paged attention kernel even with the resident one enabled) next to a 200-frame one (resident kernel
    when decoded alone, paged inside this batch: the longest utterance of a call picks the kernel) against the oracle."""
    z, cfg, w = load_codec_fixture("codec_tiny")
    eng = make_codec_engine(cfg, w, lib, max_frames=320, max_rows=700)
    rng = np.random.default_rng(11)
    n_codes = int(np.prod(cfg.levels))
    codes = [rng.integers(0, n_codes, size=300).tolist(), rng.integers(0, n_codes, size=200).tolist()]
    wavs = eng.decode(codes)
    alone = eng.decode([codes[1]])[0]
    for c, wv in zip(codes, wavs):
        ref = cr.decode_code(cfg, w, torch.tensor(c, dtype=torch.long)[None, None, :])[0, 0].numpy()
        assert wv.shape == ref.shape and rms(wv - ref) <= 1e-3 and rms(wv - ref) <= 2.5e-3 * rms(ref), (len(c), rms(wv - ref), rms(ref))
    ref1 = cr.decode_code(cfg, w, torch.tensor(codes[1], dtype=torch.long)[None, None, :])[0, 0].numpy()
    assert rms(alone - ref1) <= 1e-3 and rms(alone - wavs[1]) <= 1e-3      # two kernels, one answer within the bound (not bit-identical)


def test_codec_long_utterance_paged_attention(emu_lib):
    check_long_utterance(emu_lib)


@pytest.mark.parametrize("resident", ["1", "0"])
def test_codec_high_precision_split_operands(emu_lib, resident, monkeypatch):
    """precision = "high" (ABI 8 ntts_codec_config.precision = 1): every GEMM operand as a split bf16 pair, K-concatenated [xh | xl | xh] x
    [wh | wh | wl] -- the stem / ResNet convs over overlapping rows (segments per tap), QKV, o_proj, fc1 (whose SiLU epilogue emits the split
    row for fc2), fc2, the ISTFT head's linear; both attention kernels and both GroupNorm kernels write split rows.  Against the fp32 golden
    waveforms the error must drop well below the bf16-operand engine's on the same utterances (what is left is the attention's own bf16
    q / k / v / P and the bf16 QKV output), ragged batch and batch invariance as for the default engine."""
    monkeypatch.setenv("NTTS_CODEC_ATTN_RESIDENT", resident)
    monkeypatch.setenv("NTTS_CODEC_GN_REG", resident)
    z, cfg, w = load_codec_fixture("codec_tiny")
    codes = [z["codes_0"][0, 0].tolist(), z["codes_1"][0, 0].tolist(), z["codes_0"][1, 0].tolist()]
    gold = [z["wav_0"][0, 0], z["wav_1"][0, 0], z["wav_0"][1, 0]]
    lo = make_codec_engine(cfg, w, emu_lib, precision="bf16").decode(codes)
    eng = make_codec_engine(cfg, w, emu_lib, precision="high")
    hi = eng.decode(codes)
    for a, b, g in zip(lo, hi, gold):
        e_lo, e_hi = rms(a - g) / rms(g), rms(b - g) / rms(g)
        print(f"codec tiny: relative rms error bf16 operands {e_lo:.2e}, split operands {e_hi:.2e}")
        assert b.shape == g.shape and e_hi <= 0.4 * e_lo and e_hi <= 2.5e-3, (e_lo, e_hi)
    assert np.array_equal(eng.decode([codes[1]])[0], hi[1])
