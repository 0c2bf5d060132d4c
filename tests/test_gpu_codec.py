"""NeuCodec decoder parity on a real MI355X through the C-ABI: golden waveforms produced by transformers' xcodec2
decoder at NeuCodec geometry (fp32), tolerance = BASELINE.json's "waveform RMS within 1e-3"."""
import numpy as np
import pytest
import torch

from oracle import codec_ref as cr
from neutts import _hip
from common import load_codec_fixture, make_codec_engine, rms

pytestmark = pytest.mark.gpu

# DEFAULT engine (ABI 9): fp16 GEMM operands (v_mfma_f32_16x16x32_f16, fp32 accumulate) against an fp32 reference.  Predicted by rounding
# exactly those operands in the oracle (tools/codec_operand_sim.py): 8.0e-4 relative at NeuCodec geometry from the GEMM operands (bf16 operands: 7.4e-3); the single-term fp16 ISTFT brings the measured total to 9.5e-4.  Measured on
# MI355X: see profiles/r06*_pytest_gpu*.log; the bounds are VERDICT r5's bar for the default engine (relative <= 2.5e-3, and 1e-3 ABSOLUTE at the
# amplitude of a loud voice, signal rms 0.29 -- test_neucodec_error_budget...).
REL_BOUND = 2.5e-3
ABS_BOUND = 5e-5        # at the goldens' signal rms 1.6e-2; BASELINE.json asks for 1e-3
REL_BOUND_BF16 = 0.015  # precision = "bf16" (rounds 1-5's default): measured 6.9e-3 .. 7.2e-3


@pytest.fixture(scope="module")
def lib(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _hip.load_library(hip_lib)
    return hip_lib


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("resident", ["1", "0"])
def test_codec_tiny(lib, resident, precision, monkeypatch):
    monkeypatch.setenv("NTTS_CODEC_ATTN_RESIDENT", resident)
    monkeypatch.setenv("NTTS_CODEC_GN_REG", resident)          # likewise GroupNorm: utterance slice in registers / two-pass kernel   # resident single-sweep attention kernel (default) / paged two-sweep kernel
    z, cfg, w = load_codec_fixture("codec_tiny")
    eng = make_codec_engine(cfg, w, lib, precision=precision)
    codes = [z["codes_0"][0, 0].tolist(), z["codes_1"][0, 0].tolist(), z["codes_0"][1, 0].tolist()]
    gold = [z["wav_0"][0, 0], z["wav_1"][0, 0], z["wav_0"][1, 0]]
    wavs = eng.decode(codes)
    for wv, g in zip(wavs, gold):
        assert wv.shape == g.shape and not np.isnan(wv).any()
        print(f"codec_tiny, {precision} operands: RMS error {rms(wv - g):.3e}, signal RMS {rms(g):.3e}, relative {rms(wv - g) / rms(g):.3e}")
        # bf16 measured: 4.5e-4 .. 5.9e-4 at signal RMS 7e-2 .. 8.6e-2 (relative 5.4e-3 .. 8.5e-3); fp16 on the emulator: 7e-4 .. 1.25e-3 relative
        assert rms(wv - g) <= 1e-3 and rms(wv - g) <= (2.5e-3 if precision == "fp16" else 0.017) * rms(g), (rms(wv - g), rms(g))
    assert np.array_equal(eng.decode([codes[1]])[0], wavs[1])


@pytest.fixture(scope="module")
def neucodec(lib):
    z, cfg, w = load_codec_fixture("codec_neucodec")
    eng = make_codec_engine(cfg, w, lib, max_frames=512, max_rows=256 * 256 + 64)
    return z, cfg, w, eng


def test_neucodec_geometry_vs_golden(neucodec):
    """hop 480 / n_fft 1920 / 1024 x 12 layers; set 1 = the first 100 codes of the reference's own sample voice
    (ref:samples/dave.pt)."""
    z, cfg, w, eng = neucodec
    for i in range(int(z["n"])):
        codes = z[f"codes_{i}"][0, 0].tolist()
        wv = eng.decode([codes])[0]
        g = z[f"wav_{i}"][0, 0]
        assert wv.shape == g.shape == (480 * len(codes),)
        err, sig = rms(wv - g), rms(g)
        print(f"neucodec golden set {i}: {len(codes)} frames, RMS error {err:.3e}, signal RMS {sig:.3e}, relative {err / sig:.3e}")
        assert err <= ABS_BOUND, (i, err, sig)            # BASELINE.json: waveform RMS within 1e-3 (fp32 reference)
        assert err <= REL_BOUND * sig, (i, err, sig)      # and relative to the signal: 2x the error measured on MI355X


def test_neucodec_batch256_properties(neucodec):
    """BASELINE batch: 256 utterances x 250 frames.  Identical code sequences give identical waveforms wherever
    they sit in the batch, a shorter utterance inside the batch equals its stand-alone decode, and an oracle
    spot-check of one row holds the 1e-3 bound."""
    z, cfg, w, eng = neucodec
    rng = np.random.default_rng(5)
    base = [rng.integers(0, 65536, size=250).tolist() for _ in range(4)]
    codes = [base[i % 4] for i in range(255)] + [base[0][:97]]
    wavs = eng.decode(codes)
    for i in range(255):
        assert np.array_equal(wavs[i], wavs[i % 4]), i
    assert np.array_equal(wavs[255], eng.decode([base[0][:97]])[0])
    ref = cr.decode_code(cfg, w, torch.tensor(base[1], dtype=torch.long)[None, None, :])[0, 0].numpy()
    err, sig = rms(wavs[1] - ref), rms(ref)
    print(f"batch-256 row vs oracle: RMS error {err:.3e}, signal RMS {sig:.3e}, relative {err / sig:.3e}")
    assert err <= ABS_BOUND and err <= REL_BOUND * sig


def test_codec_long_utterance_paged_attention(lib):
    import test_emu_codec
    test_emu_codec.check_long_utterance(lib)


def test_neucodec_error_budget_by_stage_and_at_realistic_amplitude(neucodec):
    """WHERE the waveform error comes from, and what it is at the amplitude of a real voice -- on the DEFAULT engine.
    (1) Stage taps (ntts_codec_read_stage, ABI 6) against the oracle's taps (oracle/codec_ref.py decode_code(taps=...)): relative RMS
        error of the fp32 residual stream after the stem, the prior ResNet blocks, the 12 transformer layers and the post ResNet
        blocks.  Every GEMM runs 16-bit operands (weights AND activations, fp32 accumulation) against a reference that is fp32
        throughout; with bf16 operands (8 significant bits) that was ~1.1e-3 relative per rounded operand, accumulating over ~60 GEMMs in
        series to 7.0e-3 (71 % of the squared error inside the transformer layers); fp16 operands (11 bits, the default since ABI 9) carry an
        eighth of that per operand at the same matrix-core rate -- tools/codec_operand_sim.py reproduces both totals on the CPU.
    (2) The bound is RELATIVE: the same codes with the magnitude bias of the ISTFT head raised by ln 6 / ln 18 -- every STFT magnitude,
        hence every sample, exactly 6x / 18x larger in the reference -- give signal rms ~0.1 (a real voice) / ~0.29 (a LOUD one): relative
        error unchanged, absolute error inside BASELINE's 1e-3 on the default engine at both (the bf16 engine: 2.0e-3 at 18x)."""
    z, cfg, w, eng = neucodec
    codes = z["codes_1"][0, 0].tolist()                      # ref:samples/dave.pt[:100]
    eng.set_debug(True)
    try:
        wv = eng.decode([codes])[0]
        got = [eng.read_stage(k) for k in range(4)]
    finally:
        eng.set_debug(False)
    taps = {}
    ref = cr.decode_code(cfg, w, torch.tensor(codes, dtype=torch.long)[None, None, :], taps=taps)[0, 0].numpy()
    rel = []
    for k, name in enumerate(_hip.CodecEngine.STAGES):
        want = taps[name][0].numpy()
        assert got[k].shape == want.shape, (name, got[k].shape, want.shape)
        rel.append(rms(got[k] - want) / rms(want))
    err, sig = rms(wv - ref), rms(ref)
    print("codec error budget (default engine, fp16 operands), relative RMS of the residual stream vs the fp32 oracle: "
          + ", ".join(f"{n} {r:.2e}" for n, r in zip(_hip.CodecEngine.STAGES, rel)) + f"; waveform {err / sig:.2e} (RMS error {err:.2e} at signal RMS {sig:.2e})")
    # bf16 operands measured 2.29e-3, 3.84e-3, 5.49e-3, 5.60e-3 (profiles/r04d_pytest_gpu_codec.log); fp16 bars = a quarter of 1.5x that
    assert rel[0] <= 9e-4 and rel[1] <= 1.5e-3 and rel[2] <= 2e-3 and rel[3] <= 2e-3, rel
    assert err <= REL_BOUND * sig
    # ---- the same utterance at the amplitude of a real voice (6x) and of a loud one (18x), default engine
    for gain, lo, hi in ((6.0, 0.08, 0.13), (18.0, 0.2, 0.45)):
        wg = dict(w)
        b = w["decoder.head.linear.bias"].clone()
        b[: b.numel() // 2] += float(np.log(gain))          # magnitude half of the head's output (hf:models/xcodec2/modeling_xcodec2.py:771-773)
        wg["decoder.head.linear.bias"] = b
        engg = make_codec_engine(cfg, wg, neucodec_lib(eng), max_frames=128, max_rows=512)
        wvg = engg.decode([codes])[0]
        refg = cr.decode_code(cfg, wg, torch.tensor(codes, dtype=torch.long)[None, None, :])[0, 0].numpy()
        errg, sigg = rms(wvg - refg), rms(refg)
        print(f"the same codes {gain:g}x louder: RMS error {errg:.2e} at signal RMS {sigg:.2e}, relative {errg / sigg:.2e}")
        assert lo <= sigg <= hi and errg <= 1e-3 and errg <= REL_BOUND * sigg
        engg.close()


def test_neucodec_bf16_operands_option(neucodec):
    """precision = "bf16" (rounds 1-5's default, kept for weights outside fp16's range): the goldens inside its own bars, and the error it makes
    next to the default engine's on the same utterances (what ABI 9 changed the default for)."""
    z, cfg, w, eng = neucodec
    lo = make_codec_engine(cfg, w, neucodec_lib(eng), max_frames=128, max_rows=512, precision="bf16")
    for i in range(int(z["n"])):
        codes = z[f"codes_{i}"][0, 0].tolist()
        if len(codes) > 128:
            continue
        g = z[f"wav_{i}"][0, 0]
        e_bf, e_h = rms(lo.decode([codes])[0] - g) / rms(g), rms(eng.decode([codes])[0] - g) / rms(g)
        print(f"neucodec golden set {i}: relative rms error bf16 operands {e_bf:.2e}, fp16 operands (default) {e_h:.2e}")
        assert e_bf <= REL_BOUND_BF16 and e_h <= 0.35 * e_bf
    lo.close()


def neucodec_lib(eng):
    return eng.lib._name


def test_neucodec_high_precision_holds_the_bound_at_full_scale(neucodec):
    """precision = "high" (split bf16 GEMM operands, ABI 8) at NeuCodec geometry.  (1) The golden utterances and
    the stage taps: relative rms error of the waveform <= 2.5e-3 (default engine: 7.0e-3), the residual stream after the 12 layers
    likewise a fraction of the default's.  (2) The same codes with the ISTFT magnitudes raised 18x -- signal rms ~0.3, a LOUD voice -- stay
    inside BASELINE's 1e-3 ABSOLUTE, where the default engine (7e-3 relative) is at 2e-3.  (3) What it costs: the 256 x 250-frame batch
    timed on both engines (printed; DESIGN.md section 2 quotes it -- an option, not the default, unless it is within 5 % of a batch)."""
    z, cfg, w, eng = neucodec
    lib = neucodec_lib(eng)
    hi = make_codec_engine(cfg, w, lib, max_frames=512, max_rows=256 * 256 + 64, precision="high")
    bf = make_codec_engine(cfg, w, lib, max_frames=512, max_rows=256 * 256 + 64, precision="bf16")
    for i in range(int(z["n"])):
        codes = z[f"codes_{i}"][0, 0].tolist()
        g = z[f"wav_{i}"][0, 0]
        e_lo, e_hi = rms(bf.decode([codes])[0] - g) / rms(g), rms(hi.decode([codes])[0] - g) / rms(g)
        print(f"neucodec golden set {i}: relative rms error bf16 operands {e_lo:.2e}, split operands {e_hi:.2e}")
        assert e_hi <= 2.5e-3 and e_hi <= 0.45 * e_lo, (i, e_lo, e_hi)
    codes = z["codes_1"][0, 0].tolist()
    hi.set_debug(True)
    try:
        hi.decode([codes])
        got = [hi.read_stage(k) for k in range(4)]
    finally:
        hi.set_debug(False)
    taps = {}
    cr.decode_code(cfg, w, torch.tensor(codes, dtype=torch.long)[None, None, :], taps=taps)
    rel = [rms(got[k] - taps[name][0].numpy()) / rms(taps[name][0].numpy()) for k, name in enumerate(_hip.CodecEngine.STAGES)]
    print("high precision, relative rms of the residual stream vs the fp32 oracle: " + ", ".join(f"{n} {r:.2e}" for n, r in zip(_hip.CodecEngine.STAGES, rel)))
    assert max(rel) <= 2.5e-3, rel
    # (2) a loud voice: every magnitude 18x
    w18 = dict(w)
    b = w["decoder.head.linear.bias"].clone()
    b[: b.numel() // 2] += float(np.log(18.0))
    w18["decoder.head.linear.bias"] = b
    ref = cr.decode_code(cfg, w18, torch.tensor(codes, dtype=torch.long)[None, None, :])[0, 0].numpy()
    out = {}
    for prec in ("bf16", "high", "fp16"):
        e18 = make_codec_engine(cfg, w18, lib, max_frames=128, max_rows=512, precision=prec)
        out[prec] = rms(e18.decode([codes])[0] - ref)
        e18.close()
    print(f"the same codes 18x louder (signal rms {rms(ref):.3f}): rms error bf16 operands {out['bf16']:.2e}, split operands {out['high']:.2e}, fp16 operands (default) {out['fp16']:.2e}")
    assert 0.2 <= rms(ref) <= 0.45 and out["high"] <= 1e-3 and out["fp16"] <= 1e-3 and out["bf16"] > 1e-3
    # (3) cost at the benchmark's batch
    rng = np.random.default_rng(5)
    batch = [rng.integers(0, 65536, size=250).tolist() for _ in range(256)]
    ms = {}
    for name, e in (("bf16", bf), ("high", hi), ("fp16", eng)):
        e.decode(batch)
        e.decode(batch)
        ms[name] = e.last_timing()
    print(f"codec pass, 256 x 250 frames: fp16 operands (default) {ms['fp16']:.1f} ms, bf16 operands {ms['bf16']:.1f} ms, split operands {ms['high']:.1f} ms ({ms['high'] / ms['bf16']:.2f}x)")
    assert ms["fp16"] <= 1.06 * ms["bf16"]       # the default holds the bound at the bf16 engine's cost (<= 5 % of a batch)
    hi.close()
    bf.close()


def test_verify_checkpoint_codec_half_on_a_neucodec_style_state_dict(neucodec, tmp_path, capsys):
    """tools/verify_checkpoint.py --codec on what can be built offline: the synthetic NeuCodec-geometry decoder weights re-keyed into the
    original `neucodec` layout (fused c_attn) and saved as a .pt state dict.  The tool maps the keys (strictly), loads the
    decoder into the codec engine and compares decode_code with transformers' Xcodec2 modules filled with the same tensors."""
    import os
    import sys
    import re
    from test_host_logic import _neucodec_style
    z, cfg, w, eng = neucodec
    path = str(tmp_path / "neucodec_style.pt")
    torch.save(_neucodec_style(w), path)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import verify_checkpoint as vc
    vc.verify_codec(path, "cuda:0", neucodec_lib(eng))       # (returns False here: the synthetic dict has no encoder tensors, and says so)
    out = capsys.readouterr().out
    print(out)
    assert "decoder key map: all" in out and "encoder key map MISMATCH" in out
    rel = float(re.search(r"relative ([0-9.e+-]+); bar", out).group(1))
    assert rel <= REL_BOUND
