"""The drop-in class surface (ref:neutts/neutts.py:73-465) on a real MI355X through libneutts_hip.so: the reference's
own assertions (ref:tests/test_neutts.py:55-58, :78-85) and id / waveform / chunk-for-chunk streaming equivalence with
the oracle pipeline -- the same test bodies that tests/test_emu_neutts_class.py runs on the SIMT emulator, here at
NeuTTS-Air width (hidden 896, 14/2 heads, 3 layers) so that the real kernels' tile paths are the ones exercised."""
import numpy as np
import pytest
import torch

from oracle import backbone_ref as br
from oracle import codec_ref as cr
from neutts import _hip
import test_emu_neutts_class as cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tts(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _hip.load_library(hip_lib)
    return cases.build_tts(
        hip_lib, bcfg=lambda v: br.BackboneConfig(vocab_size=v, hidden_size=896, intermediate_size=1216, num_layers=3),
        max_batch=4, max_context=256, max_prefill_tokens=1024, seed=33)


# the shared bodies: collected here under the gpu mark, resolved against THIS module's `tts` fixture
test_surface_matches_reference = cases.test_surface_matches_reference
test_infer_smoke_and_equivalence = cases.test_infer_smoke_and_equivalence
test_decode_without_speech_tokens_raises = cases.test_decode_without_speech_tokens_raises
test_infer_stream_matches_reference_windowing = cases.test_infer_stream_matches_reference_windowing
test_infer_batch_shares_prompt_beginnings_and_matches_single_inference = \
    cases.test_infer_batch_shares_prompt_beginnings_and_matches_single_inference
test_infer_batch_over_an_engine_gang_matches_single_inference = cases.test_infer_batch_over_an_engine_gang_matches_single_inference
test_infer_stream_batch_equals_single_streams = cases.test_infer_stream_batch_equals_single_streams
test_encode_reference_on_the_encoder_engine = cases.test_encode_reference_on_the_encoder_engine


def test_ids_to_codes_and_decode_paths_agree(tts):
    """`_decode` (ref:neutts/neutts.py:273-295): the reference's string -> regex route and the id route
    (id - id(<|speech_0|>), range mask) select the same codes, specials and text ids are dropped, and both reach
    the HIP codec with identical results."""
    bcfg, bw, ccfg, cw, tok, eos = tts._oracle
    n_codes = int(np.prod(ccfg.levels))
    rng = np.random.default_rng(7)
    codes = rng.integers(0, n_codes, size=40).tolist()
    ids = []
    for i, c in enumerate(codes):
        ids.append(tok.speech_base + c)
        if i % 7 == 3:
            ids += [65, eos, tok.convert_tokens_to_ids("<|TEXT_PROMPT_END|>")]      # text byte + specials: dropped
    assert tts._ids_to_codes(ids) == codes
    base, tts._speech_base = tts._speech_base, None                                 # the tokenizer.decode + regex route
    try:
        assert tts._ids_to_codes(ids) == codes
    finally:
        tts._speech_base = base
    a = tts._decode_ids(ids)
    b = tts._decode(tok.decode(ids))
    want = cr.decode_code(ccfg, cw, torch.tensor(codes)[None, None, :])[0, 0].numpy()
    assert np.array_equal(a, b) and a.shape == want.shape
    err = cases.rms(a - want)
    print(f"class-level codec hand-off: RMS error {err:.3e} (signal RMS {cases.rms(want):.3e})")
    assert err <= 1e-3


def test_sampling_default_call_runs_and_stays_in_vocab(tts):
    """The reference's default call samples (do_sample=True, top_k=50, temperature=1.0, ref:neutts/neutts.py:338-347)."""
    tts.do_sample = True
    try:
        ref_codes = [3, 77, 200, 5]
        audio = tts.infer("Sampled.", ref_codes, "So I'm live.")
        assert isinstance(audio, np.ndarray) and audio.dtype == np.float32 and len(audio) % tts.hop_length == 0
        assert np.isfinite(audio).all()
    finally:
        tts.do_sample = False
test_device_side_code_handoff = cases.test_device_side_code_handoff


def test_infer_at_air_geometry_with_the_dave_reference_voice(hip_lib):
    """BASELINE.json configs[0]'s workload on the GPU: one utterance, reference voice ref:samples/dave.pt
    (372 codes), greedy, through `_apply_chat_template` -> `generate_codes` -> `infer()` at NeuTTS-Air's layer geometry (24
    layers) and NeuCodec's decoder geometry -- against tests/golden/infer_air_dave.npz, which oracle/gen_golden_infer.py made
    with transformers' generate and the codec restatement from the reference's own lines.  Prompt ids identical; generated ids
    identical (or up to a <= 2-ulp tie of transformers' own logits); waveform within BASELINE's 1e-3 RMS of the fp32 oracle's."""
    import os
    from common import GOLD, rms
    z = np.load(os.path.join(GOLD, "infer_air_dave.npz"), allow_pickle=True)
    t = cases.build_tts(hip_lib, bcfg=lambda v: br.BackboneConfig(vocab_size=v), ccfg=cr.CodecConfig.neucodec(), max_batch=1,
                        max_context=1024, max_prefill_tokens=1024, seed=int(z["seed_backbone"]))
    dave = torch.tensor(z["dave_codes"])
    prompt = t._apply_chat_template(dave, str(z["ref_text"]), str(z["text"]))
    assert prompt == z["prompt"].tolist()
    t.max_context, t.min_new_tokens = len(prompt) + int(z["n_new"]), int(z["min_new"])
    got, want, tv, ti = t.generate_codes([prompt])[0], z["ids"].tolist(), z["topv"], z["topi"]
    n = min(len(got), len(want))
    k = next((i for i in range(n) if got[i] != want[i]), None)
    if k is not None:
        band = 2.0 * 2.0 ** (np.floor(np.log2(abs(tv[k][0]))) - 7)
        assert tv[k][0] - tv[k][1] <= band and got[k] in ti[k][:2].tolist(), (k, got[k], want[k], tv[k], ti[k])
        pytest.skip(f"ids equal up to a tie of transformers' own logits at step {k}: the waveform comparison needs equal codes")
    assert abs(len(got) - len(want)) <= 1 and n >= int(z["min_new"])      # HF may or may not append the EOS it stopped on
    assert t._ids_to_codes(got) == z["codes"].tolist()[:len(t._ids_to_codes(got))]
    audio = t.infer(str(z["text"]), dave, str(z["ref_text"]))
    ref = z["wav"][:len(audio)]
    assert audio.dtype == np.float32 and len(audio) == t.hop_length * len(t._ids_to_codes(got)) and abs(len(audio) - len(z["wav"])) <= t.hop_length
    err, sig = rms(audio - ref), rms(ref)
    print(f"infer() at Air geometry, dave reference: {len(got)} ids identical to transformers', waveform RMS error {err:.3e} at signal RMS {sig:.3e}")
    assert err <= 1e-3 and err <= 2e-2 * sig
