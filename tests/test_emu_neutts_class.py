"""Drop-in class surface (ref:neutts/neutts.py:73-465) over the emulated engines: the reference's own smoke
assertions (ref:tests/test_neutts.py:55-58, :78-85) + id-level equivalence with the oracle pipeline."""
import re

import numpy as np
import pytest
import torch

from oracle import backbone_ref as br
from oracle import codec_ref as cr
import synthetic as syn
from common import check_encoder_codes, engine_cfg, rms


FakeTokenizer, FakePhonemizer = syn.ByteTokenizer, syn.LowercasePhonemizer   # stand-ins for the HF tokenizer / espeak (synthetic.py)


def build_tts(lib, bcfg=None, ccfg=None, max_batch=2, max_context=256, max_prefill_tokens=512, seed=31):
    """A NeuTTS object over in-memory synthetic weights on library `lib` (the SIMT-emulator build here, the real
    libneutts_hip.so in tests/test_gpu_neutts_class.py) + the oracle-side copies of everything it was built from."""
    from neutts import NeuTTS
    ccfg = ccfg or cr.CodecConfig.tiny()
    n_codes = int(np.prod(ccfg.levels))
    tok = FakeTokenizer(n_codes)
    bcfg = bcfg(tok.vocab_size) if callable(bcfg) else br.BackboneConfig.tiny(vocab_size=tok.vocab_size, num_layers=1)
    # greedy decoding walks a permutation of the SPEECH tokens (synthetic._make_walk): it emits codec codes, a new one every step
    deep = dict(walk_gain=8.0, walk_scale=8.0) if bcfg.num_layers > 8 else dict(walk_gain=4.0)     # (24 layers add more to the stream: oracle/gen_golden_infer.py)
    bw = br.make_weights(bcfg, seed, walk_range=(tok.speech_base, tok.speech_base + n_codes), **deep)
    cw = cr.make_weights(ccfg, 2)
    ecfg = syn.EncoderConfig.tiny()      # reference encoder (encode_reference): FSQ levels independent of the tiny decoder's
    ew = syn.make_encoder_weights(ecfg, 4)
    eos = tok.convert_tokens_to_ids("<|SPEECH_GENERATION_END|>")
    t = NeuTTS(
        backbone_repo=dict(config=engine_cfg(bcfg, max_context=max_context, max_prefill_tokens=max_prefill_tokens),
                           state_dict={k: v.numpy() for k, v in bw.items()}, inv_freq=br.rope_inv_freq(bcfg).numpy(),
                           tokenizer=tok, speech_base=tok.speech_base, eos_token_id=eos),
        backbone_device="cuda",
        codec_repo=dict(config=dict(hidden_size=ccfg.hidden_size, intermediate_size=ccfg.intermediate_size,
                                    num_layers=ccfg.num_layers, num_heads=ccfg.num_heads,
                                    quantization_dim=ccfg.quantization_dim, levels=list(ccfg.levels),
                                    hop_length=ccfg.hop_length, max_frames=256, max_rows=1024),
                        state_dict={k: v.numpy() for k, v in cw.items()},
                        encoder=dict(config=dict(ecfg.to_dict(), max_samples=20000), state_dict={k: v.numpy() for k, v in ew.items()})),
        codec_device="cuda", lib_path=lib, do_sample=False, max_batch=max_batch)
    t.phonemizer = FakePhonemizer()
    t.max_context = 120          # keep the emulated run short
    t.min_new_tokens = 5
    t._oracle = (bcfg, bw, ccfg, cw, tok, eos)
    t._oracle_encoder = (ecfg, ew)
    return t


@pytest.fixture(scope="module")
def tts(emu_lib):
    return build_tts(emu_lib)


def test_surface_matches_reference(tts):
    for name, val in dict(sample_rate=24000, streaming_overlap_frames=1, streaming_frames_per_chunk=25,
                          streaming_lookforward=5, streaming_lookback=50).items():
        assert getattr(tts, name) == val
    assert tts.streaming_stride_samples == 25 * tts.hop_length
    for attr in ("tokenizer", "backbone", "codec", "watermarker", "infer", "infer_stream", "encode_reference"):
        assert hasattr(tts, attr)
    import neuttsair
    assert issubclass(neuttsair.NeuTTSAir, type(tts))


def test_infer_smoke_and_equivalence(tts):
    bcfg, bw, ccfg, cw, tok, eos = tts._oracle
    ref_codes = torch.tensor([3, 77, 200, 5, 18, 9], dtype=torch.int32)
    audio = tts.infer("Testing.", ref_codes, "So I'm live.")
    # the reference's smoke assertions (ref:tests/test_neutts.py:55-58)
    assert isinstance(audio, np.ndarray) and len(audio) > 0 and not np.any(np.isnan(audio))
    assert audio.dtype in [np.float32, np.float64]
    # id-level equivalence with the oracle pipeline on the same prompt
    prompt = tts._apply_chat_template(ref_codes, "So I'm live.", "Testing.")
    assert prompt[-len(ref_codes):] == [tok.speech_base + int(c) for c in ref_codes]
    wd = br.cast_weights(bw, torch.bfloat16)
    ref = br.generate(bcfg, wd, prompt, 120, eos, min_new_tokens=5, keep_logits=True)
    got_ids = tts.generate_codes([prompt])[0]          # greedy: the ids `infer` just turned into audio
    br.assert_free_run_matches(got_ids, ref)           # identical, or identical up to an exact bf16 tie of the oracle's logits
    codes = [i - tok.speech_base for i in got_ids if i >= tok.speech_base]
    assert len(codes) > 0
    want = cr.decode_code(ccfg, cw, torch.tensor(codes)[None, None, :])[0, 0].numpy()
    assert audio.shape == want.shape == (len(codes) * tts.hop_length,)
    err = rms(audio - want)
    print(f"infer(): ids {'==' if got_ids == ref.ids else '~ (bf16 tie)'} oracle ({len(got_ids)} tokens); "
          f"waveform RMS error {err:.3e} (signal RMS {rms(want):.3e})")
    assert err <= 1e-3


def test_decode_without_speech_tokens_raises(tts):
    with pytest.raises(ValueError, match="No valid speech tokens found in the output."):
        tts._decode("no codes here")
    with pytest.raises(ValueError, match="No valid speech tokens found in the output."):
        tts._decode_ids([1, 2, 3])


def test_invalid_codec_repo_and_devices(emu_lib):
    from neutts import NeuTTS
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        NeuTTS(backbone_repo={"config": {}, "state_dict": {}, "inv_freq": None}, backbone_device="cpu", lib_path=emu_lib)


def test_infer_stream_matches_reference_windowing(tts):
    """Chunks are ndarrays, the main-loop chunks are exactly 25 frames, and the concatenation equals what the
    reference's streaming algorithm (ref:neutts/neutts.py:401-465, restated here with the oracle codec) yields."""
    bcfg, bw, ccfg, cw, tok, eos = tts._oracle
    ref_codes = [3, 77, 200, 5, 18, 9, 100, 41]
    tts.min_new_tokens = 70
    tts.max_context = 200
    try:
        chunks = list(tts.infer_stream("Streaming test.", ref_codes, "So I'm live."))
    finally:
        tts.min_new_tokens = 5
        tts.max_context = 120
    assert len(chunks) >= 2 and all(isinstance(c, np.ndarray) for c in chunks)
    hop = tts.hop_length
    assert all(len(c) == 25 * hop for c in chunks[:-1])
    # reference algorithm on the same generated tokens, oracle codec
    prompt = tts._apply_chat_template(ref_codes, "So I'm live.", "Streaming test.")
    wd = br.cast_weights(bw, torch.bfloat16)
    ref = br.generate(bcfg, wd, prompt, 200, eos, min_new_tokens=70, keep_logits=True)
    tts.min_new_tokens, tts.max_context = 70, 200
    try:
        ids = tts.generate_codes([prompt])[0]          # greedy: the ids the stream above was cut from
    finally:
        tts.min_new_tokens, tts.max_context = 5, 120
    br.assert_free_run_matches(ids, ref)
    new_codes = [i - tok.speech_base for i in ids if i >= tok.speech_base]

    def dec(cs):
        return cr.decode_code(ccfg, cw, torch.tensor(cs)[None, None, :])[0, 0].numpy()

    cache, audio, out = list(ref_codes), [], []
    n_tok, n_samp = len(ref_codes), 0
    for c in new_codes:
        cache.append(c)
        if len(cache) - n_tok >= 30:
            t0 = max(n_tok - 51, 0)
            s0 = (n_tok - t0) * hop
            audio.append(dec(cache[t0:n_tok + 31])[s0:s0 + 27 * hop])
            mixed = cr.linear_overlap_add(audio, 25 * hop)
            out.append(mixed[n_samp:len(audio) * 25 * hop])
            n_samp = len(audio) * 25 * hop
            n_tok += 25
    rem = len(cache) - n_tok
    if rem > 0:
        t0 = max(len(cache) - (51 + rem), 0)
        audio.append(dec(cache[t0:])[(len(cache) - t0 - rem - 1) * hop:])
        out.append(cr.linear_overlap_add(audio, 25 * hop)[n_samp:])
    want = np.concatenate(out)
    got = np.concatenate(chunks)
    assert got.shape == want.shape
    # chunk for chunk: every yielded chunk against the reference algorithm's chunk (not only the concatenation)
    assert [len(c) for c in chunks] == [len(o) for o in out]
    errs = [rms(c - o) for c, o in zip(chunks, out)]
    print(f"infer_stream(): {len(chunks)} chunks, per-chunk RMS error max {max(errs):.3e} (signal RMS {rms(want):.3e})")
    assert max(errs) <= 1e-3 and rms(got - want) <= 1e-3


@pytest.mark.parametrize("n_frames,last_len", [(1, 7), (2, 1), (5, 13), (9, 27), (4, 40)])
def test_stream_blender_is_bit_identical_to_full_reblend(n_frames, last_len):
    """The incremental cross-fade equals the reference's per-chunk re-blend of the whole audio cache
    (ref:neutts/neutts.py:441-448 main loop, :461-465 final chunk), bit for bit."""
    from neutts.neutts import _StreamBlender, _linear_overlap_add
    hop, chunk = 16, 25
    stride = chunk * hop
    rng = np.random.default_rng(n_frames * 100 + last_len)
    frames = [rng.standard_normal(27 * hop).astype(np.float32) for _ in range(n_frames - 1)]
    frames.append(rng.standard_normal(last_len * hop).astype(np.float32))
    b = _StreamBlender(stride)
    cache, n_done = [], 0
    for i, f in enumerate(frames):
        last = i == len(frames) - 1
        cache.append(f)
        full = _linear_overlap_add(cache, stride=stride)          # what the reference recomputes every chunk
        want = full[n_done:] if last else full[n_done:len(cache) * stride]
        n_done = len(cache) * stride
        got = b.push(f, last=last)
        assert got.dtype == want.dtype and np.array_equal(got, want)
    # and the oracle's restatement of _linear_overlap_add agrees with the product's
    assert np.array_equal(cr.linear_overlap_add(frames, stride), _linear_overlap_add(frames, stride=stride))


def test_infer_batch_shares_prompt_beginnings_and_matches_single_inference(tts):
    """Three utterances of one speaker in one call: the engine shares the KV pages of the common prompt beginning
    (chat header + reference-text phones, ref:neutts/neutts.py:307,315-325) and every waveform equals what `infer`
    gives for that utterance alone (greedy: same ids, hence the same codes and samples)."""
    ref_codes = torch.tensor([3, 77, 200, 5, 18, 9], dtype=torch.int32)
    ref_text = "So I'm live, and every prompt starts like this."
    texts = ["First.", "Second one."]
    tts.max_context = 150                 # the prompts are ~105 tokens here
    try:
        before = tts.backbone.kv_stats()
        batch = tts.infer_batch(texts, ref_codes, ref_text)
        after = tts.backbone.kv_stats()
        assert after["free_pages"] == after["total_pages"]
        assert after["prompt_tokens_shared"] - before["prompt_tokens_shared"] >= 64, "a 2-slot engine still shares with the running donor"
        for text, wav in zip(texts, batch):
            single = tts.infer(text, ref_codes, ref_text)
            assert wav.shape == single.shape and np.array_equal(wav, single)
    finally:
        tts.max_context = 120


def test_infer_batch_over_an_engine_gang_matches_single_inference(tts):
    """NeuTTS(engines=n): infer_batch deals its utterances out over n backbone engines on one copy of the weights (EngineGang) -- more
    utterances than one engine has slots -- and every waveform equals what `infer` gives for that utterance alone."""
    from neutts import _hip
    ref_codes = torch.tensor([3, 77, 200, 5, 18, 9], dtype=torch.int32)
    ref_text = "So I'm live, and every prompt starts like this."
    texts = ["First.", "Second one.", "And a third."]
    tts.max_context = 150
    tts.gang = _hip.EngineGang(tts.backbone, 2)          # what NeuTTS(engines=2) sets up at construction
    try:
        batch = tts.infer_batch(texts, ref_codes, ref_text)
        for e in tts.gang.engines:
            st = e.kv_stats()
            assert st["free_pages"] == st["total_pages"]
        for text, wav in zip(texts, batch):
            single = tts.infer(text, ref_codes, ref_text)        # (one utterance: the first engine, on its lane stream)
            assert wav.shape == single.shape and np.array_equal(wav, single)
    finally:
        tts.gang.close()
        tts.gang = None
        tts.max_context = 120


def test_infer_stream_batch_equals_single_streams(tts):
    """Two utterances streamed together: each one's chunks are exactly those of its own `infer_stream` -- the batch runs the DEVICE-side
    stream path (ntts_streams_*: append_codes / gather / codec / cross-fade kernels), the single streams the host loop: bit for bit."""
    ref_codes = [3, 77, 200, 5, 18, 9, 100, 41]
    texts = ["Streaming test.", "Another one, a little longer."]
    tts.min_new_tokens, tts.max_context = 34, 150      # one full window (30 tokens) + a final partial one
    try:
        tts.stream_on_device = False                                    # the single streams through the HOST loop (_infer_stream_hip) ...
        singles = [list(tts.infer_stream(t, ref_codes, "So I'm live.")) for t in texts]
        tts.stream_on_device = True                                     # ... and through the device path as a set of one stream (GPU only: time)
        if "emu" not in str(tts._lib_path):
            assert all(len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
                       for a, b in zip(singles, [list(tts.infer_stream(t, ref_codes, "So I'm live.")) for t in texts]))
        got = [[], []]
        budget = tts.backbone.cfg["max_prefill_tokens"]
        tts.backbone.cfg["max_prefill_tokens"] = 80     # smaller than the two prompts together: prefilled in two calls
        assert tts._stream_on_device([ref_codes, ref_codes])          # token caches, windows and the cross-fade on the device (csrc/stream.cpp)
        for i, chunk in tts.infer_stream_batch(texts, ref_codes, "So I'm live."):
            assert isinstance(chunk, np.ndarray)
            got[i].append(chunk)
        host = [[], []]
        tts.stream_on_device = False                                    # the same batch through the host loop (numpy windows + _StreamBlender)
        if "emu" not in str(tts._lib_path):                             # (GPU only: time; the single streams above already ran the host loop)
            for i, chunk in tts.infer_stream_batch(texts, ref_codes, "So I'm live."):
                host[i].append(chunk)
        else:
            host = got
    finally:
        tts.stream_on_device = True
        tts.backbone.cfg["max_prefill_tokens"] = budget
        tts.min_new_tokens, tts.max_context = 5, 120
    for i in range(2):
        assert len(host[i]) == len(got[i]) and all(np.array_equal(a, b) for a, b in zip(host[i], got[i]))
    st = tts.backbone.kv_stats()
    assert st["free_pages"] == st["total_pages"]
    for i in range(2):
        assert len(got[i]) == len(singles[i]) >= 2
        for a, b in zip(got[i], singles[i]):
            assert a.shape == b.shape and np.array_equal(a, b)
    with pytest.raises(ValueError, match="decode slots"):
        next(tts.infer_stream_batch(["a"] * (tts.backbone.max_batch + 1), ref_codes, "x"))


def test_infer_stream_batch_over_a_gang_with_staggered_admission(tts):
    """NeuTTS(engines=2): infer_stream_batch deals its utterances out over the gang in groups of `stream_admit`, one device-side stream set
    per group, the second group of an engine admitted while its first one already streams.  Three utterances, groups
    of one: every utterance's chunks are those of its own `infer_stream`, bit for bit -- a window is defined by token counts, not by
    where the decode bursts end."""
    from neutts import _hip
    ref_codes = [3, 77, 200, 5, 18, 9, 100, 41]
    texts = ["Streaming test.", "Another one, a little longer.", "Third."]       # three groups over two engines: engine 0 admits its second group while its first one streams
    tts.min_new_tokens, tts.max_context = 34, 150
    tts.gang = _hip.EngineGang(tts.backbone, 2)          # what NeuTTS(engines=2) sets up at construction
    tts.stream_admit = 1
    tts.stream_on_gang = True                            # (opt-in: off by default)
    try:
        tts.stream_on_device = False
        singles = [list(tts.infer_stream(t, ref_codes, "So I'm live.")) for t in texts]
        tts.stream_on_device = True
        got = [[] for _ in texts]
        order = []
        for i, chunk in tts.infer_stream_batch(texts, ref_codes, "So I'm live."):
            got[i].append(chunk)
            order.append(i)
        assert len(tts._gang_codecs) == 2
        for e in tts.gang.engines:
            st = e.kv_stats()
            assert st["free_pages"] == st["total_pages"] and e.free_slots() == e.max_batch
        for i in range(len(texts)):
            assert len(got[i]) == len(singles[i]) >= 2
            for a, b in zip(got[i], singles[i]):
                assert a.shape == b.shape and np.array_equal(a, b)
        assert order.index(2) > order.index(0)                                          # the second group of engine 0 came in behind its first one
        with pytest.raises(ValueError, match="decode slots"):
            next(tts.infer_stream_batch(["a"] * (tts.gang.max_batch + 1), ref_codes, "x"))
    finally:
        tts.stream_on_device = True
        del tts.stream_admit, tts.stream_on_gang
        for k, c in enumerate(tts._gang_codecs or []):
            c.set_stream(None)
            if k:
                c.close()
        tts._gang_codecs = None
        tts.gang.close()
        tts.gang = None
        tts.min_new_tokens, tts.max_context = 5, 120


def _device_i32(tts, shape):
    """A zeroed int32 buffer in the engines' 'device' memory (host memory on the emulator, HBM on the GPU): (pointer, reader)."""
    if "emu" in str(tts._lib_path):
        a = np.zeros(shape, dtype=np.int32)
        return a.ctypes.data, (lambda: a.copy()), a
    t = torch.zeros(shape, dtype=torch.int32, device="cuda")
    return t.data_ptr(), (lambda: t.cpu().numpy()), t


def test_device_side_code_handoff(tts):
    """ids -> codes -> waveform without the host round trip (include/neutts_hip.h ntts_backbone_export_codes +
    ntts_codec_decode_dev): the device-side selection equals `_ids_to_codes` (= the reference's tokenizer.decode + regex,
    ref:neutts/neutts.py:349,:276), specials and text ids dropped, and the codec pass fed from the device buffer -- ordered
    behind the backbone's stream -- returns exactly the waveforms of the host-fed pass."""
    bcfg, bw, ccfg, cw, tok, eos = tts._oracle
    eng, cod = tts.backbone, tts.codec.engine
    n_codes = int(np.prod(ccfg.levels))
    ref_codes = [3, 77, 200, 5, 18, 9]
    prompts = [tts._apply_chat_template(ref_codes, "So I'm live.", t) for t in ("Testing.", "A second, longer one.")]
    slots = [eng.acquire_slot() for _ in prompts]
    try:
        samp = [_hip_sampling(len(p) + 40, 12, eos) for p in prompts]
        eng.prefill(prompts, slots, samp)
        # a text byte and a special token in the middle of the stream must be dropped by the selection
        eng.decode(6)
        eng.debug_force(slots[0], 65)
        eng.decode(1)
        eng.debug_force(slots[0], tok.convert_tokens_to_ids("<|TEXT_PROMPT_END|>"))
        eng.decode(40)
        ids = [eng.read(s)[0] for s in slots]
        want = [tts._ids_to_codes(i) for i in ids]
        assert 65 in ids[0] and len(want[0]) <= len(ids[0]) - 2 and all(len(w) > 0 for w in want)
        stride = 64
        cptr, cread, _c = _device_i32(tts, (len(slots), stride))
        lptr, lread, _l = _device_i32(tts, (len(slots),))
        eng.export_codes(slots, tok.speech_base, n_codes, cptr, stride, lptr)
        eng.sync()
        lens, codes = lread(), cread()
        assert lens.tolist() == [len(w) for w in want]
        for u, w in enumerate(want):
            assert codes[u, :len(w)].tolist() == w
        wav = cod.decode_device(cptr, stride, lens, producer_stream=eng.stream())
        cod.sync()
        host = cod.decode(want)
        for u, w in enumerate(want):
            assert np.array_equal(wav[u, :len(w) * cod.hop_length], host[u])
        # the synthetic-benchmark mapping (id mod n_codes) keeps every id
        eng.export_codes(slots, 0, n_codes, cptr, stride, lptr, modulo=True)
        eng.sync()
        assert lread().tolist() == [min(len(i), stride) for i in ids]
        assert cread()[1, :len(ids[1])].tolist() == [i % n_codes for i in ids[1]][:stride]
    finally:
        eng.sync()
        for s in slots:
            eng.release(s)


def _hip_sampling(max_length, min_new, eos):
    from neutts import _hip
    return _hip.Sampling(max_length=max_length, min_new_tokens=min_new, eos_token_id=eos, do_sample=False)



def test_encode_reference_on_the_encoder_engine(tts, tmp_path):
    """ref:neutts/neutts.py:266-271 + ref:tests/test_neutts.py (reference codes are a 1-D integer tensor): a WAV file at
    another sample rate -> 16 kHz mono -> the encoder engine -> codes equal to the oracle's on the same samples; the
    `codec.encode_code` facade takes the tensor / array / path forms neucodec's does."""
    from scipy.io import wavfile
    from oracle import encoder_ref as er
    from neutts.neutts import load_audio_16k
    ecfg, ew = tts._oracle_encoder
    wav24 = syn.synthetic_speech(9000, 6, sample_rate=24000)
    path = tmp_path / "ref.wav"
    wavfile.write(str(path), 24000, (wav24 * 32767).astype(np.int16))
    wav16 = load_audio_16k(path)
    assert wav16.dtype == np.float32 and abs(wav16.size - 6000) <= 1
    codes = tts.encode_reference(path)
    assert isinstance(codes, torch.Tensor) and codes.dim() == 1 and not codes.is_floating_point()
    want, parts = er.encode(ecfg, ew, wav16, return_parts=True)
    # integer parity of a float pipeline: identical except where a latent sits on a rounding boundary (common.py)
    check_encoder_codes(ecfg, codes.numpy().astype(np.int32), tts.codec.enc_engine.read_stage("latents"), want, parts["latents"], 2e-4,
                        " class")
    via_tensor = tts.codec.encode_code(audio_or_path=torch.from_numpy(wav16)[None, None, :])
    assert via_tensor.shape == (1, 1, want.size) and torch.equal(via_tensor[0, 0], codes)
    assert torch.equal(tts.codec.encode_code(audio_or_path=str(path))[0, 0], codes)
    with pytest.raises(ValueError, match="shape"):
        tts.codec.encode_code(audio_or_path=np.zeros((1, 2, 100), np.float32))

