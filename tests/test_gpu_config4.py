"""BASELINE.json configs[4] as ONE test on a real MI355X: an fp8 backbone at the assumed NeuTTS-Nano widths
(e4m3 weights with per-channel scales + e4m3 GEMM inputs on the fp8 matrix cores) driving 64 CONCURRENT `infer_stream` utterances
(`NeuTTS.infer_stream_batch`: device-side token caches, windows, codec passes and cross-fade -- csrc/stream.cpp), every chunk of every
stream against the reference's windowing algorithm (ref:neutts/neutts.py:401-465, tests/common.py reference_stream_chunks) run on the
FP8 ORACLE's ids with the oracle codec.  Plus: what the fp8 model-level bar can and cannot see of a wrong `input_scale`."""
import numpy as np
import pytest
import torch

from oracle import backbone_ref as br
from oracle import codec_ref as cr
import synthetic as syn
from neutts import _hip
from common import engine_cfg, reference_stream_chunks, rms
import test_emu_variants as cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _hip.load_library(hip_lib)
    return hip_lib


def _fp8_tts(lib, n_streams, layers=4):
    from neutts import NeuTTS
    ccfg = cr.CodecConfig.tiny()
    n_codes = int(np.prod(ccfg.levels))
    tok = syn.ByteTokenizer(n_codes)
    # Nano WIDTHS (hidden 768, 12:4 heads, FFN 2048; the benchmark's 19 layers cut to 4 and the vocabulary to the byte tokenizer's so that the CPU
    # oracle's decode steps take seconds); greedy decoding walks a permutation of the SPEECH ids: the streams emit codec codes, a new
    # one every step, with top-1 / top-2 margins far wider than the fp8 noise -- the engine's free run IS the fp8 oracle's, id for id
    bcfg = br.BackboneConfig(vocab_size=tok.vocab_size, hidden_size=768, intermediate_size=2048, num_layers=layers, num_heads=12, num_kv_heads=4)
    bw = br.make_weights(bcfg, 41, walk_range=(tok.speech_base, tok.speech_base + n_codes), walk_gain=4.0)
    scales = br.default_fp8_input_scales(bcfg)
    cw = cr.make_weights(ccfg, 2)
    eos = tok.convert_tokens_to_ids("<|SPEECH_GENERATION_END|>")
    t = NeuTTS(
        backbone_repo=dict(config=engine_cfg(bcfg, max_context=256, max_prefill_tokens=4096, weight_dtype="fp8"),
                           state_dict={k: v.numpy() for k, v in bw.items()}, inv_freq=br.rope_inv_freq(bcfg).numpy(), input_scales=scales,
                           tokenizer=tok, speech_base=tok.speech_base, eos_token_id=eos),
        backbone_device="cuda",
        codec_repo=dict(config=dict(hidden_size=ccfg.hidden_size, intermediate_size=ccfg.intermediate_size, num_layers=ccfg.num_layers,
                                    num_heads=ccfg.num_heads, quantization_dim=ccfg.quantization_dim, levels=list(ccfg.levels),
                                    hop_length=ccfg.hop_length, max_frames=128, max_rows=n_streams * 96),
                        state_dict={k: v.numpy() for k, v in cw.items()}),
        codec_device="cuda", lib_path=lib, do_sample=False, max_batch=n_streams)
    t.phonemizer = syn.LowercasePhonemizer()
    t.watermarker = None
    return t, (bcfg, bw, scales, ccfg, cw, tok, eos)


def test_config4_fp8_backbone_64_concurrent_streams_chunk_for_chunk(lib):
    """64 streams = 4 distinct utterances (texts of different lengths: ragged prompts, ragged window phases) x 16 copies.  Every stream's
    chunks, one by one, against the reference algorithm on the fp8 ORACLE's ids (oracle/backbone_ref.py fp8_quantize_weights + generate: the
    written specification of this engine's quantisation scheme) decoded by the oracle codec in fp32: main-loop chunks are 25 frames, the
    tail is what is left, per-chunk waveform rms within BASELINE's 1e-3; copies of one utterance yield bit-identical chunks whichever slot
    they run in; all slots and KV pages come back."""
    n_streams, n_new = 64, 62                      # >= 62 tokens: two full windows (at 30 and 55 undecoded tokens) + a tail
    tts, (bcfg, bw, scales, ccfg, cw, tok, eos) = _fp8_tts(lib, n_streams)
    try:
        ref_codes = [3, 77, 200, 5, 18, 9, 100, 41]
        base = ["Streaming test.", "Another one, a little longer.", "Third.", "A fourth utterance of the batch, longer still than the others."]
        nd = len(base)
        texts = [base[i % nd] for i in range(n_streams)]
        tts.min_new_tokens, tts.max_context = n_new, 0          # (max_context set per prompt below)
        prompts = [tts._apply_chat_template(ref_codes, "So I'm live.", t) for t in base]
        tts.max_context = max(len(p) for p in prompts) + n_new   # every stream generates exactly n_new tokens (max_length stops the longest, EOS masked until then)
        tts.min_new_tokens = n_new
        assert tts._stream_on_device([ref_codes] * n_streams)
        got = [[] for _ in range(n_streams)]
        for i, chunk in tts.infer_stream_batch(texts, ref_codes, "So I'm live."):
            assert isinstance(chunk, np.ndarray) and chunk.dtype == np.float32
            got[i].append(chunk)
        st = tts.backbone.kv_stats()
        assert st["free_pages"] == st["total_pages"] and tts.backbone.free_slots() == n_streams
        # ---- the fp8 oracle's ids for the distinct prompts, and the engine's own (greedy: what the streams were cut from)
        wq = br.fp8_quantize_weights(br.cast_weights(bw, torch.bfloat16), scales)
        hop = tts.hop_length
        dec = lambda cs: cr.decode_code(ccfg, cw, torch.tensor(cs)[None, None, :])[0, 0].numpy()
        worst = 0.0
        for k, p in enumerate(prompts):
            want_ids = br.generate(bcfg, wq, p, tts.max_context, eos, min_new_tokens=n_new).ids
            n_k = tts.max_context - len(p)
            assert len(want_ids) == n_k and len(set(want_ids)) >= 0.7 * n_k, "the walk keeps emitting other speech ids (a permutation of the 256 codes: it may revisit)"
            new_codes = [i - tok.speech_base for i in want_ids if i >= tok.speech_base]
            assert len(new_codes) == n_k
            want = reference_stream_chunks(ref_codes, new_codes, dec, hop)
            assert len(want) >= 3
            for j in range(k, n_streams, nd):
                assert [len(c) for c in got[j]] == [len(c) for c in want], (j, [len(c) for c in got[j]], [len(c) for c in want])
                if j != k:
                    assert all(np.array_equal(a, b) for a, b in zip(got[j], got[k])), f"stream {j} differs from its copy {k}"
            errs = [rms(a - b) for a, b in zip(got[k], want)]
            worst = max(worst, max(errs))
            assert all(len(c) == 25 * hop for c in got[k][:-1])
            assert max(errs) <= 1e-3, (k, errs)
        print(f"configs[4]: {n_streams} concurrent fp8 streams, {len(got[0])}-{len(got[3])} chunks each, worst per-chunk rms error vs the reference windowing on the fp8 oracle's ids {worst:.2e}")
    finally:
        tts.close()


def _first_token_distance(lib, cfg, w, prompts, eng_scales, ref_rows16, ref_rows8):
    eng = _hip.BackboneEngine(engine_cfg(cfg, max_batch=len(prompts), max_context=128, max_prefill_tokens=512, weight_dtype="fp8"), 0, lib)
    eng.load_state_dict({k: v.numpy() for k, v in w.items()}, inv_freq=br.rope_inv_freq(cfg).numpy(), input_scales=eng_scales)
    eos = cfg.vocab_size - 1
    eng.set_debug(True)
    eng.prefill(prompts, list(range(len(prompts))), [_hip.Sampling(max_length=len(p) + 2, min_new_tokens=2, eos_token_id=eos, do_sample=False) for p in prompts])
    d = [cases.fp8_distance(eng.read_logits(s), ref_rows8[s], ref_rows16[s])[0] for s in range(len(prompts))]
    eng.close()
    return max(d)


def test_fp8_bar_against_a_wrong_input_scale_on_one_layer(lib):
    """is the model-level fp8 bar tight enough to see a wrong `input_scale` on ONE GEMM input of ONE layer?  Nano widths,
    19 layers, first-token logits of 4 prompts against the fp8 oracle with the RIGHT scales; the engine is loaded with layer 9's o_proj /
    down_proj input scale off by a factor.  e4m3 is a FLOATING-point format: a scale that is 2x or 4x too large only moves values inside its
    exponent range (same 3 mantissa bits: the distance stays at the correct engine's) until the small values reach the subnormals; a scale
    that is too SMALL saturates the large values at 448 -- that is the error a wrong scale makes, and the bar must catch it: 1/8 of the
    right scale on one of 76 GEMM inputs has to land ABOVE the bar, the correct engine below it.  The bar itself: 1.2x the largest distance
    the correct engine has shown (FP8_BAR_19_LAYERS)."""
    from test_gpu_variants import FP8_BAR_19_LAYERS
    cfg = br.BackboneConfig.neutts_nano_like(8192)          # (the 19 layers and the widths; a small vocabulary keeps the CPU oracle to seconds)
    w = br.make_weights(cfg, 23, peak_sigma=0.5)
    scales = br.default_fp8_input_scales(cfg)
    wb = br.cast_weights(w, torch.bfloat16)
    wq = br.fp8_quantize_weights(wb, scales)
    prompts = [br.synthetic_prompt(cfg, i, 70) for i in range(4)]
    eos = cfg.vocab_size - 1
    ref8 = [br.generate(cfg, wq, p, len(p) + 1, eos, min_new_tokens=1, keep_logits=True).logits[0].numpy() for p in prompts]
    ref16 = [br.generate(cfg, wb, p, len(p) + 1, eos, min_new_tokens=1, keep_logits=True).logits[0].numpy() for p in prompts]
    right = _first_token_distance(lib, cfg, w, prompts, scales, ref16, ref8)
    out = {}
    for name in ("self_attn.o_proj", "mlp.down_proj"):
        for f in (0.125, 4.0):
            bad = dict(scales)
            bad[f"model.layers.9.{name}.input_scale"] = scales[f"model.layers.9.{name}.input_scale"] * f
            out[(name, f)] = _first_token_distance(lib, cfg, w, prompts, bad, ref16, ref8)
    print(f"fp8, 19 layers: distance from the fp8 oracle with the right scales {right:.3f} (bar {FP8_BAR_19_LAYERS}); one input_scale of layer 9 wrong: "
          + ", ".join(f"{n} x{f:g}: {d:.3f}" for (n, f), d in out.items()))
    assert right <= FP8_BAR_19_LAYERS
    assert out[("self_attn.o_proj", 0.125)] > FP8_BAR_19_LAYERS and out[("mlp.down_proj", 0.125)] > FP8_BAR_19_LAYERS   # saturation: seen
