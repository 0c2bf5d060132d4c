"""Architecture / precision switches of the backbone engine (ABI 2) on the SIMT emulator, each against the oracle:
untied lm_head, bias-free q/k/v projections (Llama-style), the tied-head consistency check, and the fp8 model
(e4m3 weights with per-output-channel scales, e4m3 GEMM inputs with static scales)."""
import numpy as np
import pytest
import torch

from oracle import backbone_ref as br
from neutts import _hip
from common import assert_free_run_matches, engine_cfg


@pytest.fixture(scope="module")
def lib(emu_lib):
    return emu_lib


def _engine(cfg, w, lib, max_batch=2, input_scales=None, **kw):
    eng = _hip.BackboneEngine(engine_cfg(cfg, max_batch=max_batch, max_context=128, max_prefill_tokens=256,
                                         tie_word_embeddings=cfg.tie_word_embeddings, attention_bias=cfg.attention_bias, **kw), 0, lib)
    eng.load_state_dict({k: v.numpy() for k, v in w.items()}, inv_freq=br.rope_inv_freq(cfg).numpy(), input_scales=input_scales)
    return eng


def _run(eng, cfg, prompts, n_new):
    eos = cfg.vocab_size - 1
    samp = [_hip.Sampling(max_length=len(p) + n_new, min_new_tokens=n_new, eos_token_id=eos, do_sample=False) for p in prompts]
    return eng.generate(prompts, samp, steps_per_poll=4)


@pytest.mark.parametrize("small_batch", ["8", "0"])
def test_untied_head_and_no_bias_match_oracle(lib, small_batch, monkeypatch):
    """tie_word_embeddings = 0 (a separate lm_head.weight) and attention_bias = 0, on both decode paths."""
    monkeypatch.setenv("NTTS_SMALL_BATCH", small_batch)
    cfg = br.BackboneConfig(vocab_size=640, hidden_size=448, intermediate_size=1216, num_layers=2, num_heads=7, num_kv_heads=1,
                            attention_bias=False, tie_word_embeddings=False)
    w = br.make_weights(cfg, 17, peak_sigma=0.5)
    assert "lm_head.weight" in w and not any(k.endswith(".bias") for k in w)
    wd = br.cast_weights(w, torch.bfloat16)
    prompts = [br.synthetic_prompt(cfg, i, n) for i, n in enumerate((33, 7))]
    eng = _engine(cfg, w, lib)
    got = _run(eng, cfg, prompts, 8)
    for g, p in zip(got, prompts):
        assert_free_run_matches(g, br.generate(cfg, wd, p, len(p) + 8, cfg.vocab_size - 1, min_new_tokens=8, keep_logits=True))
    # the head really is the separate matrix: with the embedding in its place the ids change
    w2 = dict(w)
    w2["lm_head.weight"] = w["model.embed_tokens.weight"]
    assert _run(_engine(cfg, w2, lib), cfg, prompts, 8) != got


def test_tied_head_must_equal_embedding(lib):
    """an untied lm_head.weight handed to a tied engine is an error in EITHER load order, never dropped."""
    cfg = br.BackboneConfig.tiny(vocab_size=256, num_layers=1)
    w = br.make_weights(cfg, 3)
    other = w["model.embed_tokens.weight"] + 0.25
    for order in (("model.embed_tokens.weight", "lm_head.weight"), ("lm_head.weight", "model.embed_tokens.weight")):
        eng = _hip.BackboneEngine(engine_cfg(cfg, max_batch=1, max_context=64, max_prefill_tokens=64), 0, lib)
        eng.load_tensor(order[0], (w["model.embed_tokens.weight"] if order[0].startswith("model") else other).numpy())
        with pytest.raises(_hip.NeuTTSHipError, match="differs from"):
            eng.load_tensor(order[1], (w["model.embed_tokens.weight"] if order[1].startswith("model") else other).numpy())
        eng.close()
    # the same values under both names are fine (some exporters keep both)
    eng = _hip.BackboneEngine(engine_cfg(cfg, max_batch=1, max_context=64, max_prefill_tokens=64), 0, lib)
    sd = {k: v.numpy() for k, v in w.items()}
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    eng.load_state_dict(sd, inv_freq=br.rope_inv_freq(cfg).numpy())
    eng.close()
    # and a missing tensor is named
    eng = _hip.BackboneEngine(engine_cfg(cfg, max_batch=1, max_context=64, max_prefill_tokens=64), 0, lib)
    sd.pop("model.layers.0.mlp.up_proj.weight")
    with pytest.raises(_hip.NeuTTSHipError, match="mlp.up_proj.weight"):
        eng.load_state_dict(sd, inv_freq=br.rope_inv_freq(cfg).numpy())


def fp8_cfg(vocab=512, layers=2):
    # every GEMM K extent a multiple of 128 (one 128-byte tile of e4m3): hidden 384, q width 384, FFN 1024
    return br.BackboneConfig(vocab_size=vocab, hidden_size=384, intermediate_size=1024, num_layers=layers, num_heads=6, num_kv_heads=2)


def test_fp8_model_matches_fp8_oracle(lib):
    """weight_dtype = fp8: prefill + decode, ragged prompts, vs the oracle's restatement of the same quantisation scheme:
    greedy ids (free run, tie-aware) and the first-token logits within a few bf16 ulps."""
    cfg = fp8_cfg()
    w = br.make_weights(cfg, 23, peak_sigma=0.5)
    scales = br.default_fp8_input_scales(cfg)
    wq = br.fp8_quantize_weights(br.cast_weights(w, torch.bfloat16), scales)
    prompts = [br.synthetic_prompt(cfg, i, n) for i, n in enumerate((40, 70, 5))]
    eng = _engine(cfg, w, lib, max_batch=3, input_scales=scales, weight_dtype="fp8")
    eng.set_debug(True)
    eos = cfg.vocab_size - 1
    samp = [_hip.Sampling(max_length=len(p) + 10, min_new_tokens=10, eos_token_id=eos, do_sample=False) for p in prompts]
    eng.prefill(prompts, [0, 1, 2], samp)
    want = [br.generate(cfg, wq, p, len(p) + 10, eos, min_new_tokens=10, keep_logits=True) for p in prompts]
    for s in range(3):
        row = eng.read_logits(s)
        ref = want[s].logits[0].numpy()
        fin = np.isfinite(ref)
        err = np.abs(row[fin] - ref[fin]) / np.array([br.bf16_ulp(v) for v in ref[fin]])
        assert err.max() <= 4.0 and err.mean() <= 0.6, (s, err.max(), err.mean())
    eng.set_debug(False)
    eng.decode(9)
    for s in range(3):
        ids, fin = eng.read(s)
        assert fin
        assert_free_run_matches(ids, want[s])
    # the quantised model is a different model from the bf16 one, but a close one
    bf = br.generate(cfg, br.cast_weights(w, torch.bfloat16), prompts[0], len(prompts[0]) + 10, eos, min_new_tokens=10, keep_logits=True)
    a, b = bf.logits[0].numpy()[:-1], want[0].logits[0].numpy()[:-1]
    assert np.corrcoef(a, b)[0, 1] > 0.97


def test_fp8_needs_its_input_scales(lib):
    cfg = fp8_cfg(layers=1)
    w = br.make_weights(cfg, 1)
    eng = _hip.BackboneEngine(engine_cfg(cfg, max_batch=1, max_context=64, max_prefill_tokens=64, weight_dtype="fp8"), 0, lib)
    with pytest.raises(_hip.NeuTTSHipError, match="input_scale"):
        eng.load_state_dict({k: v.numpy() for k, v in w.items()}, inv_freq=br.rope_inv_freq(cfg).numpy())
    bf = _hip.BackboneEngine(engine_cfg(cfg, max_batch=1, max_context=64, max_prefill_tokens=64), 0, lib)
    with pytest.raises(_hip.NeuTTSHipError, match="fp8 model"):
        bf.load_tensor("lm_head.input_scale", np.asarray([0.5], dtype=np.float32))
    with pytest.raises(_hip.NeuTTSHipError, match="multiples of 128"):
        _hip.BackboneEngine(engine_cfg(br.BackboneConfig.tiny(), max_batch=1, weight_dtype="fp8"), 0, lib)
