"""Shared bodies of tests/test_emu_checkpoint_dir.py (CPU SIMT emulator) and tests/test_gpu_checkpoint_dir.py (MI355X).

The reference's loading path (ref:neutts/neutts.py:163-166): `NeuTTS(backbone_repo=<HF checkpoint>)` reads the
tokenizer and the Qwen2 weights with `transformers`.  Here a tiny Qwen2 checkpoint + a real `tokenizers` tokenizer
(with the NeuTTS special / `<|speech_N|>` tokens) are written with `save_pretrained`, loaded through that path onto
the emulated engine, and the class's prompt construction + greedy ids are compared with transformers' own
`generate` on the very same checkpoint, called as the reference calls it (ref:neutts/neutts.py:338-347)."""
import numpy as np
import pytest
import torch

from oracle import backbone_ref as br
from oracle import codec_ref as cr
from oracle.gen_golden import hf_backbone

SPECIALS = ["<|TEXT_REPLACE|>", "<|SPEECH_REPLACE|>", "<|TEXT_PROMPT_START|>", "<|TEXT_PROMPT_END|>",
            "<|SPEECH_GENERATION_START|>", "<|SPEECH_GENERATION_END|>"]


def build_tokenizer(n_codes):
    """Byte-level BPE without merges (one token per byte) in Qwen2's tokenizer format -- AutoTokenizer resolves a
    `model_type: qwen2` checkpoint to Qwen2Tokenizer, which assumes the ByteLevel alphabet -- plus the NeuTTS tokens."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {c: i for i, c in enumerate(sorted(pre_tokenizers.ByteLevel.alphabet()))}
    tk = Tokenizer(models.BPE(vocab=vocab, merges=[]))
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tk.decoder = decoders.ByteLevel()
    fast = PreTrainedTokenizerFast(tokenizer_object=tk)
    fast.add_special_tokens({"additional_special_tokens": SPECIALS})
    fast.add_tokens([f"<|speech_{i}|>" for i in range(n_codes)], special_tokens=True)
    return fast




def make_ckpt(d, wide=False):
    """A Qwen2 checkpoint directory + tokenizer written by transformers / tokenizers `save_pretrained`.  wide: NeuTTS-Air's width
    (hidden 896, 14 / 2 heads, FFN 4864) with two layers -- the GPU twin; else the tiny geometry the emulator finishes in seconds."""
    ccfg = cr.CodecConfig.tiny()
    n_codes = int(np.prod(ccfg.levels))
    tok = build_tokenizer(n_codes)
    cfg = (br.BackboneConfig(vocab_size=len(tok), num_layers=2) if wide else br.BackboneConfig.tiny(vocab_size=len(tok), num_layers=1))
    base = tok.convert_tokens_to_ids("<|speech_0|>")
    w = br.make_weights(cfg, 77, walk_gain=4.0, walk_range=(base, base + n_codes))   # greedy decoding walks a permutation of the speech tokens
    m = hf_backbone(cfg, w, torch.float32)             # the reference loads fp32 weights (ref :164)
    m.save_pretrained(str(d))
    tok.save_pretrained(str(d))
    return str(d), cfg, w, tok, ccfg


def codec_spec(ccfg, cw):
    return dict(config=dict(hidden_size=ccfg.hidden_size, intermediate_size=ccfg.intermediate_size, num_layers=ccfg.num_layers,
                            num_heads=ccfg.num_heads, quantization_dim=ccfg.quantization_dim, levels=list(ccfg.levels),
                            hop_length=ccfg.hop_length, max_frames=256, max_rows=1024),
                state_dict={k: v.numpy() for k, v in cw.items()})


class Phon:   # espeak is not installed here; the text front-end is off the hot path
    def phonemize(self, texts):
        return [t.lower() for t in texts]


def hf_greedy(hf, prompt, max_length, eos, min_new):
    """transformers' own generate on the checkpoint, called as the reference calls it (ref:neutts/neutts.py:338-347) with
    sampling off; returns the new ids and the processed fp32 scores of every step."""
    out = hf.generate(torch.tensor([prompt]), max_length=max_length, eos_token_id=eos, pad_token_id=eos, do_sample=False,
                      use_cache=True, min_new_tokens=min_new, output_scores=True, return_dict_in_generate=True)
    return out.sequences[0, len(prompt):].tolist(), [s[0].float() for s in out.scores]


def assert_ids_match_hf(got, want, scores, max_ulps=2.0):
    """Identical ids (HF may or may not append the EOS it stopped on), or identical up to the first step where transformers' OWN
    top-2 logits are within `max_ulps` bf16 ulps -- an exact or near tie, which fp32 summation order decides (the two-layer model
    at NeuTTS-Air's width has them now and then) -- with our token one of those two.  Returns the number of ids compared equal."""
    n = min(len(got), len(want))
    k = next((i for i in range(n) if got[i] != want[i]), None)
    if k is None:
        assert abs(len(got) - len(want)) <= 1, (len(got), len(want))
        return n
    top = torch.topk(scores[k], 2)
    gap = float(top.values[0] - top.values[1])
    assert gap <= max_ulps * br.bf16_ulp(float(top.values[0])) and got[k] in top.indices.tolist(), \
        f"step {k}: got {got[k]}, transformers {want[k]}, its top-2 gap {gap}"
    print(f"checkpoint-dir case: ids equal up to a {gap:g} tie of transformers' own logits at step {k}")
    return k


def run_checkpoint_dir_case(ckpt, lib):
    from neutts import NeuTTS
    d, cfg, w, tok, ccfg = ckpt
    cw = cr.make_weights(ccfg, 4)
    tts = NeuTTS(backbone_repo=d, backbone_device="cuda", codec_repo=codec_spec(ccfg, cw), codec_device="cuda", lib_path=lib, do_sample=False)
    tts.phonemizer = Phon()
    tts.max_context, tts.min_new_tokens = 160, 6          # lowered to prompt + 24 once the prompt is known
    assert tts._backbone_loader == "safetensors"      # tensors streamed from the shards, no nn.Module copy on the host
    assert tts._speech_base == tok.convert_tokens_to_ids("<|speech_0|>")
    assert tts._eos_id == tok.convert_tokens_to_ids("<|SPEECH_GENERATION_END|>")

    ref_codes = torch.tensor([3, 77, 200, 5, 18, 9], dtype=torch.int32)
    prompt = tts._apply_chat_template(ref_codes, "So I'm live.", "Testing.")
    # prompt layout of ref:neutts/neutts.py:303-332, checked through the tokenizer's own decode
    text = tts.tokenizer.decode(prompt, skip_special_tokens=False)
    assert text == ("user: Convert the text to speech:<|TEXT_PROMPT_START|>so i'm live. testing.<|TEXT_PROMPT_END|>"
                    "\nassistant:<|SPEECH_GENERATION_START|>" + "".join(f"<|speech_{int(c)}|>" for c in ref_codes))

    tts.max_context = len(prompt) + 24            # keeps the emulated run short
    # transformers on the same checkpoint, bf16 compute, the reference's generate() call with sampling off
    from transformers import AutoModelForCausalLM
    hf = AutoModelForCausalLM.from_pretrained(d, attn_implementation="eager").to(torch.bfloat16).eval()
    hf.model.rotary_emb.inv_freq = br.rope_inv_freq(cfg)            # keep the fp32 buffer (oracle/backbone_ref.py docstring)
    hf.model.rotary_emb.original_inv_freq = br.rope_inv_freq(cfg)
    want, scores = hf_greedy(hf, prompt, tts.max_context, tts._eos_id, 6)
    got = tts.generate_codes([prompt])[0]
    n = assert_ids_match_hf(got, want, scores)
    assert n >= 23 and len(set(got[:n])) >= n - 1          # the whole run (HF may or may not append the EOS it stopped on), a new id every step
    # id -> code hand-off = tokenizer.decode + regex of the reference (ref :349, :276)
    import re
    s = tok.decode(got, skip_special_tokens=False)
    assert tts._ids_to_codes(got) == [int(x) for x in re.findall(r"<\|speech_(\d+)\|>", s)]
    audio = tts.infer("Testing.", ref_codes, "So I'm live.")
    assert isinstance(audio, np.ndarray) and audio.dtype == np.float32 and len(audio) == tts.hop_length * len(tts._ids_to_codes(got))


def run_llama_dispatch_case(tmp_path, lib, wide=False):
    """The AutoModelForCausalLM dispatch of ref:neutts/neutts.py:164 beyond Qwen2: a Llama checkpoint (no q/k/v bias, UNTIED
    lm_head -- `model_type: llama`, `tie_word_embeddings: false`, `attention_bias: false` in its config.json) saved by
    transformers, loaded through `NeuTTS(backbone_repo=dir)`, gives the ids of transformers' own LlamaForCausalLM.generate
    on that checkpoint (bf16, eager attention, greedy)."""
    from transformers import AutoModelForCausalLM, LlamaConfig, LlamaForCausalLM
    from neutts import NeuTTS
    from neutts.neutts import _engine_config_from_hf
    ccfg = cr.CodecConfig.tiny()
    n_codes = int(np.prod(ccfg.levels))
    tok = build_tokenizer(n_codes)
    cfg = (br.BackboneConfig(vocab_size=len(tok), num_layers=2, attention_bias=False, tie_word_embeddings=False) if wide else
           br.BackboneConfig(vocab_size=len(tok), hidden_size=448, intermediate_size=1216, num_layers=1, num_heads=7, num_kv_heads=1,
                             attention_bias=False, tie_word_embeddings=False))
    base = tok.convert_tokens_to_ids("<|speech_0|>")
    w = br.make_weights(cfg, 78, walk_gain=4.0, walk_range=(base, base + n_codes))
    hc = LlamaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                     num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
                     head_dim=64, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, max_position_embeddings=2048,
                     tie_word_embeddings=False, attention_bias=False, mlp_bias=False)
    m = LlamaForCausalLM(hc).eval()
    missing, unexpected = m.load_state_dict(w, strict=False)
    assert not unexpected and all("inv_freq" in k for k in missing), (missing, unexpected)
    d = str(tmp_path)
    m.save_pretrained(d)
    tok.save_pretrained(d)
    cw = cr.make_weights(ccfg, 4)
    tts = NeuTTS(backbone_repo=d, backbone_device="cuda", codec_repo=codec_spec(ccfg, cw), codec_device="cuda", lib_path=lib, do_sample=False)
    assert tts.backbone.cfg["tie_word_embeddings"] is False and tts.backbone.cfg["attention_bias"] is False
    prompt = list(b"hello") + [tok.convert_tokens_to_ids("<|SPEECH_GENERATION_START|>")] + [base + c for c in (3, 77, 200, 5, 18, 9)]
    tts.max_context, tts.min_new_tokens = len(prompt) + 20, 6
    hf = AutoModelForCausalLM.from_pretrained(d, attn_implementation="eager").to(torch.bfloat16).eval()
    hf.model.rotary_emb.inv_freq = br.rope_inv_freq(cfg)
    hf.model.rotary_emb.original_inv_freq = br.rope_inv_freq(cfg)
    want, scores = hf_greedy(hf, prompt, tts.max_context, tts._eos_id, 6)
    got = tts.generate_codes([prompt])[0]
    n = assert_ids_match_hf(got, want, scores)
    assert n >= 20 and len(set(got[:n])) >= n - 1          # the whole run, a new id every step (walk weights)


def run_qwen3_dispatch_case(tmp_path, lib, hidden=256, heads=2, kv_heads=1, ffn=512, layers=2):
    """Round 6: a Qwen3 checkpoint -- `model_type: qwen3`: per-head q/k RMSNorm before RoPE, head_dim 128 (q width != hidden),
    bias-free projections -- saved by transformers and loaded through `NeuTTS(backbone_repo=dir)` (config dispatch, safetensors streaming incl. the
    `self_attn.q_norm / k_norm` weights) gives the ids of transformers' own Qwen3ForCausalLM.generate on that checkpoint (bf16, eager attention,
    greedy).  The reference would reach this through AutoModelForCausalLM (ref:neutts/neutts.py:164) if its default repo were Qwen3-based."""
    from transformers import AutoModelForCausalLM, Qwen3Config, Qwen3ForCausalLM
    from neutts import NeuTTS
    ccfg = cr.CodecConfig.tiny()
    n_codes = int(np.prod(ccfg.levels))
    tok = build_tokenizer(n_codes)
    cfg = br.BackboneConfig(vocab_size=len(tok), hidden_size=hidden, intermediate_size=ffn, num_layers=layers, num_heads=heads, num_kv_heads=kv_heads,
                            head_dim=128, attention_bias=False, qk_norm=True)
    base = tok.convert_tokens_to_ids("<|speech_0|>")
    w = br.make_weights(cfg, 91, walk_gain=4.0, walk_range=(base, base + n_codes))
    hc = Qwen3Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                     num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
                     head_dim=128, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, max_position_embeddings=2048,
                     tie_word_embeddings=True, attention_bias=False)
    m = Qwen3ForCausalLM(hc).eval()
    sd = dict(w)
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("inv_freq" in k for k in missing), (missing, unexpected)
    d = str(tmp_path)
    m.save_pretrained(d)
    tok.save_pretrained(d)
    cw = cr.make_weights(ccfg, 4)
    tts = NeuTTS(backbone_repo=d, backbone_device="cuda", codec_repo=codec_spec(ccfg, cw), codec_device="cuda", lib_path=lib, do_sample=False)
    assert tts.backbone.cfg["qk_norm"] is True and tts.backbone.cfg["head_dim"] == 128 and tts.backbone.cfg["attention_bias"] is False
    prompt = list(b"hello") + [tok.convert_tokens_to_ids("<|SPEECH_GENERATION_START|>")] + [base + c for c in (3, 77, 200, 5, 18, 9)]
    tts.max_context, tts.min_new_tokens = len(prompt) + 20, 6
    hf = AutoModelForCausalLM.from_pretrained(d, attn_implementation="eager").to(torch.bfloat16).eval()
    hf.model.rotary_emb.inv_freq = br.rope_inv_freq(cfg)
    hf.model.rotary_emb.original_inv_freq = br.rope_inv_freq(cfg)
    want, scores = hf_greedy(hf, prompt, tts.max_context, tts._eos_id, 6)
    got = tts.generate_codes([prompt])[0]
    n = assert_ids_match_hf(got, want, scores)
    assert n >= 20 and len(set(got[:n])) >= n - 1
    tts.close()
