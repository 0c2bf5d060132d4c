"""Generate tests/golden/infer_air_dave.npz: BASELINE.json configs[0]'s workload -- one utterance, the reference voice
`ref:samples/dave.pt` (372 codec codes), greedy -- through the reference's own `infer()` path restated with the third-party
implementations it calls, at NeuTTS-Air's layer geometry (hidden 896, 24 layers, 14:2 heads, FFN 4864) and NeuCodec's decoder
geometry, on synthetic weights (no checkpoint is reachable offline).

    python -m oracle.gen_golden_infer          (build container only: reads /root/reference/samples/dave.pt, needs transformers)

What is restated, line by line:
  ref:neutts/neutts.py:303-332  _apply_chat_template  (prompt ids: chat header, phones of ref text + text, the ref codes as
                                                       <|speech_N|> tokens after <|SPEECH_GENERATION_START|>)
  ref:neutts/neutts.py:334-352  _infer_torch          (transformers generate as the reference calls it, sampling off, bf16)
  ref:neutts/neutts.py:273-295  _decode               (regex over the decoded string -> codes -> codec.decode_code)
The tokenizer is the byte-level stand-in synthetic.ByteTokenizer (every special / speech token one id) and the
phonemizer synthetic.LowercasePhonemizer (espeak is not installed here; the text front-end is off the hot path).
The fixture holds inputs and outputs only: dave's codes, the texts, the prompt ids, the generated ids with their top-4 logits
per step, and the fp32 waveform of the oracle codec on the generated codes.  Weights are rebuilt from their seeds.
"""
from __future__ import annotations

import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import backbone_ref as br  # noqa: E402
from oracle import codec_ref as cr  # noqa: E402
from oracle.gen_golden import hf_backbone  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED_BACKBONE, SEED_CODEC = 41, 2
REF_TEXT = "So I'm live on radio."
TEXT = "My name is Dave, and um, I'm from London."
N_NEW, MIN_NEW = 120, 50


def backbone_config(vocab):
    return br.BackboneConfig(vocab_size=vocab)          # NeuTTS-Air's layer geometry; the vocabulary is the stand-in tokenizer's


def make_backbone_weights(cfg, speech_base, n_codes):
    # greedy decoding walks a permutation of the speech tokens with wide margins (synthetic._make_walk; the recipe of
    # tests/test_emu_neutts_class.py build_tts at 24 layers): 120 different codec codes, ids comparable id for id
    return br.make_weights(cfg, SEED_BACKBONE, walk_gain=8.0, walk_scale=8.0, walk_range=(speech_base, speech_base + n_codes))


def apply_chat_template(tok, phon, ref_codes, ref_text, input_text):
    """ref:neutts/neutts.py:303-332, statement by statement."""
    def to_phones(text):
        return " ".join(phon.phonemize([text])[0].split())
    input_text = to_phones(ref_text) + " " + to_phones(input_text)
    speech_replace = tok.convert_tokens_to_ids("<|SPEECH_REPLACE|>")
    speech_gen_start = tok.convert_tokens_to_ids("<|SPEECH_GENERATION_START|>")
    text_replace = tok.convert_tokens_to_ids("<|TEXT_REPLACE|>")
    text_prompt_start = tok.convert_tokens_to_ids("<|TEXT_PROMPT_START|>")
    text_prompt_end = tok.convert_tokens_to_ids("<|TEXT_PROMPT_END|>")
    input_ids = tok.encode(input_text, add_special_tokens=False)
    ids = tok.encode("user: Convert the text to speech:<|TEXT_REPLACE|>\nassistant:<|SPEECH_REPLACE|>")
    i = ids.index(text_replace)
    ids = ids[:i] + [text_prompt_start] + input_ids + [text_prompt_end] + ids[i + 1:]
    i = ids.index(speech_replace)
    codes = tok.encode("".join(f"<|speech_{c}|>" for c in ref_codes), add_special_tokens=False)
    return ids[:i] + [speech_gen_start] + list(codes)


def main():
    import synthetic as syn
    FakeTokenizer, FakePhonemizer = syn.ByteTokenizer, syn.LowercasePhonemizer
    dave = torch.load("/root/reference/samples/dave.pt").to(torch.int64).tolist()
    ccfg = cr.CodecConfig.neucodec()
    tok = FakeTokenizer(int(np.prod(ccfg.levels)))
    cfg = backbone_config(tok.vocab_size)
    w = make_backbone_weights(cfg, tok.speech_base, int(np.prod(ccfg.levels)))
    eos = tok.convert_tokens_to_ids("<|SPEECH_GENERATION_END|>")
    prompt = apply_chat_template(tok, FakePhonemizer(), dave, REF_TEXT, TEXT)
    m = hf_backbone(cfg, w, torch.bfloat16)
    t = time.time()
    out = m.generate(torch.tensor([prompt]), max_length=len(prompt) + N_NEW, eos_token_id=eos, pad_token_id=eos, do_sample=False,
                     use_cache=True, min_new_tokens=MIN_NEW, output_scores=True, return_dict_in_generate=True)
    ids = out.sequences[0, len(prompt):].tolist()
    top = [torch.topk(s[0].float(), 4) for s in out.scores]
    print(f"[infer_air_dave] prompt {len(prompt)} ids, {len(ids)} generated in {time.time() - t:.1f}s, "
          f"{len(set(ids))} distinct, min top-2 margin {min(float(x.values[0] - x.values[1]) for x in top):.4g}")
    text = tok.decode(ids, add_special_tokens=False)                                  # ref :348-351
    codes = [int(x) for x in re.findall(r"<\|speech_(\d+)\|>", text)]                 # ref :276
    assert codes, "no speech tokens"
    cw = cr.make_weights(ccfg, SEED_CODEC)
    wav = cr.decode_code(ccfg, cw, torch.tensor(codes, dtype=torch.long)[None, None, :])[0, 0].numpy()   # ref :288-292
    np.savez_compressed(os.path.join(GOLD, "infer_air_dave.npz"), dave_codes=np.array(dave, dtype=np.int32), ref_text=REF_TEXT, text=TEXT,
                        prompt=np.array(prompt, dtype=np.int64), ids=np.array(ids, dtype=np.int64),
                        topv=np.stack([x.values.numpy() for x in top]).astype(np.float32),
                        topi=np.stack([x.indices.numpy() for x in top]).astype(np.int64), codes=np.array(codes, dtype=np.int32),
                        wav=wav.astype(np.float32), n_new=N_NEW, min_new=MIN_NEW, eos=eos, seed_backbone=SEED_BACKBONE, seed_codec=SEED_CODEC)
    print(f"[infer_air_dave] {len(codes)} codes -> {len(wav)} samples, RMS {float(np.sqrt(np.mean(wav.astype(np.float64) ** 2))):.4g}")


if __name__ == "__main__":
    main()
