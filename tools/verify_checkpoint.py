#!/usr/bin/env python
"""Day-one validator for REAL checkpoints -- for a user who has network access and the weights, which this
repository's build container never had (ref:neutts/neutts.py:77,164,188 download `neuphonic/neutts-air`, `neuphonic/neutts-nano`,
`neuphonic/neucodec`).  Never imported by the product.

    python tools/verify_checkpoint.py <backbone_dir_or_repo> [--codec <neucodec state dict .pt/.safetensors | 'neuphonic/neucodec'>]
                                      [--steps 32] [--device cuda:0] [--lib path/to/libneutts_hip.so]

1. config.json: what `_engine_config_from_hf` makes of it, and the delta to the geometries this repository ASSUMED offline
   (synthetic.BackboneConfig.neutts_air() / .neutts_nano_like()).
2. Loads the backbone through the product's own loader (the code path of `NeuTTS(backbone_repo=...)`), builds a prompt with
   `_apply_chat_template`-style token ids (chat header, the id of "<|speech_0|>" checked against the tokenizer), and runs `--steps`
   TEACHER-FORCED greedy steps against `transformers` on the CPU (bf16, eager attention -- the engine's numeric contract): every id equal
   to transformers', or different only where transformers' own top-2 logits lie within 4 bf16 ulps (then ours must be one of its top 4);
   with the engine's debug tap, the logit error at transformers' top-4 ids in bf16 ulps (the bars of tests/test_gpu_backbone.py:
   mean <= 0.8, p99 <= 2.5, max <= 3.5).
3. --codec: the NeuCodec state dict goes through `neucodec_to_xcodec2_names` / `neucodec_encoder_to_xcodec2_names` non-strictly and
   every unmapped / missing / unrecognised key is PRINTED (the key maps were written from memory); the decoder is then
   loaded into the codec engine and `decode_code` of 100 codes is compared with `neucodec` itself when that package is importable,
   else with transformers' Xcodec2 decoder modules filled with the same tensors (relative RMS bar 1.5e-2, tests/test_gpu_codec.py).
Exit status 0 = everything that could be checked passed.  tests/test_tools.py runs steps 1-2 on a synthetic `save_pretrained` directory."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "neutts-air_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def bf16_ulp(x):
    x = abs(float(x))
    return 2.0 ** -133 if x == 0 else 2.0 ** (np.floor(np.log2(x)) - 7)


def report_config(path):
    from transformers import AutoConfig
    from neutts.neutts import _engine_config_from_hf
    import synthetic as syn
    hc = AutoConfig.from_pretrained(path)
    cfg = _engine_config_from_hf(hc)
    print(f"[config] model_type {hc.model_type}: engine config {json.dumps(cfg)}")
    for name, ref in (("neutts_air()", syn.BackboneConfig.neutts_air()), ("neutts_nano_like() (ASSUMED geometry)", syn.BackboneConfig.neutts_nano_like())):
        want = dict(vocab_size=ref.vocab_size, hidden_size=ref.hidden_size, intermediate_size=ref.intermediate_size, num_layers=ref.num_layers,
                    num_heads=ref.num_heads, num_kv_heads=ref.num_kv_heads, tie_word_embeddings=ref.tie_word_embeddings, attention_bias=ref.attention_bias)
        delta = {k: (cfg.get(k), v) for k, v in want.items() if cfg.get(k) != v}
        print(f"[config] delta to synthetic.BackboneConfig.{name}: " + (json.dumps({k: {'checkpoint': a, 'assumed': b} for k, (a, b) in delta.items()}) if delta else "none"))
    return hc, cfg


def verify_backbone(path, steps, device, lib):
    from transformers import AutoModelForCausalLM
    from neutts import NeuTTS, _hip
    hc, cfg = report_config(path)
    tts = NeuTTS.__new__(NeuTTS)                 # the backbone half of NeuTTS.__init__ (no codec needed for this check)
    tts.max_context, tts._max_batch, tts._lib_path = 2048, 1, lib
    tts._load_backbone(path, device)
    tok = tts.tokenizer
    base, eos = tts._speech_base, tts._eos_id
    print(f"[tokenizer] id('<|speech_0|>') = {base}, id('<|SPEECH_GENERATION_END|>') = {eos}, vocabulary {len(tok)} (config {cfg['vocab_size']}), loader: {tts._backbone_loader}")
    ok = base is not None and base >= 0 and eos is not None and eos >= 0
    n_codes = min(65536, len(tok) - base)
    for k in (1, 77, n_codes - 1):
        if tok.convert_tokens_to_ids(f"<|speech_{k}|>") != base + k:
            print(f"[tokenizer] FAIL: id('<|speech_{k}|>') = {tok.convert_tokens_to_ids(f'<|speech_{k}|>')} != base + {k}: speech ids are not contiguous")
            ok = False
    # a prompt of the reference's shape (ref:neutts/neutts.py:303-332) without the phonemizer: chat header, some text, reference codes
    ids = tok.encode("user: Convert the text to speech:<|TEXT_PROMPT_START|>hello there, this is a test.<|TEXT_PROMPT_END|>\nassistant:<|SPEECH_GENERATION_START|>")
    rng = np.random.default_rng(0)
    prompt = list(ids) + [base + int(c) for c in rng.integers(0, n_codes, 48)]
    hf = AutoModelForCausalLM.from_pretrained(path, attn_implementation="eager").to(torch.bfloat16).eval()
    from neutts.neutts import _default_rope_theta
    theta, d = _default_rope_theta(hc), cfg["head_dim"]
    if theta is not None:    # from_pretrained(dtype=bf16) keeps this buffer in fp32; the post-hoc .to(bfloat16) above rounded it: restore
        inv = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.int64).to(dtype=torch.float) / d))
        hf.model.rotary_emb.inv_freq = inv
        hf.model.rotary_emb.original_inv_freq = inv
    out = hf.generate(torch.tensor([prompt]), max_length=len(prompt) + steps, eos_token_id=eos, pad_token_id=eos, do_sample=False, use_cache=True,
                      min_new_tokens=steps, output_scores=True, return_dict_in_generate=True)
    gold = out.sequences[0, len(prompt):].tolist()
    top = [torch.topk(s[0].float(), 4) for s in out.scores]
    eng = tts.backbone
    eng.set_debug(True)
    samp = _hip.Sampling(max_length=len(prompt) + steps, min_new_tokens=steps, eos_token_id=eos, do_sample=False)
    eng.prefill([prompt], [0], [samp])
    exact = ties = 0
    errs = []
    for k in range(len(gold)):
        if k > 0:
            eng.decode(1)
        got, _ = eng.read(0)
        row = eng.read_logits(0)
        tv, ti = top[k].values.numpy(), top[k].indices.numpy()
        errs += [abs(float(row[int(i)]) - float(v)) / bf16_ulp(v) for i, v in zip(ti, tv) if np.isfinite(v)]
        if got[-1] == gold[k]:
            exact += 1
            continue
        band = 4.0 * bf16_ulp(tv[0])
        cand = {int(i): float(v) for i, v in zip(ti, tv)}
        if got[-1] in cand and tv[0] - cand[got[-1]] <= band:
            ties += 1
            eng.debug_force(0, int(gold[k]))
        else:
            print(f"[backbone] FAIL at step {k}: engine {got[-1]}, transformers {gold[k]} (its top-4 {cand})")
            ok = False
            break
    e = np.array(errs)
    print(f"[backbone] {exact} of {len(gold)} teacher-forced ids equal to transformers' (+ {ties} near-ties of its own logits); logit error at its top-4 ids "
          f"in bf16 ulps: mean {e.mean():.3f} p99 {np.percentile(e, 99):.2f} max {e.max():.2f}  (bars 0.8 / 2.5 / 3.5)")
    ok = ok and exact + ties == len(gold) and e.mean() <= 0.8 and np.percentile(e, 99) <= 2.5 and e.max() <= 3.5
    eng.release(0)
    eng.close()
    return ok


def load_sd(path):
    if path.startswith("neuphonic/"):
        from neucodec import DistillNeuCodec, NeuCodec
        return (NeuCodec if path == "neuphonic/neucodec" else DistillNeuCodec).from_pretrained(path).state_dict()
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu")
    return sd.get("state_dict", sd) if isinstance(sd, dict) else sd.state_dict()


def verify_codec(path, device, lib):
    from neutts import _hip
    from neutts.neutts import neucodec_encoder_to_xcodec2_names, neucodec_to_xcodec2_names, _device_index
    sd = load_sd(path)
    print(f"[codec] {len(sd)} tensors in {path}")
    ok = True
    try:
        dec = neucodec_to_xcodec2_names(sd, strict=True)
        print(f"[codec] decoder key map: all {len(dec)} engine tensors found, no unrecognised decoder-side key")
    except ValueError as ex:
        print(f"[codec] decoder key map MISMATCH:\n{ex}")
        dec, ok = neucodec_to_xcodec2_names(sd, strict=False), False
    try:
        enc = neucodec_encoder_to_xcodec2_names(sd)
        print(f"[codec] encoder key map: {len(enc)} engine tensors found")
    except (ValueError, KeyError, StopIteration) as ex:
        print(f"[codec] encoder key map MISMATCH: {ex}")
        ok = False
    known = ("generator.", "fc_post_a.", "fc_prior.", "semantic_model.", "SemanticEncoder_module.", "CodecEnc.")
    print("[codec] top-level prefixes of the checkpoint: " + ", ".join(sorted({k.split('.')[0] for k in sd})))
    stray = sorted(k for k in sd if not k.startswith(known))
    if stray:
        print(f"[codec] {len(stray)} keys under no known prefix: " + ", ".join(stray[:20]))
    eng = _hip.CodecEngine(dict(max_frames=256, max_rows=1024), _device_index(device, "device"), lib)
    eng.load_state_dict({k: (v.float() if hasattr(v, "float") else v) for k, v in dec.items()})
    codes = np.random.default_rng(1).integers(0, 65536, 100)
    ours = eng.decode([codes.tolist()])[0]
    ref = None
    try:
        from neucodec import NeuCodec
        m = NeuCodec.from_pretrained("neuphonic/neucodec").eval()
        with torch.no_grad():
            ref = m.decode_code(torch.tensor(codes, dtype=torch.long)[None, None, :])[0, 0].numpy()
        src = "neucodec.NeuCodec.decode_code"
    except ImportError:
        from transformers import Xcodec2Config
        from transformers.models.xcodec2.modeling_xcodec2 import Xcodec2Decoder, Xcodec2Quantizer
        xc = Xcodec2Config(downsampling_ratios=(2, 2, 4, 5, 6), rope_parameters={"rope_type": "default", "rope_theta": 10000.0})
        q, d = Xcodec2Quantizer(xc).eval(), Xcodec2Decoder(xc).eval()
        q.load_state_dict({k[len("quantizer."):]: v.float() for k, v in dec.items() if k.startswith("quantizer.")}, strict=False)
        missing, unexpected = d.load_state_dict({k[len("decoder."):]: v.float() for k, v in dec.items() if k.startswith("decoder.")}, strict=False)
        print(f"[codec] transformers Xcodec2Decoder: missing {missing[:8]}, unexpected {unexpected[:8]}")
        with torch.no_grad():
            ref = d(q.from_codes(torch.tensor(codes, dtype=torch.long)[None, :]))[0, 0].numpy()
        src = "transformers Xcodec2Quantizer + Xcodec2Decoder @ hop 480 (neucodec is not importable: the neucodec == xcodec2 equivalence stays unverified)"
    err = float(np.sqrt(np.mean((ours.astype(np.float64) - ref[: len(ours)]) ** 2)))
    sig = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    print(f"[codec] decode_code of 100 codes vs {src}: RMS error {err:.3e} at signal RMS {sig:.3e} (relative {err / sig:.3e}; bar 1.5e-2)")
    return ok and err <= 1.5e-2 * sig


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("backbone")
    ap.add_argument("--codec", default=None)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--lib", default=None)
    a = ap.parse_args(argv)
    ok = verify_backbone(a.backbone, a.steps, a.device, a.lib)
    if a.codec:
        ok = verify_codec(a.codec, a.device, a.lib) and ok
    print("[verify_checkpoint] " + ("PASS" if ok else "FAIL"))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
