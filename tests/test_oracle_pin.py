"""Pin the oracle to the LIVE third-party implementation (transformers Qwen2ForCausalLM, eager attention):
bit-identical logits in fp32 and bf16.  Skipped where transformers is not importable (it is in the
build container; the committed golden vectors carry the pin everywhere else)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import backbone_ref as br

transformers = pytest.importorskip("transformers")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bit_exact_vs_hf(dtype):
    from oracle.gen_golden import hf_backbone
    cfg = br.BackboneConfig(vocab_size=512, hidden_size=896, intermediate_size=640, num_layers=2)
    w = br.make_weights(cfg, 7)
    m = hf_backbone(cfg, w, dtype)
    wd = br.cast_weights(w, dtype)
    prompt = br.synthetic_prompt(cfg, 3, 41)
    with torch.no_grad():
        o = m(torch.tensor([prompt]), output_hidden_states=True)
        cache = br.KVCache(cfg.num_layers)
        h = F.embedding(torch.tensor([prompt]), wd["model.embed_tokens.weight"])
        cos, sin = br.rope_cos_sin(cfg, torch.arange(len(prompt)), dtype)
        assert torch.equal(h, o.hidden_states[0])
        for i in range(cfg.num_layers):
            h = br.decoder_layer(cfg, wd, i, h, cos, sin, cache)
            if i + 1 < cfg.num_layers:
                assert torch.equal(h, o.hidden_states[i + 1]), f"layer {i}"
        hn = br.rms_norm(h, wd["model.norm.weight"], cfg.rms_eps)
        assert torch.equal(F.linear(hn, wd["model.embed_tokens.weight"]), o.logits)
    eos = cfg.vocab_size - 1
    out = m.generate(torch.tensor([prompt]), max_length=len(prompt) + 12, eos_token_id=eos, pad_token_id=eos,
                     do_sample=False, use_cache=True, min_new_tokens=5, output_scores=True, return_dict_in_generate=True)
    r = br.generate(cfg, wd, prompt, len(prompt) + 12, eos, min_new_tokens=5, keep_logits=True)
    assert out.sequences[0, len(prompt):].tolist() == r.ids
    for k in range(1, len(r.ids)):
        assert torch.equal(out.scores[k][0], r.logits[k])
