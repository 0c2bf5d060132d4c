"""The CPU oracle reproduces the committed golden vectors (produced by oracle/gen_golden.py from
transformers' Qwen2ForCausalLM, the dependency the reference calls at ref:neutts/neutts.py:338-347)."""
import numpy as np
import pytest
import torch

from oracle import backbone_ref as br
from common import load_fixture

CASES = [("backbone_tiny", "fp32"), ("backbone_tiny", "bf16"), ("backbone_small", "fp32"), ("backbone_small", "bf16"),
         ("backbone_small_walk", "bf16")]


@pytest.mark.parametrize("name,tag", CASES)
def test_oracle_matches_golden(name, tag):
    z, cfg, w = load_fixture(name)
    wd = br.cast_weights(w, torch.float32 if tag == "fp32" else torch.bfloat16)
    S, N, mn, eos = int(z["s_len"]), int(z["n_new"]), int(z["min_new"]), int(z["eos"])
    for u in z["utts"]:
        r = br.generate(cfg, wd, br.synthetic_prompt(cfg, int(u), S), S + N, eos, min_new_tokens=mn, keep_logits=True)
        assert r.ids == z[f"{tag}_ids_{u}"].tolist()
        # decode-step logits are bit-identical to HF's processed scores (same shapes, same ops)
        for k in range(1, len(r.ids)):
            tv = torch.topk(r.logits[k], 4)
            assert np.array_equal(tv.values.numpy(), z[f"{tag}_topv_{u}"][k])
            assert np.array_equal(tv.indices.numpy(), z[f"{tag}_topi_{u}"][k])


def test_eos_and_min_new_tokens_contract():
    """EOS is masked for the first min_new_tokens tokens and stops generation afterwards
    (hf:generation/logits_process.py:164-236, stopping_criteria.py:534-582)."""
    z, cfg, w = load_fixture("backbone_tiny")
    wd = br.cast_weights(w, torch.float32)
    prompt = br.synthetic_prompt(cfg, 0, 20)
    free = br.generate(cfg, wd, prompt, 60, eos_id=cfg.vocab_size - 1, min_new_tokens=0)
    eos = free.ids[3]                       # make the 4th greedy token the EOS id
    r0 = br.generate(cfg, wd, prompt, 60, eos_id=eos, min_new_tokens=0)
    assert r0.ids[-1] == eos and len(r0.ids) <= 4
    r1 = br.generate(cfg, wd, prompt, 60, eos_id=eos, min_new_tokens=10)
    assert eos not in r1.ids[:10] and len(r1.ids) >= 10
    r2 = br.generate(cfg, wd, prompt, 26, eos_id=cfg.vocab_size - 1, min_new_tokens=50)
    assert len(r2.ids) == 6                 # MaxLengthCriteria: total length == max_length
