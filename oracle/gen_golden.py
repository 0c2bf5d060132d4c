"""Generate tests/golden/*.npz from the REAL third-party implementation the reference calls.

Run in the build container (needs `transformers`; nothing here runs on the GPU box):
    python -m oracle.gen_golden [--full] [--codec]
    python -m oracle.gen_golden --walk          # the free-running fixtures (permutation-walk weights): small + NeuTTS-Air size

Backbone: transformers.Qwen2ForCausalLM driven exactly like ref:neutts/neutts.py:338-347
(`generate(..., use_cache=True, min_new_tokens=...)`, greedy so the result is reproducible),
attention pinned to "eager", weights = oracle.backbone_ref.make_weights (numpy PCG64, so the
tests can rebuild the same tensors anywhere).  `inv_freq` is kept in fp32, which is what
`from_pretrained(dtype=bfloat16)` does (a post-hoc `.to(bfloat16)` would round the buffer).

Fixtures hold inputs + outputs only (ids, top-2 margins, a few logits rows): weights are
regenerated from the seed.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import backbone_ref as br  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def hf_backbone(cfg: br.BackboneConfig, w, dtype):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    kw = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
              intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_layers,
              num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
              max_position_embeddings=32768, rms_norm_eps=cfg.rms_eps, tie_word_embeddings=True,
              rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta},
              attn_implementation="eager")
    if getattr(cfg, "qk_norm", False):          # Qwen3-style: per-head q/k RMSNorm, explicit head_dim, bias-free projections
        from transformers import Qwen3Config, Qwen3ForCausalLM
        assert not cfg.attention_bias
        hc = Qwen3Config(head_dim=cfg.head_dim, attention_bias=False, **kw)
        with torch.device("meta"):
            m = Qwen3ForCausalLM(hc)
    else:
        hc = Qwen2Config(**kw)
        with torch.device("meta"):
            m = Qwen2ForCausalLM(hc)
    m = m.to_empty(device="cpu").eval()
    sd = {k: v.to(dtype) for k, v in w.items()}
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    m = m.to(dtype)
    m.load_state_dict(sd, strict=True, assign=True)
    m.tie_weights()
    m.model.rotary_emb.inv_freq = br.rope_inv_freq(cfg)          # fp32, see module docstring
    m.model.rotary_emb.original_inv_freq = br.rope_inv_freq(cfg)
    return m


def run_hf(m, prompt, n_new, eos, min_new):
    out = m.generate(torch.tensor([prompt]), max_length=len(prompt) + n_new, eos_token_id=eos,
                     pad_token_id=eos, do_sample=False, use_cache=True, min_new_tokens=min_new,
                     output_scores=True, return_dict_in_generate=True)
    ids = out.sequences[0, len(prompt):].tolist()
    top = [torch.topk(s[0].float(), 4) for s in out.scores]
    tv = np.stack([t.values.numpy() for t in top])
    ti = np.stack([t.indices.numpy() for t in top])
    return ids, tv, ti


def backbone_fixture(name, cfg, seed, utts, s_len, n_new, min_new, dtypes, init="unit", peak_sigma=0.0, walk_gain=0.0, walk_scale=4.0):
    w = br.make_weights(cfg, seed, init=init, peak_sigma=peak_sigma, walk_gain=walk_gain, walk_scale=walk_scale)
    eos = cfg.vocab_size - 1
    rec = dict(cfg=np.array(list(cfg.to_dict().items()), dtype=object), seed=seed, eos=eos, s_len=s_len,
               n_new=n_new, min_new=min_new, utts=np.array(utts), init=init, peak_sigma=np.float32(peak_sigma),
               walk_gain=np.float32(walk_gain), walk_scale=np.float32(walk_scale))
    for dtype, tag in dtypes:
        m = hf_backbone(cfg, w, dtype)
        for u in utts:
            prompt = br.synthetic_prompt(cfg, u, s_len)
            t = time.time()
            ids, tv, ti = run_hf(m, prompt, n_new, eos, min_new)
            ulps = (tv[:, 0] - tv[:, 1]) / 2.0 ** (np.floor(np.log2(np.abs(tv[:, 0]))) - 7)
            print(f"[{name}] {tag} utt {u}: {len(ids)} ids ({len(set(ids))} distinct) in {time.time() - t:.1f}s, "
                  f"min top1-top2 margin {float((tv[:, 0] - tv[:, 1]).min()):.4g} = {float(ulps.min()):.1f} bf16 ulps of the top logit")
            rec[f"{tag}_ids_{u}"] = np.array(ids, dtype=np.int64)
            rec[f"{tag}_topv_{u}"] = tv.astype(np.float32)
            rec[f"{tag}_topi_{u}"] = ti.astype(np.int64)
        del m
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also the NeuTTS-Air-size fixture (minutes, ~6 GB RAM)")
    ap.add_argument("--codec", action="store_true", help="codec-decoder fixtures (xcodec2 @ hop 480)")
    ap.add_argument("--walk", action="store_true", help="ONLY the permutation-walk fixtures (VERDICT r3 item 2): small + NeuTTS-Air size, 8 utterances")
    a = ap.parse_args()
    if a.walk:
        # greedy decoding walks a seeded permutation of the vocabulary (synthetic._make_walk): 250 DIFFERENT ids per utterance, every
        # top-1 / top-2 margin tens of bf16 ulps wide -> free-running ids comparable id for id, no tie clause
        cfgw = br.BackboneConfig(vocab_size=2048, hidden_size=896, intermediate_size=1216, num_layers=2)
        backbone_fixture("backbone_small_walk", cfgw, 1, [0, 1], 70, 60, 60, [(torch.bfloat16, "bf16")], walk_gain=4.0, walk_scale=4.0)
        backbone_fixture("backbone_air_walk8", br.BackboneConfig.neutts_air(), 0, list(range(8)), 500, 250, 250, [(torch.bfloat16, "bf16")],
                         walk_gain=8.0, walk_scale=8.0)
        return
    torch.manual_seed(0)
    both = [(torch.float32, "fp32"), (torch.bfloat16, "bf16")]
    backbone_fixture("backbone_tiny", br.BackboneConfig.tiny(), 0, [0, 1, 2], 37, 40, 10, both)
    # GQA with 2 kv heads + odd prompt lengths crossing a KV page boundary
    cfg2 = br.BackboneConfig(vocab_size=2048, hidden_size=896, intermediate_size=1216, num_layers=2)
    backbone_fixture("backbone_small", cfg2, 1, [0, 1], 70, 30, 30, both)
    if a.full:
        bf = [(torch.bfloat16, "bf16")]
        backbone_fixture("backbone_air", br.BackboneConfig.neutts_air(), 0, [0, 1, 2, 3], 500, 250, 250, bf)
    if a.codec:
        from oracle import gen_golden_codec
        gen_golden_codec.main()


if __name__ == "__main__":
    main()
