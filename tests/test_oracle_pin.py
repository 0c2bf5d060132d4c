"""Pin the oracle to the LIVE third-party implementation (transformers Qwen2ForCausalLM, eager attention):
bit-identical logits in fp32 and bf16.  Skipped where transformers is not importable (it is in the
build container; the committed golden vectors carry the pin everywhere else)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import backbone_ref as br

transformers = pytest.importorskip("transformers")


@pytest.mark.parametrize("arch", ["qwen2", "qwen3"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bit_exact_vs_hf(dtype, arch):
    """qwen2: NeuTTS-Air's family (transformers.Qwen2ForCausalLM).  qwen3 (round 6): the oracle's qk_norm / head_dim 128 switch against
    transformers.Qwen3ForCausalLM -- per-head q/k RMSNorm before RoPE, q width != hidden, scaling 128^-0.5 (not a power of two: the
    product with the bf16 scores rounds), bias-free projections."""
    from oracle.gen_golden import hf_backbone
    if arch == "qwen3":
        cfg = br.BackboneConfig(vocab_size=512, hidden_size=512, intermediate_size=640, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=128,
                                attention_bias=False, qk_norm=True)
    else:
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


def test_fp8_linear_vs_torch_scaled_mm():
    """The fp8 variant has no third-party MODEL to pin against (oracle/backbone_ref.py header: parity unpinned), but its one arithmetic
    building block has a third-party implementation in this image: torch._scaled_mm on CPU (oneDNN) -- e4m3 x e4m3 products, fp32
    accumulation, per-tensor scales.  The oracle's `linear` on fp8 entries -- e4m3(x / in_scale) @ e4m3(W / w_scale[n])^T, scaled by
    in_scale * w_scale[n], bias added, one rounding to bf16 -- must agree with it: same e4m3 operands bit for bit (torch's own
    conversion), results equal up to the order of the fp32 sums (at most one bf16 ulp, on a few elements in a thousand)."""
    if not hasattr(torch, "_scaled_mm"):
        pytest.skip("torch._scaled_mm not available")
    torch.manual_seed(5)
    M, K, N = 48, 896, 320
    x = (torch.randn(M, K) * 1.7).to(torch.bfloat16)
    wt = torch.randn(N, K) * 0.05
    wt[7] = 0                                                               # an all-zero output row: scale 1, result = bias
    bias = torch.randn(N).to(torch.bfloat16)
    name = "model.layers.0.self_attn.o_proj.weight"
    scales = {f"model.layers.0.{t}.input_scale": float(x.float().abs().max()) / br.FP8_MAX * 1.1
              for t in ("self_attn.q_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.down_proj")}
    scales["lm_head.input_scale"] = 1.0
    w = {f"model.layers.0.{t}.weight": wt for t in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                                                     "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")}
    w["model.embed_tokens.weight"] = torch.randn(16, K)
    wq = br.fp8_quantize_weights(w, scales)
    got = br.linear(x, wq, name, bias)
    assert got.dtype == torch.bfloat16
    xs = scales["model.layers.0.self_attn.o_proj.input_scale"]
    a8 = (x.float() * (1.0 / xs)).clamp(-br.FP8_MAX, br.FP8_MAX).to(torch.float8_e4m3fn)
    b8 = wq[name + "::q"].to(torch.float8_e4m3fn)
    assert torch.equal(b8.float(), wq[name + "::q"]) and torch.equal(a8.float(), br.fp8_act(x, xs))      # the stored VALUES are e4m3 numbers
    try:
        acc = torch._scaled_mm(a8, b8.t(), scale_a=torch.tensor(1.0), scale_b=torch.tensor(1.0), out_dtype=torch.float32)
    except (RuntimeError, NotImplementedError) as ex:
        pytest.skip(f"torch._scaled_mm has no CPU path here: {ex}")
    want = (acc.double() * (xs * wq[name + "::scale"].double()) + bias.double()).float().to(torch.bfloat16)
    assert torch.equal(want[:, 7], bias.expand(M, N)[:, 7])
    neq = got != want
    assert neq.float().mean() < 5e-3, float(neq.float().mean())
    ulp = torch.abs(got.view(torch.int16).int() - want.view(torch.int16).int())
    assert int(ulp.max()) <= 1
    # and through the per-tensor scale path of the library itself (scale_a = in_scale): the same numbers before the per-row factor
    acc2 = torch._scaled_mm(a8, b8.t(), scale_a=torch.tensor(xs, dtype=torch.float32), scale_b=torch.tensor(1.0), out_dtype=torch.float32)
    assert torch.allclose(acc2, acc * xs, rtol=1e-6, atol=0)
