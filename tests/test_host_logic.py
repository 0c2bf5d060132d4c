"""Host-side logic of the NeuTTS class surface that needs no engine: the NeuCodec checkpoint key mapping (strict), the
HF-config dispatch, the slot pool of the engine wrapper."""
import numpy as np
import pytest
import torch

from oracle import codec_ref as cr


def _neucodec_style(sd):
    """xcodec2-named decoder weights -> the original `neucodec` key layout of SURVEY.md B.4 (fused c_attn)."""
    out = {}
    ren = {"quantizer.project_out.": "generator.quantizer.project_out.", "decoder.fc.": "fc_post_a.", "decoder.embed.": "generator.backbone.embed.",
           "decoder.norm.": "generator.backbone.final_layer_norm.", "decoder.head.linear.": "generator.head.out."}
    layers = {}
    for k, v in sd.items():
        for a, b in ren.items():
            if k.startswith(a):
                out[b + k[len(a):]] = v
                break
        else:
            if k.startswith("decoder.prior_net.") or k.startswith("decoder.post_net."):
                out["generator.backbone." + k[len("decoder."):]] = v
            elif k.startswith("decoder.layers."):
                i, rest = k[len("decoder.layers."):].split(".", 1)
                layers.setdefault(int(i), {})[rest] = v
            else:
                raise AssertionError(k)
    for i, d in layers.items():
        p = f"generator.backbone.transformers.{i}."
        out[p + "att.c_attn.weight"] = torch.cat([d["self_attn.q_proj.weight"], d["self_attn.k_proj.weight"], d["self_attn.v_proj.weight"]])
        out[p + "att.c_proj.weight"] = d["self_attn.o_proj.weight"]
        out[p + "att_norm.weight"] = d["input_layernorm.weight"]
        out[p + "ffn_norm.weight"] = d["post_attention_layernorm.weight"]
        out[p + "mlp.fc1.weight"], out[p + "mlp.fc2.weight"] = d["mlp.fc1.weight"], d["mlp.fc2.weight"]
    out["generator.head.istft.window"] = torch.hann_window(16)          # a buffer: ignored
    out["generator.quantizer.project_in.weight"] = torch.zeros(8, 8)    # encoder side: ignored
    out["encoder.something.weight"] = torch.zeros(2)                    # not decoder-side: ignored
    return out


def test_neucodec_key_mapping_round_trip_and_strictness():
    from neutts.neutts import neucodec_to_xcodec2_names
    cfg = cr.CodecConfig.tiny()
    w = cr.make_weights(cfg, 1)
    nsd = _neucodec_style(w)
    back = neucodec_to_xcodec2_names(nsd)
    assert set(back) == set(w)
    for k in w:
        assert torch.equal(torch.as_tensor(back[k]), w[k]), k
    # a missing source key is named, not skipped
    broken = dict(nsd)
    broken.pop("generator.backbone.transformers.0.mlp.fc2.weight")
    with pytest.raises(ValueError, match=r"missing source keys.*transformers\.0\.mlp\.fc2\.weight"):
        neucodec_to_xcodec2_names(broken)
    # a decoder-side tensor the mapping does not know is reported (a renamed layer must not be dropped silently)
    extra = dict(nsd)
    extra["generator.backbone.transformers.0.mlp.fc3.weight"] = torch.zeros(2, 2)
    with pytest.raises(ValueError, match=r"unrecognised decoder-side keys.*fc3"):
        neucodec_to_xcodec2_names(extra)
    assert "decoder.layers.0.mlp.fc1.weight" in neucodec_to_xcodec2_names(extra, strict=False)
    with pytest.raises(ValueError, match="missing source keys"):
        neucodec_to_xcodec2_names({"generator.head.out.weight": torch.zeros(2, 2)})


def _neucodec_style_encoder(w, cfg, extra_layers=2, parametrize=False):
    """The encoder half of a neucodec state dict as the neucodec / xcodec2 sources lay it out (nested nn.Sequential
    containers, weight-normalised convolutions, the full-depth w2v-BERT), built from xcodec2-named weights."""
    out = {}
    for k, v in w.items():
        if k.startswith("semantic_encoder."):
            out["semantic_model." + k[len("semantic_encoder."):]] = v
    last = cfg.sem_layers - 1
    for k, v in list(out.items()):                       # layers 17..24 of the real model: present, never used
        if f".layers.{last}." in k:
            for j in range(extra_layers):
                out[k.replace(f".layers.{last}.", f".layers.{cfg.sem_layers + j}.")] = v + 1.0
    out["semantic_model.masked_spec_embed"] = torch.zeros(cfg.sem_hidden)
    for dst, src in (("initial_conv.weight", "conv1.weight"), ("residual_blocks.1.weight", "conv2.weight"),
                     ("residual_blocks.1.bias", "conv2.bias"), ("residual_blocks.3.weight", "conv3.weight"),
                     ("residual_blocks.3.bias", "conv3.bias"), ("final_conv.weight", "conv4.weight")):
        out["SemanticEncoder_module." + dst] = w["semantic_adapter." + src]
    out["fc_prior.weight"], out["fc_prior.bias"] = w["fc_encoder.weight"], w["fc_encoder.bias"]
    out["generator.quantizer.project_in.weight"] = w["quantizer.project_in.weight"]
    out["generator.quantizer.project_in.bias"] = w["quantizer.project_in.bias"]

    def wn(prefix, name):
        v = w[name + ".weight"]
        g = torch.linalg.vector_norm(v, dim=(1, 2), keepdim=True)               # weight norm: w = g * v / ||v||
        vv = v * 3.0                                                            # any positive scale of the direction tensor
        gk, vk = ("parametrizations.weight.original0", "parametrizations.weight.original1") if parametrize else ("weight_g", "weight_v")
        out[f"{prefix}.{gk}"], out[f"{prefix}.{vk}"] = g, vv
        out[prefix + ".bias"] = w[name + ".bias"]

    def snake(prefix, name):
        out[prefix + ".act.alpha"], out[prefix + ".act.beta"] = w[name + ".act.alpha"], w[name + ".act.beta"]
        out[prefix + ".upsample.filter"] = torch.zeros(1, 1, 12)
        out[prefix + ".downsample.lowpass.filter"] = torch.zeros(1, 1, 12)

    wn("CodecEnc.conv_blocks.0", "acoustic_encoder.conv1")
    for bi in range(len(cfg.ratios)):
        b, hb = f"CodecEnc.conv_blocks.{bi + 1}.block", f"acoustic_encoder.block.{bi}."
        for u in range(3):
            snake(f"{b}.{u}.block.0", f"{hb}res_unit{u + 1}.snake1")
            wn(f"{b}.{u}.block.1", f"{hb}res_unit{u + 1}.conv1")
            snake(f"{b}.{u}.block.2", f"{hb}res_unit{u + 1}.snake2")
            wn(f"{b}.{u}.block.3", f"{hb}res_unit{u + 1}.conv2")
        snake(f"{b}.3", hb + "snake1")
        wn(f"{b}.4", hb + "conv1")
    snake("CodecEnc.conv_final_block.0", "acoustic_encoder.snake1")
    wn("CodecEnc.conv_final_block.1", "acoustic_encoder.conv2")
    return out


@pytest.mark.parametrize("parametrize", [False, True])
def test_neucodec_encoder_key_mapping_round_trip_and_strictness(parametrize):
    import synthetic as syn
    from neutts.neutts import neucodec_encoder_to_xcodec2_names
    cfg = syn.EncoderConfig.tiny()
    w = syn.make_encoder_weights(cfg, 5)
    nsd = _neucodec_style_encoder(w, cfg, parametrize=parametrize)
    back = neucodec_encoder_to_xcodec2_names(nsd, n_layers=cfg.sem_layers)
    assert set(back) == set(w), (set(back) ^ set(w))
    for k in w:
        assert torch.allclose(torch.as_tensor(back[k]).reshape(w[k].shape), w[k], rtol=1e-6, atol=1e-7), k   # weight norm folded
    broken = dict(nsd)
    broken.pop("fc_prior.bias")
    with pytest.raises(ValueError, match=r"fc_prior\.bias"):
        neucodec_encoder_to_xcodec2_names(broken, n_layers=cfg.sem_layers)
    broken = {k: v for k, v in nsd.items() if not k.startswith("CodecEnc.conv_blocks.3.block.1.")}   # one residual unit gone
    with pytest.raises(ValueError, match="ENCODER layout|not the convolution expected|channels"):
        neucodec_encoder_to_xcodec2_names(broken, n_layers=cfg.sem_layers)
    swapped = dict(nsd)                                                                            # a conv of the wrong kernel size
    key = [k for k in nsd if k.startswith("CodecEnc.conv_blocks.2.block.0.block.1.") and k.endswith(("weight_v", "original1"))][0]
    swapped[key] = nsd[key][:, :, :5]
    with pytest.raises(ValueError, match="not the convolution expected"):
        neucodec_encoder_to_xcodec2_names(swapped, n_layers=cfg.sem_layers)


def test_hf_config_dispatch():
    from transformers import LlamaConfig, Qwen2Config
    from neutts.neutts import _engine_config_from_hf
    q = _engine_config_from_hf(Qwen2Config(hidden_size=896, num_attention_heads=14, num_key_value_heads=2, intermediate_size=4864,
                                           num_hidden_layers=24, vocab_size=1000, tie_word_embeddings=True))
    assert q["attention_bias"] is True and q["tie_word_embeddings"] is True and q["head_dim"] == 64 and q["num_kv_heads"] == 2
    l = _engine_config_from_hf(LlamaConfig(hidden_size=768, num_attention_heads=12, num_key_value_heads=4, intermediate_size=2048,
                                           num_hidden_layers=2, vocab_size=1000, tie_word_embeddings=False))
    assert l["attention_bias"] is False and l["tie_word_embeddings"] is False
    l128 = _engine_config_from_hf(LlamaConfig(hidden_size=1024, num_attention_heads=8, num_key_value_heads=8, vocab_size=100))
    assert l128["head_dim"] == 128 and l128["qk_norm"] is False          # round 6: the general attention path
    with pytest.raises(NotImplementedError, match="head_dim 256"):
        _engine_config_from_hf(LlamaConfig(hidden_size=1024, num_attention_heads=4, num_key_value_heads=4, vocab_size=100))
    from transformers import Qwen3Config
    q3 = _engine_config_from_hf(Qwen3Config(hidden_size=1024, num_attention_heads=16, num_key_value_heads=8, head_dim=128, vocab_size=100))
    assert q3["qk_norm"] is True and q3["head_dim"] == 128 and q3["attention_bias"] is False and q3["num_kv_heads"] == 8

    class Other:
        model_type = "gpt2"
    with pytest.raises(NotImplementedError, match="model_type 'gpt2'"):
        _engine_config_from_hf(Other())


def test_engine_slot_pool_and_generate_cleanup(emu_lib):
    """a failing request must not strand slots; streams draw their slot from the same pool as generate()."""
    from oracle import backbone_ref as br
    from neutts import _hip
    from common import engine_cfg
    cfg = br.BackboneConfig.tiny(vocab_size=256, num_layers=1)
    w = br.make_weights(cfg, 2, walk_gain=4.0)
    eng = _hip.BackboneEngine(engine_cfg(cfg, max_batch=2, max_context=64, max_prefill_tokens=96, num_pages=3), 0, emu_lib)
    eng.load_state_dict({k: v.numpy() for k, v in w.items()}, inv_freq=br.rope_inv_freq(cfg).numpy())
    eos = cfg.vocab_size - 1
    ok = _hip.Sampling(max_length=40, min_new_tokens=4, eos_token_id=eos, do_sample=False)
    # up-front validation: nothing is admitted when one request cannot run
    with pytest.raises(ValueError, match="max_length"):
        eng.generate([[1, 2, 3], list(range(50))], [ok, ok])
    assert eng.free_slots() == 2 and eng.kv_stats()["free_pages"] == 3
    # KV pool too small for both at once (3 pages, each request needs 2): the second waits for the first, both finish
    p = [br.synthetic_prompt(cfg, i, 30) for i in range(2)]
    got = eng.generate(p, [_hip.Sampling(max_length=40, min_new_tokens=10, eos_token_id=eos, do_sample=False)] * 2)
    assert [len(g) for g in got] == [10, 10] and eng.free_slots() == 2 and eng.kv_stats()["free_pages"] == 3
    # a single request the pool can never hold fails up front -- and leaves the engine usable
    eng2 = _hip.BackboneEngine(engine_cfg(cfg, max_batch=2, max_context=64, max_prefill_tokens=96, num_pages=1), 0, emu_lib)
    with pytest.raises(_hip.NeuTTSHipError, match="KV pages"):
        eng2.generate([br.synthetic_prompt(cfg, 3, 40)], [_hip.Sampling(max_length=64, min_new_tokens=24, eos_token_id=eos, do_sample=False)])
    eng2.close()
    assert eng.free_slots() == 2 and eng.kv_stats()["free_pages"] == 3
    assert len(eng.generate([[5, 6, 7]], [ok])[0]) >= 4
    # a held slot (an open stream) is never handed out again
    s = eng.acquire_slot()
    assert eng.free_slots() == 1
    got = eng.generate([[1, 2, 3], [4, 5, 6]], [ok, ok])          # two requests through the one remaining slot
    assert all(len(g) >= 4 for g in got)
    eng.release(s)
    assert eng.free_slots() == 2


def test_bench_engine_slots_rule():
    """bench.py sizes the static gang's engines from the job: even gang steps, at most 1024 slots, whole 64-row tiles, never
    less than a batch."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f = bench.engine_slots_for
    assert [f(k, 256, 4) for k in (1, 3, 4, 5, 8, 10, 16, 20, 32, 40)] == [256, 256, 256, 320, 512, 640, 1024, 640, 1024, 896]
    assert f(20, 256, 2) == 896 and f(2, 2, 2) == 64 and f(20, 512, 4) == 896
    for k in range(1, 70):
        b = f(k, 256, 4)
        n_gs = -(-(k * 256) // (4 * 1024))
        assert 256 <= b <= 1024 and b % 64 == 0 and 4 * b * n_gs >= k * 256        # the gang steps hold the job
