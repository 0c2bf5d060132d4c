"""Architecture / precision switches of the backbone engine (ABI 2) on a real MI355X through libneutts_hip.so: the bodies of
tests/test_emu_variants.py (untied head, bias-free projections, tied-head check, the fp8 model vs the fp8 oracle) plus the fp8
model at the width of the assumed NeuTTS-Nano geometry (BASELINE.json configs[4]), where the 256 x 256 / 128 x 128 / 64 x 64
fp8 tiles, the fp8 lm_head and the fp8-emitting norm / attention / SiLU epilogues all run at their production shapes."""
import numpy as np
import pytest
import torch

from oracle import backbone_ref as br
from neutts import _hip
from common import assert_free_run_matches, engine_cfg
import test_emu_variants as cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _hip.load_library(hip_lib)
    return hip_lib


test_untied_head_and_no_bias_match_oracle = cases.test_untied_head_and_no_bias_match_oracle
test_tied_head_must_equal_embedding = cases.test_tied_head_must_equal_embedding
test_fp8_model_matches_fp8_oracle = cases.test_fp8_model_matches_fp8_oracle
test_fp8_needs_its_input_scales = cases.test_fp8_needs_its_input_scales


def test_fp8_nano_width_batch64_vs_oracle(lib):
    """hidden 768 / 12:4 heads / FFN 2048 (the assumed Nano widths), 4 layers, vocabulary 8192, batch 64 (the large-batch tile
    path incl. the 128 x 128 gate/up tile and the 256-wide lm_head tile), 70-token prompts, 24 greedy tokens: first-token
    logits of every slot against the fp8 oracle, free-running ids tie-aware, and identical rows for identical prompts."""
    cfg = br.BackboneConfig(vocab_size=8192, hidden_size=768, intermediate_size=2048, num_layers=4, num_heads=12, num_kv_heads=4)
    w = br.make_weights(cfg, 29, peak_sigma=0.5)
    scales = br.default_fp8_input_scales(cfg)
    wq = br.fp8_quantize_weights(br.cast_weights(w, torch.bfloat16), scales)
    eng = _hip.BackboneEngine(engine_cfg(cfg, max_batch=64, max_context=128, max_prefill_tokens=64 * 70, weight_dtype="fp8"), 0, lib)
    eng.load_state_dict({k: v.numpy() for k, v in w.items()}, inv_freq=br.rope_inv_freq(cfg).numpy(), input_scales=scales)
    base = [br.synthetic_prompt(cfg, i, 70) for i in range(4)]
    prompts = [base[i % 4] for i in range(64)]
    eos, N = cfg.vocab_size - 1, 24
    samp = [_hip.Sampling(max_length=70 + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)] * 64
    want = [br.generate(cfg, wq, p, 70 + N, eos, min_new_tokens=N, keep_logits=True) for p in base]
    eng.set_debug(True)
    eng.prefill(prompts, list(range(64)), samp)
    errs = []
    for s in range(4):
        row, ref = eng.read_logits(s), want[s].logits[0].numpy()
        fin = np.isfinite(ref)
        errs.append(np.abs(row[fin] - ref[fin]) / np.array([br.bf16_ulp(v) for v in ref[fin]]))
    err = np.concatenate(errs)
    print(f"fp8 model, first-token logits vs the fp8 oracle (all {len(err)} vocabulary entries of 4 prompts), bf16 ulps: "
          f"mean {err.mean():.3f}  p99 {np.percentile(err, 99):.2f}  max {err.max():.2f}")
    assert err.mean() <= 0.6 and err.max() <= 4.0
    eng.set_debug(False)
    eng.decode(N - 1)
    rows = [eng.read(s)[0] for s in range(64)]
    for s in range(64):
        assert rows[s] == rows[s % 4], s
    exact = 0
    for s in range(4):
        assert_free_run_matches(rows[s], want[s])
        exact += rows[s] == want[s].ids
    print(f"fp8 model free run: {exact} of 4 prompts identical to the oracle over {N} tokens (the others leave at an exact tie)")
    eng.close()
