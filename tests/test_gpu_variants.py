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
test_fp8_decode_step_logits_small_and_tile_path = cases.test_fp8_decode_step_logits_small_and_tile_path
test_fp8_walk_free_running_exact = cases.test_fp8_walk_free_running_exact
test_fp8_prequantised_checkpoint_equals_quantise_on_upload = cases.test_fp8_prequantised_checkpoint_equals_quantise_on_upload


@pytest.mark.parametrize("knobs", [k for k, _ in cases.HEAD_TILES], ids=[i for _, i in cases.HEAD_TILES])
def test_speech_range_head_is_the_full_head_inside_the_range(lib, knobs, monkeypatch):
    cases._speech_range_body(lib, knobs, monkeypatch)


test_fp8_calibration_from_a_bf16_engine = cases.test_fp8_calibration_from_a_bf16_engine


@pytest.mark.parametrize("batch", [64, 256])
def test_fp8_nano_width_batch64_vs_oracle(lib, batch):
    """hidden 768 / 12:4 heads / FFN 2048 (the assumed Nano widths), 2 layers, vocabulary 8192, batch 64 (the large-batch tile
    path incl. the 128 x 128 gate/up tile and the 256-wide lm_head tile) and batch 256 (XCD row-block placement of the fp8
    split-K GEMMs / norms / attention on, as at the benchmark's batch 512), 70-token prompts, 24 greedy tokens: first-token
    logits against the fp8 oracle (bar: cases.check_fp8_model) and identical rows for identical prompts (batch invariance)."""
    cfg = br.BackboneConfig(vocab_size=8192, hidden_size=768, intermediate_size=2048, num_layers=2, num_heads=12, num_kv_heads=4)
    base = [br.synthetic_prompt(cfg, i, 70) for i in range(4)]
    prompts = [base[i % 4] for i in range(batch)]
    rows, worst = cases.check_fp8_model(lib, cfg, prompts, 24, max_batch=batch)
    for s in range(batch):
        assert rows[s] == rows[s % 4], s


def test_fp8_nano_geometry_19_layers_batch512(lib):
    """BASELINE.json configs[4] at its own size: the ASSUMED NeuTTS-Nano geometry (hidden 768, 19 layers, 12:4 heads, FFN 2048,
    vocabulary 142 080), fp8 weights + GEMM inputs, batch 512 (one XCD per 64-row m-block, the 256 x 256 fp8 lm_head tile, 2048
    attention workgroups), 70-token prompts, 10 greedy tokens.  First-token logits of the four distinct prompts against the fp8
    oracle, and identical rows for identical prompts across all 512 slots (batch / slot invariance).  The fp8 noise grows with
    the number of quantisation points (tests/test_emu_variants.py::check_fp8_model: ~1.3 % of the logits each): 19 layers x 4
    GEMM inputs measured 0.097-0.108 relative RMS on MI355X (correlation 0.994-0.995) against 0.21-0.28 for the quantisation
    itself (fp8 oracle vs bf16 oracle) -> bar at 1.2 x the largest measured distance / correlation >= 0.99."""
    cfg = br.BackboneConfig.neutts_nano_like()
    base = [br.synthetic_prompt(cfg, i, 70) for i in range(4)]
    prompts = [base[i % 4] for i in range(512)]
    rows, worst = cases.check_fp8_model(lib, cfg, prompts, 10, max_batch=512, bar=FP8_BAR_19_LAYERS, corr_bar=0.99)   # (10 tokens: the CPU oracle's steps are this test's minute)
    for s in range(512):
        assert rows[s] == rows[s % 4], s


FP8_BAR_19_LAYERS = 0.13   # 1.2 x the largest distance measured on MI355X (0.097-0.108: profiles/r03b_pytest_gpu.log ... r05o); the quantisation itself is 0.21-0.28 there.  What the bar sees of a wrong scale: tests/test_gpu_config4.py
