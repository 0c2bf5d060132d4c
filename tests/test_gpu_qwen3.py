"""General attention geometry (head_dim 128 / qk_norm) on a real MI355X through the C-ABI: tests/qwen3_cases.py at Qwen3-0.6B's attention
widths (hidden 1024, 16 query / 8 kv heads of head_dim 128, FFN 3072; 2 layers and a 3000-row vocabulary keep the CPU oracle cheap)."""
import pytest
import torch

from oracle import backbone_ref as br
from neutts import _hip
import qwen3_cases as cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _hip.load_library(hip_lib)
    return hip_lib


@pytest.mark.parametrize("batch", [1, 16, 256, 512])
def test_qwen3_geometry_decode_logits_vs_oracle(lib, batch):
    """Batch 1 (no GEMV step on this path: the tile kernels at one row), 16, 256 and 512 (the WIDE decode shape on top of the general
    attention path); prompts of 31 / 32 / 33 / 64 / 65 / 300 / 500 tokens spread from the first to the last row of the batch, dirty pages,
    12 teacher-forced steps."""
    cfg = br.BackboneConfig.qwen3_like(vocab_size=3000)
    cases.run_case(lib, cfg, [31, 32, 33, 64, 65, 300, 500], batch, 12)


def test_head_dim_64_with_qk_norm_and_head_dim_128_without(lib):
    """The two other corners of the switch at NeuTTS-Air's width: q/k norm on 64-wide heads WITH q/k/v biases, and 7 : 1 GQA groups of head_dim 128
    heads without the norm (q width 1792 != hidden 896)."""
    cases.run_case(lib, br.BackboneConfig(vocab_size=3000, hidden_size=896, intermediate_size=1216, num_layers=2, qk_norm=True), [31, 33, 65, 500], 16, 8)
    cases.run_case(lib, br.BackboneConfig(vocab_size=3000, hidden_size=896, intermediate_size=1216, num_layers=2, num_heads=14, num_kv_heads=2,
                                          head_dim=128, attention_bias=False), [31, 33, 65, 500], 16, 8)
