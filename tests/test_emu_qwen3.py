"""General attention geometry (head_dim 128 / qk_norm) on the SIMT emulator: tests/qwen3_cases.py at small widths."""
import pytest

from oracle import backbone_ref as br
import qwen3_cases as cases


@pytest.mark.parametrize("hd,qkn,bias", [(128, True, False), (64, True, True), (128, False, False)], ids=["qwen3-hd128-qknorm", "hd64-qknorm-bias", "llama-hd128"])
def test_general_attention_path_vs_oracle(emu_lib, hd, qkn, bias):
    """Three corners of the switch: Qwen3 (head_dim 128 + q/k norm, bias-free), q/k norm at NeuTTS-Air's head_dim with its q/k/v biases, and a
    Llama-style head_dim 128 without the norm.  hidden 256, 2:1 heads, FFN 512, 2 layers, vocabulary 384; prompts of 31 / 33 / 70 tokens
    (page edges, three pages), batch 3, 4 teacher-forced steps."""
    cfg = br.BackboneConfig(vocab_size=384, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, head_dim=hd,
                            attention_bias=bias, qk_norm=qkn)
    cases.run_case(emu_lib, cfg, [31, 33, 70], 3, 4)
