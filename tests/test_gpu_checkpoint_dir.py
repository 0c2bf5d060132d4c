"""The reference's loading path (ref:neutts/neutts.py:163-166,186-191) on the MI355X: a `save_pretrained` directory (Qwen2 with
bias + tied head; Llama-style without bias, untied head) at NeuTTS-Air's WIDTH goes through `NeuTTS(backbone_repo=dir)` --
safetensors shards -> arena, tokenizer ids, `_apply_chat_template` -- and the greedy ids equal transformers' own `generate` on
the same directory.  Bodies shared with the emulator twin: tests/ckpt_dir_cases.py."""
import pytest

import ckpt_dir_cases as cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    return cases.make_ckpt(tmp_path_factory.mktemp("neutts_air_width_ckpt"), wide=True)


def test_checkpoint_dir_through_the_reference_loading_path(ckpt, hip_lib):
    cases.run_checkpoint_dir_case(ckpt, hip_lib)


def test_llama_style_checkpoint_dispatch(tmp_path, hip_lib):
    cases.run_llama_dispatch_case(tmp_path, hip_lib, wide=True)


def test_qwen3_style_checkpoint_dispatch(tmp_path, hip_lib):
    """Qwen3-0.6B's attention geometry (hidden 1024, 16:8 heads of head_dim 128, FFN 3072), 2 layers."""
    cases.run_qwen3_dispatch_case(tmp_path, hip_lib, hidden=1024, heads=16, kv_heads=8, ffn=3072, layers=2)
