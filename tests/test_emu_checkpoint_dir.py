"""The reference's loading path (ref:neutts/neutts.py:163-166) on the CPU SIMT emulator: bodies in tests/ckpt_dir_cases.py
(shared with the MI355X twin tests/test_gpu_checkpoint_dir.py)."""
import pytest

import ckpt_dir_cases as cases


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    return cases.make_ckpt(tmp_path_factory.mktemp("neutts_tiny_ckpt"))


def test_checkpoint_dir_through_the_reference_loading_path(ckpt, emu_lib):
    cases.run_checkpoint_dir_case(ckpt, emu_lib)


def test_llama_style_checkpoint_dispatch(tmp_path, emu_lib):
    cases.run_llama_dispatch_case(tmp_path, emu_lib)


def test_qwen3_style_checkpoint_dispatch(tmp_path, emu_lib):
    cases.run_qwen3_dispatch_case(tmp_path, emu_lib)
