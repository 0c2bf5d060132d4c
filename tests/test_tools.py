"""tools/verify_checkpoint.py (the day-one validator for real checkpoints) exercised on what CAN be built offline:
a Qwen2 checkpoint directory + tokenizer written by transformers' own save_pretrained (tests/ckpt_dir_cases.py), on the SIMT emulator."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_verify_checkpoint_on_a_synthetic_save_pretrained_dir(emu_lib, tmp_path, capsys):
    import ckpt_dir_cases as cases
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import verify_checkpoint as vc
    d, cfg, w, tok, ccfg = cases.make_ckpt(tmp_path)
    rc = vc.main([d, "--steps", "10", "--lib", emu_lib, "--device", "cuda"])
    out = capsys.readouterr().out
    print(out)
    assert rc == 0 and "[verify_checkpoint] PASS" in out
    assert "delta to synthetic.BackboneConfig.neutts_air()" in out and "teacher-forced ids equal to transformers'" in out
