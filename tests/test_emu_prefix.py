"""Prefix sharing (include/neutts_hip.h: ntts_backbone_prefill_shared) on the CPU SIMT emulator.
The property: a prompt prefilled on top of another slot's shared KV pages produces EXACTLY what the plain prefill of
the whole prompt produces -- first-token logits bit for bit, greedy ids token for token -- because every K/V row and
every query sees the same operands in the same order; plus page accounting (reference counts) and error paths."""
import numpy as np
import pytest

from oracle import backbone_ref as br
from neutts import _hip
from common import make_engine


def _prompts(cfg):
    rng = np.random.default_rng(5)
    tok = lambda n: rng.integers(0, cfg.vocab_size - 1, n).tolist()   # noqa: E731
    prefix = tok(75)
    a = prefix + tok(20)                 # donor: 95 tokens, 2 full pages of common prefix
    b = prefix + tok(33)                 # shares 75 -> 64 cached, 44 computed
    c = a[:70]                           # a strict prefix of the donor: 69 shareable -> 64 cached, 6 computed
    d = prefix[:40] + tok(30)            # shares 40 -> 32 cached
    e = tok(50)                          # unrelated
    f = prefix + tok(3)                  # named as sharing MORE than is common with its donor (b): checked below
    return prefix, [a, b, c, d, e, f]


def _run(eng, prompts, donors, n_new, eos):
    samp = [_hip.Sampling(max_length=len(p) + n_new, min_new_tokens=n_new, eos_token_id=eos, do_sample=False) for p in prompts]
    slots = list(range(len(prompts)))
    eng.set_debug(True)
    eng.prefill(prompts, slots, samp, donors)
    logits = [eng.read_logits(s).copy() for s in slots]
    eng.decode(n_new - 1)
    ids = [eng.read(s)[0] for s in slots]
    for s in slots:
        eng.release(s)
    eng.set_debug(False)
    return logits, ids


def test_shared_prefill_is_bit_identical_to_plain_prefill(emu_lib):
    cfg = br.BackboneConfig.tiny(vocab_size=512, num_layers=2)
    w = br.make_weights(cfg, 21, walk_gain=4.0)
    prefix, prompts = _prompts(cfg)
    eos = cfg.vocab_size - 1
    eng = make_engine(cfg, w, emu_lib, max_batch=6, max_context=160, max_prefill_tokens=512)
    total = eng.kv_stats()["total_pages"]
    want_logits, want_ids = _run(eng, prompts, None, 8, eos)
    st0 = eng.kv_stats()
    assert st0["free_pages"] == total and st0["prompt_tokens_shared"] == 0
    # same call: slot 0 is the donor of 1, 2, 3; slot 5 shares with slot 1 (itself a sharer: chains work)
    donors = [None, (0, 75), (0, 70), (0, 40), None, (1, 75)]
    got_logits, got_ids = _run(eng, prompts, donors, 8, eos)
    st1 = eng.kv_stats()
    assert st1["free_pages"] == total, "every page returns to the pool once its last owner is released"
    assert st1["prompt_tokens_shared"] - st0["prompt_tokens_shared"] == 64 + 64 + 32 + 64
    assert st1["prompt_tokens_computed"] - st0["prompt_tokens_computed"] == sum(map(len, prompts)) - (64 + 64 + 32 + 64)
    for i in range(len(prompts)):
        assert np.array_equal(got_logits[i], want_logits[i]), f"prompt {i}: first-token logits differ"
        assert got_ids[i] == want_ids[i], f"prompt {i}"
    eng.close()


def test_donor_from_an_earlier_call_and_released_first(emu_lib):
    """The donor was prefilled earlier and has already decoded; it is released while its sharers still run: the shared
    pages stay alive (reference counts) and the sharers' results do not change."""
    cfg = br.BackboneConfig.tiny(vocab_size=512, num_layers=1)
    w = br.make_weights(cfg, 22, walk_gain=4.0)
    prefix, prompts = _prompts(cfg)
    a, b, c = prompts[0], prompts[1], prompts[2]
    eos = cfg.vocab_size - 1
    n_new = 40                                         # crosses page boundaries during decode
    mk = lambda p: _hip.Sampling(max_length=len(p) + n_new, min_new_tokens=n_new, eos_token_id=eos, do_sample=False)  # noqa: E731
    eng = make_engine(cfg, w, emu_lib, max_batch=3, max_context=192, max_prefill_tokens=256)
    total = eng.kv_stats()["total_pages"]
    # reference run: plain prefill, one at a time
    want = {}
    for name, p in (("b", b), ("c", c)):
        eng.prefill([p], [0], [mk(p)])
        eng.decode(n_new - 1)
        want[name] = eng.read(0)[0]
        eng.release(0)
    eng.prefill([a], [2], [mk(a)])
    eng.decode(5)                                      # the donor is mid-generation when the others arrive
    eng.prefill([b, c], [0, 1], [mk(b), mk(c)], [(2, 75), (2, 69)])
    used_shared = total - eng.kv_stats()["free_pages"]
    assert used_shared == 4 + (4 - 2) + (3 - 2), "a: 95 + 5 tokens -> 4 pages; b (108 tokens) and c (70) only add their own tails"
    eng.decode(3)
    eng.release(2)                                     # donor gone; pages 0-1 of its prompt live on
    assert total - eng.kv_stats()["free_pages"] == 2 + (4 - 2) + (3 - 2)
    eng.decode(n_new - 1 - 3)
    assert eng.read(0)[0] == want["b"] and eng.read(1)[0] == want["c"]
    eng.release(0)
    assert total - eng.kv_stats()["free_pages"] == 4   # c (70 + 39 tokens) alone still holds the two shared pages + its own two
    eng.release(1)
    assert eng.kv_stats()["free_pages"] == total
    eng.close()


def test_shared_prefill_error_paths(emu_lib):
    cfg = br.BackboneConfig.tiny(vocab_size=512, num_layers=1)
    w = br.make_weights(cfg, 23)
    prefix, prompts = _prompts(cfg)
    a, b = prompts[0], prompts[1]
    eos = cfg.vocab_size - 1
    mk = lambda p: _hip.Sampling(max_length=len(p) + 4, min_new_tokens=1, eos_token_id=eos, do_sample=False)  # noqa: E731
    eng = make_engine(cfg, w, emu_lib, max_batch=3, max_context=160, max_prefill_tokens=256)
    total = eng.kv_stats()["total_pages"]
    with pytest.raises(_hip.NeuTTSHipError, match="holds no prompt"):
        eng.prefill([b], [0], [mk(b)], [(2, 64)])                 # donor slot is free
    eng.prefill([a], [2], [mk(a)])
    with pytest.raises(_hip.NeuTTSHipError, match="does not start with"):
        eng.prefill([b], [0], [mk(b)], [(2, 80)])                 # b and a differ from token 75 on
    with pytest.raises(_hip.NeuTTSHipError, match="out of range"):
        eng.prefill([b], [0], [mk(b)], [(7, 64)])
    assert total - eng.kv_stats()["free_pages"] == 3, "failed calls leave no pages behind"
    eng.prefill([b], [0], [mk(b)], [(2, 10)])                     # less than a page in common: nothing shared, still fine
    assert eng.kv_stats()["prompt_tokens_shared"] == 0
    for s in (0, 2):
        eng.release(s)
    assert eng.kv_stats()["free_pages"] == total
    eng.close()


def test_generate_with_share_prefix_equals_without(emu_lib):
    """Continuous batching (more prompts than slots) over two 'speakers': same ids with and without sharing."""
    cfg = br.BackboneConfig.tiny(vocab_size=512, num_layers=1)
    w = br.make_weights(cfg, 24, walk_gain=4.0)
    rng = np.random.default_rng(9)
    tok = lambda n: rng.integers(0, cfg.vocab_size - 1, n).tolist()   # noqa: E731
    spk = [tok(70), tok(45)]
    prompts = [spk[i % 2] + tok(5 + 3 * i) for i in range(7)]
    eos = cfg.vocab_size - 1
    samp = [_hip.Sampling(max_length=len(p) + 6, min_new_tokens=6, eos_token_id=eos, do_sample=False) for p in prompts]
    eng = make_engine(cfg, w, emu_lib, max_batch=3, max_context=128, max_prefill_tokens=256)
    want = eng.generate(prompts, samp, steps_per_poll=2)
    assert eng.kv_stats()["prompt_tokens_shared"] == 0
    got = eng.generate(prompts, samp, steps_per_poll=2, share_prefix=True)
    st = eng.kv_stats()
    assert got == want
    assert st["prompt_tokens_shared"] >= 64 + 32, st
    assert st["free_pages"] == st["total_pages"]
    eng.close()
