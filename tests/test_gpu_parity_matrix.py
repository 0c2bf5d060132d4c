"""Logits-level decode parity on RANDOM-init weights, MI355X through the C-ABI.

Round 4's free-running id tests run on "walk" weights whose next id barely depends on attention or on what the KV pages hold.
This file is the sensitive net under the decode-path variants: NeuTTS-Air WIDTH (hidden 896, 14 / 2 heads of 64; 2 layers and a
3000-row vocabulary keep the CPU oracle cheap), N(0, 1 / fan_in) weights, and at every decode step the engine's fp32 logits row is
compared with the oracle's (`oracle/backbone_ref.py`, pinned bit for bit to transformers' Qwen2: tests/test_oracle_pin.py) at the
oracle's top-4 ids, in bf16 ulps of the oracle's value -- the bar `test_air_teacher_forced_vs_hf_golden` uses at full size: mean <= 0.8,
max <= 3.5.  The decode launches under test hold EIGHT DIFFERENT prompts of ragged lengths that straddle the 32-token page edges
(31 / 32 / 33, 63 / 64 / 65, 500, 900) in the first, middle and last m-block of the batch, next to filler sequences of other lengths,
on KV pages that a previous occupant of every slot left dirty.  Teacher-forced (the oracle's id is fed back wherever a near-tie
legitimately flips the argmax), so all 24 steps stay comparable.

hf:models/qwen2/modeling_qwen2.py:150-172 (eager attention), :176-234 (attention block); ref:neutts/neutts.py:338-347 (the call).
"""
import os

import numpy as np
import pytest
import torch

from oracle import backbone_ref as br
from neutts import _hip
from common import bf16_ulp, make_engine

pytestmark = pytest.mark.gpu

CFG = br.BackboneConfig(vocab_size=3000, hidden_size=896, intermediate_size=1216, num_layers=2)
LENS = [31, 32, 33, 63, 64, 65, 500, 900]
N = 24
EOS = CFG.vocab_size - 1
MAX_CTX = 960


@pytest.fixture(scope="module")
def lib(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _hip.load_library(hip_lib)
    return hip_lib


class Gold:
    def __init__(self, prompt, ref):
        self.prompt = prompt
        self.ids = list(ref.ids)
        top = [torch.topk(lg, 4) for lg in ref.logits]
        self.topv = [t.values.numpy() for t in top]
        self.topi = [t.indices.numpy() for t in top]


@pytest.fixture(scope="module")
def model():
    w = br.make_weights(CFG, 41)                    # init="unit": N(0, 1 / fan_in) matrices -- attention and the KV contents matter
    wd = br.cast_weights(w, torch.bfloat16)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    golds = []
    for i, n in enumerate(LENS):
        p = br.synthetic_prompt(CFG, 300 + i, n)
        golds.append(Gold(p, br.generate(CFG, wd, p, n + N, EOS, min_new_tokens=N, keep_logits=True)))
    assert len({tuple(g.ids) for g in golds}) == len(golds)          # eight different continuations
    return w, wd, golds


def samp_for(p, n=N, **kw):
    return _hip.Sampling(max_length=len(p) + n, min_new_tokens=n, eos_token_id=EOS, do_sample=False, **kw)


def checked_slots(B):
    """First, middle and last m-block of the batch (64-row blocks on the tile path)."""
    want = [0, 1, B // 2 - 1, B // 2, B // 2 + 1, B - 3, B - 2, B - 1]
    out = []
    for s in want:
        s = min(max(s, 0), B - 1)
        if s not in out:
            out.append(s)
    for s in range(B):                               # small batches: whatever slots are left, up to eight
        if len(out) >= min(8, B):
            break
        if s not in out:
            out.append(s)
    return out[:8]


def prefill_chunks(eng, prompts, slots, samps, budget=4096):
    i = 0
    while i < len(prompts):
        j, used = i, 0
        while j < len(prompts) and (j == i or used + len(prompts[j]) <= budget):
            used += len(prompts[j])
            j += 1
        eng.prefill(prompts[i:j], slots[i:j], samps[i:j])
        i = j


def fill(eng, B, slots, prompts, seed0, n_dec):
    """The checked prompts in `slots`, filler sequences (other lengths, other contents) in every third of the remaining slots."""
    fillers = [s for s in range(B) if s not in slots][::3]
    fp = [br.synthetic_prompt(CFG, seed0 + s, 20 + (7 * s) % 50) for s in fillers]
    prefill_chunks(eng, list(prompts) + fp, list(slots) + fillers, [samp_for(p, n_dec) for p in list(prompts) + fp])
    return fillers


def teacher_forced_many(eng, slots, golds, n, max_ulps=2.0, also=()):
    """common.teacher_forced_compare for several slots of ONE decode launch: every step, each checked slot's logits row against its
    own golden run.  `also`: engines to advance in step (a gang's other chains)."""
    stats, exact, tie = [], 0, 0
    for k in range(n):
        if k > 0:
            for e in (eng,) + tuple(also):
                e.decode(1)
        for s, g in zip(slots, golds):
            ids, _ = eng.read(s)
            assert len(ids) == k + 1, (s, k, len(ids))
            row = eng.read_logits(s)
            for i, v in zip(g.topi[k], g.topv[k]):
                if np.isfinite(v):
                    stats.append(abs(float(row[int(i)]) - float(v)) / bf16_ulp(float(v)))
            tok = ids[-1]
            if tok == g.ids[k]:
                exact += 1
            else:
                band = max_ulps * bf16_ulp(g.topv[k][0])
                cand = {int(i): float(v) for i, v in zip(g.topi[k], g.topv[k])}
                assert tok in cand and g.topv[k][0] - cand[tok] <= band, (
                    f"slot {s} (prompt of {len(g.prompt)}) step {k}: got {tok}, golden {g.ids[k]}, golden top4 {cand}, band {band}")
                tie += 1
                if k + 1 < n:
                    eng.debug_force(s, g.ids[k])
    return exact, tie, np.asarray(stats)


def check_stats(tag, exact, tie, stats, n_rows):
    print(f"parity-matrix {tag}: {exact} exact + {tie} near-tie of {n_rows} steps; logits error at the oracle's top-4: "
          f"mean {stats.mean():.3f}, p99 {np.percentile(stats, 99):.2f}, max {stats.max():.2f} bf16 ulps")
    assert exact + tie == n_rows and exact >= 0.8 * n_rows, (tag, exact, tie)
    assert stats.mean() <= 0.8 and stats.max() <= 3.5, (tag, stats.mean(), stats.max())


CASES = (
    [(b, {}) for b in (1, 8, 16, 40, 64, 96, 128, 256, 512, 640, 768, 1024)] +                    # small-batch path (<= 8), tile path, every attention form by batch size; from 512 the wide shape (640 = bench.py's engines, part-filled last tiles)
    [(8, {"NTTS_SMALL_BATCH": "0"})] +                                                             # the tile path at 8 rows
    [(b, {"NTTS_ATTN_SPLIT": "8", "NTTS_ATTN_SPLIT_CTX": "64"}) for b in (1, 16, 64, 128)] +       # context-split attention from context 64 on: two launches + statistics (+ combine pass)
    [(64, {"NTTS_XCD_AFFINE": "7"}), (128, {"NTTS_XCD_AFFINE": "7"}),                              # row-block XCD placement where it is not the default ...
     (256, {"NTTS_XCD_AFFINE": "0"}), (512, {"NTTS_XCD_AFFINE": "0"})] +                           # ... and off where it is
    [(16, {"NTTS_TALL": "3", "NTTS_XCD_AFFINE": "0"}), (256, {"NTTS_TALL": "3", "NTTS_XCD_AFFINE": "0"}),   # round 5: the gang's 256-row o_proj / down_proj tiles
     (40, {"NTTS_TALL": "3", "NTTS_GU_TILE": "1"}), (128, {"NTTS_GU_TILE": "3", "NTTS_KS_O": "2", "NTTS_KS_D": "8"}),
     (256, {"NTTS_TALL": "3", "NTTS_XCD_AFFINE": "0", "NTTS_QKV_WSTAT": "1"}), (96, {"NTTS_XCD_AFFINE": "0", "NTTS_QKV_WSTAT": "1"})])   # QKV column blocks dealt to XCDs


@pytest.mark.parametrize("max_batch,knobs", CASES, ids=[f"b{b}-" + ("default" if not k else "-".join(f"{a[5:].lower()}{v}" for a, v in k.items())) for b, k in CASES])
def test_decode_logits_vs_oracle_ragged_dirty_pages(lib, model, max_batch, knobs, monkeypatch):
    w, wd, golds = model
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    B = max_batch
    eng = make_engine(CFG, w, lib, max_batch=B, max_context=MAX_CTX, max_prefill_tokens=4096, bf16_upload=True)
    try:
        eng.set_debug(True)
        slots = checked_slots(B)
        rounds = [list(range(len(slots)))] if B >= 8 else [[2, 7][r:r + B] for r in range(0, 2, B)]   # batch 1: the 33- and the 900-token prompt, one after the other
        tot_e = tot_t = 0
        allst = []
        for which in rounds:
            gs = [golds[i] for i in which]
            sl = slots[:len(gs)]
            # a previous occupant of every slot: other prompts of the same lengths (the page pool is last-in first-out: the checked prompts
            # get exactly these pages back), decoded past the positions the checked run will write, then released
            junk = [br.synthetic_prompt(CFG, 900 + i, len(g.prompt)) for i, g in enumerate(gs)]
            fillers = fill(eng, B, sl, junk, 5000, N + 2)
            eng.decode(N + 1)
            eng.release_many(sl + fillers)
            fillers = fill(eng, B, sl, [g.prompt for g in gs], 7000, N)
            e_, t_, st = teacher_forced_many(eng, sl, gs, N)
            tot_e += e_; tot_t += t_; allst.append(st)
            eng.release_many(sl + fillers)
        check_stats(f"batch {B} {knobs or 'default'}", tot_e, tot_t, np.concatenate(allst), N * sum(len(r) for r in rounds))
    finally:
        eng.close()


def test_decode_logits_one_engine_of_a_gang(lib, model):
    """Engine 1 of a gang of three 256-slot engines on one arena (lane streams, the gang's decode shape -- ntts_backbone_set_gang:
    256-row o_proj / down_proj tiles, XCD placement off), the other two chains decoding beside it: the same logits-level check."""
    w, wd, golds = model
    B = 256
    eng = make_engine(CFG, w, lib, max_batch=B, max_context=MAX_CTX, max_prefill_tokens=4096, bf16_upload=True)
    gang = _hip.EngineGang(eng, 3)
    try:
        slots = checked_slots(B)
        for e in gang.engines:
            e.set_debug(True)
        others = [gang.engines[0], gang.engines[2]]
        for k, e in enumerate(others):
            fill(e, B, [], [], 11000 + 1000 * k, N + 4)
        me = gang.engines[1]
        fillers = fill(me, B, slots, [g.prompt for g in golds], 7000, N)
        e_, t_, st = teacher_forced_many(me, slots, golds, N, also=others)
        check_stats("one engine of a gang of three", e_, t_, st, N * len(slots))
        me.release_many(slots + fillers)
    finally:
        gang.close()
        eng.close()


def test_generate_with_slot_and_page_recycling_logits(lib, model, monkeypatch):
    """BackboneEngine.generate (continuous batching: 40 ragged requests through 16 slots, the tile path, slots and pages changing
    hands all the time): every request's ids follow the oracle's logits ALONG ITS OWN PATH (each id is the oracle's argmax or sits
    within the near-tie band), and the LAST step's logits row of four late requests -- all on recycled slots and pages -- is within
    the ulp bar of the oracle's logits for that very step."""
    w, wd, golds = model
    monkeypatch.setenv("NTTS_SMALL_BATCH", "0")
    eng = make_engine(CFG, w, lib, max_batch=16, max_context=256, max_prefill_tokens=1024, bf16_upload=True)
    try:
        eng.set_debug(True)
        rng = np.random.default_rng(5)
        R = 40
        plen = [int(x) for x in rng.integers(20, 130, size=R)]
        nnew = [int(x) for x in rng.integers(6, 20, size=R)]
        prompts = [br.synthetic_prompt(CFG, 400 + i, plen[i]) for i in range(R)]
        samp = [_hip.Sampling(max_length=plen[i] + nnew[i], min_new_tokens=nnew[i], eos_token_id=EOS, do_sample=False) for i in range(R)]
        seen = {}

        def hook(i, slot, n_new):
            seen[i] = (slot, eng.read(slot)[0], eng.read_logits(slot).copy())
        eng.generate(prompts, samp, steps_per_poll=1, run_ahead=False, on_finished=hook, prefill_token_budget=400)
        assert sorted(seen) == list(range(R))
        assert len({seen[i][0] for i in range(R)}) <= 16 and max(sum(1 for i in range(R) if seen[i][0] == s) for s in range(16)) >= 2   # slots did change hands
        stats = []
        for i in (R - 1, R - 2, R - 5, R - 9, 3):
            slot, ids, row = seen[i]
            assert len(ids) == nnew[i]
            ref = br.generate(CFG, wd, prompts[i], plen[i] + nnew[i], EOS, min_new_tokens=nnew[i], force_ids=ids, keep_logits=True)
            for k, (tok, lg) in enumerate(zip(ids, ref.logits)):
                top = torch.topk(lg, 2).values
                assert float(lg[tok]) >= float(top[0]) - 2.0 * bf16_ulp(float(top[0])), (i, k, tok, float(lg[tok]), float(top[0]))
            last = torch.topk(ref.logits[-1], 4)
            for j, v in zip(last.indices.numpy(), last.values.numpy()):
                stats.append(abs(float(row[int(j)]) - float(v)) / bf16_ulp(float(v)))
        stats = np.asarray(stats)
        print(f"parity-matrix generate with recycling: last-step logits of 5 requests, mean {stats.mean():.3f} max {stats.max():.2f} bf16 ulps")
        assert stats.mean() <= 0.8 and stats.max() <= 3.5
    finally:
        eng.close()


def test_both_decode_shapes_of_a_256_row_engine_give_the_same_bits(lib, model):
    """ABI 9: an engine counts the decode chains of its arena at every decode call and switches between the single-chain shape and the gang's
    (256-row o_proj / down_proj tiles, QKV column blocks per XCD, no row-block placement) with a captured graph per shape -- which is only safe
    if the two shapes compute the same BITS (same K slices, same summation order per output element).  256 slots, the eight ragged prompts +
    fillers, 6 steps: every checked slot's logits row identical between ntts_backbone_set_gang(1), (4), and (0) = counting (alone: the
    single-chain shape)."""
    w, wd, golds = model
    B = 256
    eng = make_engine(CFG, w, lib, max_batch=B, max_context=MAX_CTX, max_prefill_tokens=4096, bf16_upload=True)
    try:
        eng.set_debug(True)
        slots = checked_slots(B)
        rows = {}
        for chains in (1, 4, 0):
            eng.set_gang(chains)
            fillers = fill(eng, B, slots, [g.prompt for g in golds], 7000, 8)
            got = []
            for k in range(6):
                if k:
                    eng.decode(1)
                got.append([eng.read_logits(s).copy() for s in slots])
            rows[chains] = got
            eng.release_many(slots + fillers)
        for k in range(6):
            for j in range(len(slots)):
                assert np.array_equal(rows[1][k][j], rows[4][k][j]), (k, j)
                assert np.array_equal(rows[1][k][j], rows[0][k][j]), (k, j)
    finally:
        eng.set_gang(0)
        eng.close()


def test_shared_prefix_slots_logits_vs_oracle(lib, model):
    """prefix KV sharing (ntts_backbone_prefill_shared) against the ORACLE directly, not against
    the engine's own plain prefill: six utterances of one "speaker" -- a common 200-token beginning, then 37 ... 120 tokens of their own -- are
    prefilled with five of them re-using the first one's KV pages (192 tokens = 6 whole pages); the donor is released after two steps; every
    step's logits row of the five SHARING slots (and of the donor while it lives) against the oracle's run of the full prompt, teacher-forced,
    at the bars of the matrix above."""
    w, wd, golds = model
    rng = np.random.default_rng(77)
    head = rng.integers(0, CFG.vocab_size - 1, 200).tolist()
    tails = [37, 64, 65, 100, 120, 90]
    prompts = [head + rng.integers(0, CFG.vocab_size - 1, t).tolist() for t in tails]
    n = 10
    gs = [Gold(p, br.generate(CFG, wd, p, len(p) + n, EOS, min_new_tokens=n, keep_logits=True)) for p in prompts]
    eng = make_engine(CFG, w, lib, max_batch=16, max_context=384, max_prefill_tokens=4096, bf16_upload=True)
    try:
        eng.set_debug(True)
        slots = [3, 0, 7, 15, 8, 12]
        st0 = eng.kv_stats()
        eng.prefill(prompts, slots, [samp_for(p, n) for p in prompts], [None] + [(slots[0], 200)] * 5)
        assert eng.kv_stats()["prompt_tokens_shared"] - st0["prompt_tokens_shared"] == 5 * 192
        e1, t1, s1 = teacher_forced_many(eng, slots, gs, 2)                       # donor + sharers, first token and one decode step
        eng.release(slots[0])                                                     # the shared pages outlive their donor
        stats, exact, tie = [s1], e1, t1
        for k in range(2, n):
            eng.decode(1)
            for s, g in zip(slots[1:], gs[1:]):
                ids, _ = eng.read(s)
                assert len(ids) == k + 1
                row = eng.read_logits(s)
                stats.append(np.asarray([abs(float(row[int(i)]) - float(v)) / bf16_ulp(float(v)) for i, v in zip(g.topi[k], g.topv[k]) if np.isfinite(v)]))
                if ids[-1] == g.ids[k]:
                    exact += 1
                else:
                    cand = {int(i): float(v) for i, v in zip(g.topi[k], g.topv[k])}
                    assert ids[-1] in cand and g.topv[k][0] - cand[ids[-1]] <= 2.0 * bf16_ulp(g.topv[k][0]), (s, k)
                    tie += 1
                    if k + 1 < n:
                        eng.debug_force(s, g.ids[k])
        check_stats("shared-prefix slots vs the oracle", exact, tie, np.concatenate(stats), 2 * 6 + (n - 2) * 5)
        eng.release_many(slots[1:])
        st = eng.kv_stats()
        assert st["free_pages"] == st["total_pages"]
    finally:
        eng.close()


@pytest.mark.parametrize("temperature", [0.7, 1.5])
def test_sampling_draw_with_temperature(lib, model, temperature, monkeypatch):
    """north_star "greedy/temperature sampling": the device sampler's draw at temperature != 1 (the reference passes 1.0,
    ref:neutts/neutts.py:343; TemperatureLogitsWarper hf:generation/logits_process.py:250-305 divides the scores by T before top-k)
    token for token against its written specification (oracle/sampling_ref.sample_topk) on the engine's own logits rows -- first token
    and three decode steps, 16 requests with seeds of their own, the tile path's sampler."""
    from oracle.sampling_ref import sample_topk
    w, wd, golds = model
    monkeypatch.setenv("NTTS_SMALL_BATCH", "0")
    eng = make_engine(CFG, w, lib, max_batch=16, max_context=256, max_prefill_tokens=4096, bf16_upload=True)
    try:
        eng.set_debug(True)
        prompts = [golds[i % 6].prompt for i in range(16)]
        seeds = [9100 + 37 * i for i in range(16)]
        samp = [_hip.Sampling(max_length=len(p) + 5, min_new_tokens=5, eos_token_id=EOS, do_sample=True, top_k=50, temperature=temperature, seed=s)
                for p, s in zip(prompts, seeds)]
        eng.prefill(prompts, list(range(16)), samp)
        exact = total = 0
        for step in range(4):
            if step:
                eng.decode(1)
            for s in range(16):
                ids, _ = eng.read(s)
                assert len(ids) == step + 1
                row = eng.read_logits(s)
                assert row[ids[-1]] >= np.sort(row)[-50]
                want, margin = sample_topk(row, 50, temperature, seeds[s], step)
                total += 1
                if margin > 1e-5:
                    assert ids[-1] == want, (temperature, step, s, ids[-1], want, margin)
                    exact += 1
        assert exact >= total - 4, (exact, total)
        greedy = [int(np.argmax(eng.read_logits(s))) for s in range(16)]
        assert sum(int(eng.read(s)[0][-1] != greedy[s]) for s in range(16)) >= (2 if temperature > 1 else 0)    # (it does sample)
    finally:
        eng.close()
