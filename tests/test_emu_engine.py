"""The whole backbone engine (C-ABI -> prefill -> decode steps -> sampling/bookkeeping) executed on the
CPU SIMT emulator and checked against the golden vectors / the oracle.  Same test bodies as the GPU
parity tests (tests/test_gpu_backbone.py), just a different library file."""
import numpy as np
import pytest
import torch

from oracle import backbone_ref as br
from neutts import _hip
from common import load_fixture, make_engine, teacher_forced_compare


def test_tiny_teacher_forced(emu_lib):
    z, cfg, w = load_fixture("backbone_tiny")
    S, N, mn, eos = int(z["s_len"]), 12, int(z["min_new"]), int(z["eos"])
    eng = make_engine(cfg, w, emu_lib, max_batch=2)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=mn, eos_token_id=eos, do_sample=False)
    eng.prefill([br.synthetic_prompt(cfg, 0, S), br.synthetic_prompt(cfg, 1, S)], [1, 0], [samp, samp])
    # slots are decoded together; compare slot 1 (utt 0) step by step, slot 0 (utt 1) at the end
    ex, tie = teacher_forced_compare(eng, 1, z["bf16_ids_0"][:N], z["bf16_topv_0"], z["bf16_topi_0"])
    assert ex + tie == N and ex >= N - 3
    ids1, fin1 = eng.read(0)
    assert fin1 and len(ids1) == N
    agree = sum(int(a == b) for a, b in zip(ids1, z["bf16_ids_1"][:N]))
    assert agree >= N - 2, (ids1, z["bf16_ids_1"][:N])


@pytest.mark.parametrize("knobs", [
    {},                                                                                    # small-batch path (gemv.h)
    {"NTTS_ATTN_SPLIT": "3", "NTTS_ATTN_SPLIT_CTX": "40"},                                                               # small-batch path, context-split attention from context 40 on (3 chunks: ragged page ranges, an empty chunk early on; the run crosses the switch)
    {"NTTS_SMALL_BATCH": "0", "NTTS_HEAD_TILE": "2",
     "NTTS_ATTN_SPLIT": "2", "NTTS_ATTN_SPLIT_CTX": "45"},   # large-batch path (gemm.h tiles, fused QKV + RoPE + K append), 256 x 256 lm_head tile, context-split attention + combine pass from context 45 on
    {"NTTS_SMALL_BATCH": "0", "NTTS_HEAD_TILE": "1"},        # ... 128 x 128 lm_head tile
    # the lm_head's natural-order tile (gemm.h TN = 6: 256 x 288, 12 waves, uneven LDS-DMA loader split, partial last tile)
        {"NTTS_SMALL_BATCH": "0", "NTTS_HEAD_TILE": "4"}])
def test_small_gqa2_page_crossing_walk_exact(emu_lib, knobs, monkeypatch):
    """2 kv heads, prompt of 70 (3 pages), decode crosses the 96-token page boundary; walk weights (wide margins, a new id every step) so the
    free-running greedy ids must be bit-identical to HF's -- on the small-batch decode path (wave-per-16-features GEMVs with
    the fused norm prologue, slabs reduced in the attention prologue) and on the large-batch path (QKV with RoPE and the K append in
    its epilogue, attention without a prologue, split-K of o/down reduced in the norm kernel) with every lm_head tile."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    z, cfg, w = load_fixture("backbone_small_walk")
    # 70 + 27 tokens cross the 96-token page boundary; the other two lm_head tiles of the large-batch path need no second crossing (8 tokens: on the
    # emulator a 256-row tile over the whole vocabulary costs seconds per step)
    S, N, eos = int(z["s_len"]), (30 if not knobs else 8 if knobs.get("NTTS_HEAD_TILE") in ("1", "4") else 27), int(z["eos"])
    eng = make_engine(cfg, w, emu_lib, max_batch=1)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    eng.prefill([br.synthetic_prompt(cfg, 0, S)], [0], [samp])
    eng.decode(N - 1)
    ids, fin = eng.read(0)
    assert fin and ids == z["bf16_ids_0"][:N].tolist()


@pytest.mark.parametrize("small_batch", ["8", "0"])
def test_short_sequence_beside_a_long_one_across_the_split_switch(emu_lib, monkeypatch, small_batch):
    """ADVICE r2 (see the GPU twin in tests/test_gpu_backbone.py): the split switch follows the longest context of the batch; a
    one-page sequence beside a long one runs the split kernels with empty chunks -- each row must still match the oracle's solo run."""
    for k, v in {"NTTS_SMALL_BATCH": small_batch, "NTTS_ATTN_SPLIT": "3", "NTTS_ATTN_SPLIT_CTX": "40"}.items():
        monkeypatch.setenv(k, v)
    z, cfg, w = load_fixture("backbone_small_walk")
    wd = br.cast_weights(w, torch.bfloat16)
    N, eos = 8, int(z["eos"])
    prompts = [br.synthetic_prompt(cfg, 3, 6), br.synthetic_prompt(cfg, 0, 70)]
    want = [br.generate(cfg, wd, p, len(p) + N, eos_id=eos, min_new_tokens=N, keep_logits=True) for p in prompts]
    eng = make_engine(cfg, w, emu_lib, max_batch=2)
    samp = [_hip.Sampling(max_length=len(p) + N, min_new_tokens=N, eos_token_id=eos, do_sample=False) for p in prompts]
    eng.prefill(prompts, [0, 1], samp)
    eng.decode(N - 1)
    for s in (0, 1):
        ids, fin = eng.read(s)
        assert fin and ids == want[s].ids, (s, ids, want[s].ids)


@pytest.mark.parametrize("caps", [("512", "1024"), ("64", "128"), ("32", "96"), ("0", "1024"), ("0", "0")])
def test_prompt_pass_attention_split_by_position(emu_lib, monkeypatch, caps):
    """Prompt-pass attention runs on three kernels, chosen by the query's POSITION alone (attn_prefill.h): below NTTS_PF_RES_CAP (512) the
    resident kernel (pages in LDS, one exp per score, 16-query blocks dealt out from both ends of the prompt), below NTTS_PF_DEEP_CAP (1024)
    the deep one (K resident, V^T through a ring, packed scores), from there on the two-sweep kernel.  A 150-token and a 37-token prompt with
    everything on the resident kernel, with cuts at 64 / 128 and 32 / 96 (the long prompt uses all three), with everything on the deep kernel
    and with everything on the two-sweep kernel: HF's ids bit for bit every time (walk weights)."""
    cap = caps
    monkeypatch.setenv("NTTS_PF_RES_CAP", caps[0])
    monkeypatch.setenv("NTTS_PF_DEEP_CAP", caps[1])
    z, cfg, w = load_fixture("backbone_small_walk")
    wd = br.cast_weights(w, torch.bfloat16)
    N, eos = 6, int(z["eos"])
    prompts = [br.synthetic_prompt(cfg, 2, 150), br.synthetic_prompt(cfg, 5, 37)]
    want = [br.generate(cfg, wd, p, len(p) + N, eos_id=eos, min_new_tokens=N, keep_logits=True) for p in prompts]
    eng = make_engine(cfg, w, emu_lib, max_batch=2, max_context=192)
    samp = [_hip.Sampling(max_length=len(p) + N, min_new_tokens=N, eos_token_id=eos, do_sample=False) for p in prompts]
    eng.prefill(prompts, [1, 0], samp)
    eng.decode(N - 1)
    for s, u in ((1, 0), (0, 1)):
        ids, fin = eng.read(s)
        assert fin and ids == want[u].ids, (cap, u, ids, want[u].ids)


def test_prompt_pass_attention_tiers_logits(emu_lib, monkeypatch):
    """The sensitive twin of the test above (walk ids barely depend on attention): RANDOM-init weights, first-token logits of a 150- and a
    37-token prompt against the oracle's bf16 run, for the two-sweep kernel alone, the default tiers (all resident here), everything on the
    deep kernel and cuts at 32 / 96 (all three kernels).  Every setting within the same distance of the oracle, and of each other."""
    cfg = br.BackboneConfig(vocab_size=600, hidden_size=128, intermediate_size=256, num_layers=3, num_heads=2, num_kv_heads=1)
    w = br.make_weights(cfg, 21)
    wd = br.cast_weights(w, torch.bfloat16)
    eos = cfg.vocab_size - 1
    prompts = [br.synthetic_prompt(cfg, 2, 150), br.synthetic_prompt(cfg, 5, 37)]
    ref = [br.generate(cfg, wd, p, len(p) + 1, eos, min_new_tokens=1, keep_logits=True).logits[0].double() for p in prompts]
    rows = {}
    for caps in (("0", "0"), ("512", "1024"), ("0", "1024"), ("32", "96")):
        monkeypatch.setenv("NTTS_PF_RES_CAP", caps[0])
        monkeypatch.setenv("NTTS_PF_DEEP_CAP", caps[1])
        eng = make_engine(cfg, w, emu_lib, max_batch=2, max_context=192)
        eng.set_debug(True)
        samp = [_hip.Sampling(max_length=len(p) + 2, min_new_tokens=2, eos_token_id=eos, do_sample=False) for p in prompts]
        eng.prefill(prompts, [1, 0], samp)
        rows[caps] = [torch.from_numpy(eng.read_logits(s)).double() for s in (1, 0)]
        eng.close()

    def rel(a, b):
        fin = torch.isfinite(a) & torch.isfinite(b)
        return float((a[fin] - b[fin]).norm() / b[fin].norm())
    for u in range(2):
        r0 = rel(rows[("0", "0")][u], ref[u])
        for caps, rr in rows.items():
            assert rel(rr[u], ref[u]) <= 1.5 * r0 + 2e-3, (caps, u, rel(rr[u], ref[u]), r0)
            assert rel(rr[u], rows[("0", "0")][u]) <= 2.0 * r0 + 2e-3, (caps, u)
        assert torch.equal(rows[("512", "1024")][u], rows[("0", "1024")][u])    # the resident and the deep kernel are the same arithmetic


def test_xcd_row_block_placement(emu_lib, monkeypatch):
    """NTTS_XCD_AFFINE=7 (the default above batch 128): split-K GEMMs, the norms behind them and decode attention place the rows of a
    64-row m-block on one group of XCDs (gemm.h xcd_maffine, norm.h xcd_row).  A pure permutation of which workgroup does what:
    at batch 64 (one m-block spread over all 8 XCDs, padding workgroups in the GEMM grids) two sequences in far-apart slots still
    give HF's ids bit for bit."""
    for k, v in {"NTTS_SMALL_BATCH": "0", "NTTS_XCD_AFFINE": "7"}.items():
        monkeypatch.setenv(k, v)
    z, cfg, w = load_fixture("backbone_small_walk")
    S, N, eos = int(z["s_len"]), 20, int(z["eos"])
    eng = make_engine(cfg, w, emu_lib, max_batch=64)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    eng.prefill([br.synthetic_prompt(cfg, 0, S), br.synthetic_prompt(cfg, 1, S)], [5, 40], [samp, samp])
    eng.decode(N - 1)
    for slot, u in ((5, 0), (40, 1)):
        ids, fin = eng.read(slot)
        assert fin and ids == z[f"bf16_ids_{u}"][:N].tolist(), (slot, ids)


def test_continuous_batching_ragged_vs_oracle(emu_lib):
    """More prompts than slots, ragged prompt lengths, EOS stop before max_length, slot recycling:
    every prompt's ids equal the oracle's single-sequence run (ref:neutts/neutts.py:334-352 is batch 1)."""
    cfg = br.BackboneConfig.tiny(vocab_size=512, num_layers=1)
    w = br.make_weights(cfg, 11, walk_gain=4.0)
    wd = br.cast_weights(w, torch.bfloat16)
    lens = [5, 33, 64, 17, 40]
    prompts = [br.synthetic_prompt(cfg, 10 + i, n) for i, n in enumerate(lens)]
    # choose an EOS that the model really emits for prompt 3 after a few tokens (and not before)
    probe = br.generate(cfg, wd, prompts[3], lens[3] + 12, eos_id=cfg.vocab_size - 1, min_new_tokens=0)
    eos = probe.ids[4]
    assert eos not in probe.ids[:4]
    want_r = [br.generate(cfg, wd, p, len(p) + 10, eos_id=eos, min_new_tokens=3, keep_logits=True) for p in prompts]
    want = [r.ids for r in want_r]
    assert any(len(x) < 10 for x in want), "test should exercise an early EOS stop"
    eng = make_engine(cfg, w, emu_lib, max_batch=2, max_prefill_tokens=128)
    samp = [_hip.Sampling(max_length=len(p) + 10, min_new_tokens=3, eos_token_id=eos, do_sample=False) for p in prompts]
    got = eng.generate(prompts, samp, steps_per_poll=3, prefill_token_budget=70)
    assert got == want                                    # walk weights: wide margins, id for id
    # the scheduler runs one burst ahead of its bookkeeping by default (poll_begin / poll_end / read_finished): the ids cannot depend
    # on WHEN the host notices a finished row -- same result with the blocking poll, and with a device-side hand-off hook
    assert eng.generate(prompts, samp, steps_per_poll=3, prefill_token_budget=70, run_ahead=False) == got
    assert eng.generate(prompts, samp, steps_per_poll=1, prefill_token_budget=70) == got
    seen = {}
    codes = np.zeros((len(prompts), 16), dtype=np.int32)
    lens_out = np.zeros(len(prompts), dtype=np.int32)

    def hook(i, slot, n_new):          # before the slot is released: export its ids on the "device" (the emulator's is host memory)
        seen[i] = n_new
        eng.export_codes([slot], 0, cfg.vocab_size, codes[i:i + 1].ctypes.data, 16, lens_out[i:i + 1].ctypes.data)
    assert eng.generate(prompts, samp, steps_per_poll=3, prefill_token_budget=70, on_finished=hook) == [[] for _ in prompts]
    eng.sync()
    assert seen == {i: len(g) for i, g in enumerate(got)}
    for i, g in enumerate(got):
        assert lens_out[i] == len(g) and codes[i, :len(g)].tolist() == g
    # the same requests dealt out over a GANG of three engines on one arena (EngineGang: schedulers advanced in turn, decode chains
    # side by side on lane streams): id for id what one engine gives, hook and all
    gang = _hip.EngineGang(eng, 3)
    try:
        assert gang.max_batch == 3 * eng.max_batch and len({e.arena()[0] for e in gang.engines}) == 1
        assert gang.generate(prompts, samp, steps_per_poll=3, prefill_token_budget=70) == got
        seen2 = {}

        def hook2(i, slot, n_new, e):
            seen2[i] = (n_new, e.read(slot)[0])
        assert gang.generate(prompts, samp, steps_per_poll=2, prefill_token_budget=70, on_finished=hook2) == [[] for _ in prompts]
        assert seen2 == {i: (len(g), g) for i, g in enumerate(got)}
        # a request that cannot run (an empty prompt) fails the call; the engines that had already admitted theirs release them
        with pytest.raises(ValueError):
            gang.generate(prompts[:4] + [[]], samp, steps_per_poll=3, prefill_token_budget=70)
        for e in gang.engines:
            st = e.kv_stats()
            assert e.free_slots() == e.max_batch and st["free_pages"] == st["total_pages"]
        assert gang.generate(prompts, samp, prefill_token_budget=70) == got                  # (default burst length of a gang)
        # how the engines' prompt passes are placed against each other is scheduling only (one engine out of the gang at a time /
        # all at once): the ids stay, and the gate never starves an engine with nothing running
        for admit in ("spaced:2", "spaced:50", "wave", "wave:2:3"):
            assert gang.generate(prompts, samp, prefill_token_budget=70, min_admit=2, admit=admit) == got, admit
        with pytest.raises(ValueError):
            gang.generate(prompts, samp, admit="sometimes")
    finally:
        gang.close()
    assert eng.generate(prompts, samp, steps_per_poll=3, prefill_token_budget=70) == got     # engine 0 is back on its own stream


def test_parking_rows_keep_decode_slots_busy_and_the_ids(emu_lib):
    """Round 6 (ABI 9 ntts_backbone_config.park_slots + ntts_backbone_activate): an engine of 2 decode slots + 3 PARKING rows.  The scheduler
    admits prompts into decode slots first and into parking rows next (their prompt pass, KV pages and first token happen there); a parked
    request moves into a decode slot as soon as one is released.  Same ids as the engine without parking -- a request's arithmetic does not
    depend on its row, nor on the row it was prefilled in --, for the run-ahead and the blocking scheduler, with the hand-off hook, and with
    an early-EOS request that is already FINISHED when it is activated; slots, parking rows and KV pages all come back; misuse is refused."""
    cfg = br.BackboneConfig.tiny(vocab_size=512, num_layers=1)
    w = br.make_weights(cfg, 11, walk_gain=4.0)
    wd = br.cast_weights(w, torch.bfloat16)
    lens = [5, 33, 64, 17, 40, 9, 21]
    prompts = [br.synthetic_prompt(cfg, 10 + i, n) for i, n in enumerate(lens)]
    probe = br.generate(cfg, wd, prompts[3], lens[3] + 12, eos_id=cfg.vocab_size - 1, min_new_tokens=0)
    eos = probe.ids[4]
    want = [br.generate(cfg, wd, p, len(p) + 10, eos_id=eos, min_new_tokens=3).ids for p in prompts]
    samp = [_hip.Sampling(max_length=len(p) + 10, min_new_tokens=3, eos_token_id=eos, do_sample=False) for p in prompts]
    eng = make_engine(cfg, w, emu_lib, max_batch=2, max_prefill_tokens=256, park_slots=3)
    assert eng.max_batch == 2 and eng.park_slots == 3 and eng.n_rows == 5
    assert eng.generate(prompts, samp, steps_per_poll=3) == want
    assert eng.counters["prefill_calls"] <= 3              # 7 prompts through 2 decode slots in waves of up to 5, not one pass per freed slot
    assert eng.generate(prompts, samp, steps_per_poll=2, run_ahead=False, min_admit=3) == want
    seen = {}

    def hook(i, slot, n_new):
        assert slot < eng.max_batch                        # requests finish in decode slots only
        seen[i] = (n_new, eng.read(slot)[0])
    assert eng.generate(prompts, samp, steps_per_poll=3, on_finished=hook) == [[] for _ in prompts]
    assert seen == {i: (len(g), g) for i, g in enumerate(want)}
    st = eng.kv_stats()
    assert eng.free_slots() == 2 and len(eng._free_park) == 3 and st["free_pages"] == st["total_pages"]
    # a request whose FIRST token already ends it (max_length = prompt + 1) is activated in the FINISHED state and leaves through the ordinary path
    one = [_hip.Sampling(max_length=len(p) + 1, min_new_tokens=0, eos_token_id=eos, do_sample=False) for p in prompts]
    assert eng.generate(prompts, one, steps_per_poll=2) == [g[:1] for g in want]
    # by hand: prefill into a parking row, activate, decode
    p_row, slot = eng.acquire_park(), eng.acquire_slot()
    eng.prefill([prompts[1]], [p_row], [samp[1]])
    with pytest.raises(_hip.NeuTTSHipError):
        eng.activate([slot], [p_row])                      # (the wrong way round)
    eng.activate([p_row], [slot])
    with pytest.raises(_hip.NeuTTSHipError):
        eng.activate([p_row], [slot])                      # the parking row is empty now, the slot taken
    eng.decode(9)
    assert eng.read(slot)[0] == want[1]
    eng.release(slot)
    st = eng.kv_stats()
    assert eng.free_slots() == 2 and len(eng._free_park) == 3 and st["free_pages"] == st["total_pages"]
    with pytest.raises(_hip.NeuTTSHipError):
        make_engine(cfg, w, emu_lib, max_batch=2, park_slots=-1)
    eng.close()


@pytest.mark.timeout(300)
def test_generate_with_pages_held_outside_the_call(emu_lib, monkeypatch):
    """KV pages held outside a generate() call -- a suspended stream, a slot another caller prefilled -- are not
    the scheduler's to hand out.  (1) Requests that fit beside the holder one at a time still complete, with the ids they have alone.
    (2) A request the FREE pages can never hold is refused up front.  (3) With the up-front check blinded (kv_stats reporting the
    whole pool as free, the round-4 behaviour) the run-ahead scheduler meets the exhausted pool in the middle of decoding: it must
    raise and release its slots, not spin on poll_end / poll_begin."""
    cfg = br.BackboneConfig.tiny(vocab_size=256, num_layers=1)
    w = br.make_weights(cfg, 3)
    eng = make_engine(cfg, w, emu_lib, max_batch=3, max_context=128, num_pages=4, max_prefill_tokens=128)
    short = _hip.Sampling(max_length=60, min_new_tokens=40, eos_token_id=1, do_sample=False)      # 2 pages
    prompts = [br.synthetic_prompt(cfg, i, 12) for i in range(2)]
    alone = eng.generate(prompts, short, steps_per_poll=3)
    holder = eng.acquire_slot()
    eng.prefill([br.synthetic_prompt(cfg, 7, 40)], [holder], [_hip.Sampling(max_length=64, min_new_tokens=0, eos_token_id=1, do_sample=False)])   # holds 2 of the 4 pages
    assert eng.kv_stats()["free_pages"] == 2
    assert eng.generate(prompts, short, steps_per_poll=3) == alone                                 # one after the other beside the holder
    long_ = _hip.Sampling(max_length=128, min_new_tokens=100, eos_token_id=1, do_sample=False)     # 4 pages: never beside the holder
    with pytest.raises(_hip.NeuTTSHipError) as ei:
        eng.generate(prompts[:1], long_, steps_per_poll=3)
    assert ei.value.code == -3
    real = eng.kv_stats
    monkeypatch.setattr(eng, "kv_stats", lambda: dict(real(), free_pages=real()["total_pages"]))
    with pytest.raises(_hip.NeuTTSHipError) as ei:                                                 # admitted, then the pool runs dry mid-decode
        eng.generate(prompts[:1], long_, steps_per_poll=3)
    assert ei.value.code == -3
    monkeypatch.undo()
    assert eng.free_slots() == 2 and eng.kv_stats()["free_pages"] == 2                             # everything the failed calls held is back
    eng.release(holder)
    assert eng.generate(prompts, short, steps_per_poll=3) == alone


def test_wide_decode_shape_matches_the_narrow_shape(emu_lib, monkeypatch):
    """Round 6: the WIDE decode step (engines of >= 512 slots -- the static benchmark's four 256-utterance batches as ONE 1024-row chain): QKV +
    RoPE + K append on 64 / 128 batch rows per workgroup (qkv_rope.h TMQ = 2 / 4), o_proj on whole-K 64 x 64 tiles with the residual add in
    the epilogue (no fp32 slabs; the norm behind it only normalises), down_proj on the 128 x 128 / 8-wave split-K tile, gate/up on 256 x 192.
    Forced on a 3-slot engine through NTTS_WIDE (the GPU suite runs it at 1024 slots against the oracle, tests/test_gpu_parity_matrix.py):
    the ids of walk weights are those of the narrow step at every step, every logits row agrees to fp32-summation-order noise (o_proj sums
    its K in one chain instead of four slabs), ragged contexts incl. a free slot between running ones."""
    monkeypatch.setenv("NTTS_SMALL_BATCH", "0")
    z, cfg, w = load_fixture("backbone_small_walk")
    S, N, eos = 40, 5, int(z["eos"])
    prompts = [br.synthetic_prompt(cfg, 0, S), br.synthetic_prompt(cfg, 1, S - 7)]
    samp = [_hip.Sampling(max_length=len(p) + N, min_new_tokens=N, eos_token_id=eos, do_sample=False) for p in prompts]

    def run(eng):
        eng.prefill(prompts, [0, 2], samp)                       # slot 1 stays free
        rows = []
        for k in range(N):
            if k:
                eng.decode(1)
            rows.append([eng.read_logits(s).copy() for s in (0, 2)])
        ids = [eng.read(s)[0] for s in (0, 2)]
        eng.release_many([0, 2])
        return rows, ids

    eng = make_engine(cfg, w, emu_lib, max_batch=3, max_context=96, bf16_upload=True)
    eng.set_debug(True)
    rows1, ids1 = run(eng)
    eng.close()
    assert all(len(set(i)) == N for i in ids1)
    monkeypatch.setenv("NTTS_WIDE", "1")
    for qkv_rows in ("2", "4"):                                  # 64 / 128 batch rows per QKV workgroup
        monkeypatch.setenv("NTTS_WIDE_QKV", qkv_rows)
        eng = make_engine(cfg, w, emu_lib, max_batch=3, max_context=96, bf16_upload=True)
        eng.set_debug(True)
        rowsw, idsw = run(eng)
        rowsw2, idsw2 = run(eng) if qkv_rows == "2" else (rowsw, idsw)     # recycled slots, dirty pages: same bits
        eng.close()
        assert idsw == ids1 and idsw2 == ids1
        for a, b, c in zip(rows1, rowsw, rowsw2):
            for s in (0, 1):
                assert np.array_equal(b[s], c[s])
                fin = np.isfinite(a[s])
                assert np.array_equal(fin, np.isfinite(b[s])) and fin.sum() == len(fin) - 1
                assert np.abs(a[s][fin] - b[s][fin]).max() <= 2.0 ** -6 * np.abs(a[s][fin]).max()


def test_gang_decode_shape_matches_the_single_chain_shape(emu_lib, monkeypatch):
    """Round 5 (ABI 8 ntts_backbone_set_gang): the gang's decode shape -- o_proj / down_proj on the 256 x 64 tile (8 waves, all rows of a
    chain in one m-block, one K slice per XCD pair), QKV column blocks dealt to XCDs (surplus workgroups return at once), no row-block
    placement -- computes what the single-chain shape computes: the ids of walk weights are the same at every step, every logits row
    agrees to fp32-summation-order noise (the K slices are cut differently), re-capturing the step graph (set_gang) changes
    nothing bit for bit, and other split-K factors keep the ids.  (A 2-slot engine takes the shape from the NTTS_TALL / NTTS_QKV_WSTAT /
    NTTS_XCD_AFFINE overrides -- set_gang picks it from 129 slots up, and a 256-row tile costs the emulator a minute per dozen steps; the
    GPU suite runs the real shapes at batch 16 ... 512 against the oracle, tests/test_gpu_parity_matrix.py, and at full depth,
    test_air_golden_slot_in_a_full_ragged_dirty_batch.)"""
    monkeypatch.setenv("NTTS_SMALL_BATCH", "0")
    z, cfg, w = load_fixture("backbone_small_walk")
    S, N, eos = 40, 5, int(z["eos"])
    prompts = [br.synthetic_prompt(cfg, 0, S), br.synthetic_prompt(cfg, 1, S - 7)]
    samp = [_hip.Sampling(max_length=len(p) + N, min_new_tokens=N, eos_token_id=eos, do_sample=False) for p in prompts]

    def run(eng):
        eng.prefill(prompts, [0, 1], samp)
        rows = []
        for k in range(N):
            if k:
                eng.decode(1)
            rows.append([eng.read_logits(s).copy() for s in (0, 1)])
        ids = [eng.read(s)[0] for s in (0, 1)]
        eng.release_many([0, 1])
        return rows, ids

    eng = make_engine(cfg, w, emu_lib, max_batch=2, max_context=96, bf16_upload=True)
    eng.set_debug(True)
    rows1, ids1 = run(eng)
    eng.close()
    assert all(len(set(i)) == N for i in ids1)
    for k, v in {"NTTS_TALL": "3", "NTTS_QKV_WSTAT": "1", "NTTS_XCD_AFFINE": "0"}.items():
        monkeypatch.setenv(k, v)
    eng = make_engine(cfg, w, emu_lib, max_batch=2, max_context=96, bf16_upload=True)
    eng.set_debug(True)
    rows4, ids4 = run(eng)
    eng.set_gang(4)
    rows4b, ids4b = run(eng)
    eng.close()
    assert ids4 == ids1 and ids4b == ids1
    for a, b, c in zip(rows1, rows4, rows4b):
        for s in (0, 1):
            assert np.array_equal(b[s], c[s])
            fin = np.isfinite(a[s])                                              # (the masked EOS column is -inf in both)
            assert np.array_equal(fin, np.isfinite(b[s])) and fin.sum() == len(fin) - 1
            assert np.abs(a[s][fin] - b[s][fin]).max() <= 2.0 ** -6 * np.abs(a[s][fin]).max()       # a couple of bf16 ulps of the largest logit
    monkeypatch.setenv("NTTS_KS_O", "2")
    monkeypatch.setenv("NTTS_KS_D", "3")
    eng = make_engine(cfg, w, emu_lib, max_batch=2, max_context=96, bf16_upload=True)
    eng.prefill(prompts, [0, 1], samp)
    eng.decode(N - 1)
    assert [eng.read(s)[0] for s in (0, 1)] == ids1          # (other slab counts: other fp32 sums, the walk's margins hold the ids)
    eng.close()


def test_engine_error_paths(emu_lib):
    cfg = br.BackboneConfig.tiny(vocab_size=256, num_layers=1)
    w = br.make_weights(cfg, 3)
    eng = _hip.BackboneEngine(dict(vocab_size=256, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                   num_layers=1, num_heads=7, num_kv_heads=1, max_context=64, max_batch=2, num_pages=2,
                                   max_prefill_tokens=64), 0, emu_lib)
    samp = _hip.Sampling(max_length=64, min_new_tokens=0, eos_token_id=1, do_sample=False)
    with pytest.raises(_hip.NeuTTSHipError) as ei:      # weights not loaded
        eng.prefill([[1, 2, 3]], [0], [samp])
    assert ei.value.code == -4
    with pytest.raises(_hip.NeuTTSHipError):            # wrong shape
        eng.load_tensor("model.norm.weight", np.zeros(7, dtype=np.float32))
    with pytest.raises(_hip.NeuTTSHipError):            # unknown name
        eng.load_tensor("model.layers.0.nope", np.zeros(7, dtype=np.float32))
    eng.load_state_dict({k: v.numpy() for k, v in w.items()}, inv_freq=br.rope_inv_freq(cfg).numpy())
    with pytest.raises(_hip.NeuTTSHipError) as ei:      # nonsensical sampling parameters -> loud, not silent greedy
        eng.prefill([[1, 2, 3]], [0], [_hip.Sampling(max_length=64, eos_token_id=1, do_sample=True, top_k=0)])
    assert ei.value.code == -1
    with pytest.raises(_hip.NeuTTSHipError):
        eng.prefill([[1, 2, 3]], [0], [_hip.Sampling(max_length=64, eos_token_id=1, do_sample=True, temperature=0.0)])
    with pytest.raises(_hip.NeuTTSHipError):            # token id out of range
        eng.prefill([[1, 2, 999]], [0], [samp])
    with pytest.raises(_hip.NeuTTSHipError) as ei:      # 3 pages needed, pool has 2
        eng.prefill([list(range(1, 35)), list(range(1, 31))], [0, 1], [samp, samp])
    assert ei.value.code == -3
    eng.prefill([list(range(1, 30))], [0], [samp])      # pool intact after the failed call
    with pytest.raises(_hip.NeuTTSHipError) as ei:      # slot busy
        eng.prefill([[1, 2, 3]], [0], [samp])
    assert ei.value.code == -4
    eng.release(0)
    eng.prefill([[1, 2, 3]], [0], [samp])
    st, nn = eng.poll()
    assert st[0] in (1, 2) and nn[0] == 1 and st[1] == 0


def test_warm_up_and_side_stream_prefill(emu_lib):
    """Start-up hygiene and the serving knobs leave results alone: warm_up() (one throw-away request through slot 0), then the
    prompt pass routed through the side-stream code path (ntts_backbone_set_prefill_cu_mask; the emulator has one queue, the
    point here is the stream swap, its ordering calls and the restore on every exit), then the golden teacher-forced run."""
    z, cfg, w = load_fixture("backbone_tiny")
    S, N, mn, eos = int(z["s_len"]), 8, int(z["min_new"]), int(z["eos"])
    eng = make_engine(cfg, w, emu_lib, max_batch=3)
    eng.warm_up()
    eng.set_prefill_cu_mask([0xffff])
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=mn, eos_token_id=eos, do_sample=False)
    with pytest.raises(_hip.NeuTTSHipError):                     # an error inside the swapped region must restore the engine's stream
        eng.prefill([br.synthetic_prompt(cfg, 0, S)], [7], [samp])
    eng.prefill([br.synthetic_prompt(cfg, u, S) for u in (0, 1, 2)], [2, 0, 1], [samp] * 3)
    eng.set_prefill_cu_mask(None)
    ex, tie = teacher_forced_compare(eng, 2, z["bf16_ids_0"][:N], z["bf16_topv_0"], z["bf16_topi_0"])
    assert ex + tie == N and ex >= N - 2


def test_twin_engine_and_async_snapshots(emu_lib):
    """BackboneEngine.twin(): a second engine that READS THE FIRST ONE'S ARENA (ntts_backbone_share_arena, ABI 7: what bench.py's engine gangs
    run on) and one filled by a device-to-device copy of it (share=False) generate the same ids as the engine they were made from; and the asynchronous snapshot calls (poll_begin / poll_end /
    read_finished, ABI 5) report what the blocking poll / read report, refuse to be opened twice and refuse rows that were not finished."""
    z, cfg, w = load_fixture("backbone_tiny")
    S, N, mn, eos = int(z["s_len"]), 8, int(z["min_new"]), int(z["eos"])
    eng = make_engine(cfg, w, emu_lib, max_batch=2)
    tw, tw_copy = eng.twin(), eng.twin(share=False)
    assert tw.arena()[0] == eng.arena()[0] and tw_copy.arena()[0] != eng.arena()[0]
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    prompts = [br.synthetic_prompt(cfg, u, S) for u in (0, 1)]
    got = []
    for e in (eng, tw, tw_copy):
        e.prefill(prompts, [0, 1], [samp] * 2)
        e.decode(3)
        e.poll_begin()
        with pytest.raises(_hip.NeuTTSHipError):
            e.poll_begin()                                   # one snapshot at a time
        e.decode(N)                                          # enqueued behind the snapshot: must not show in it
        st, nn = e.poll_end()
        assert st.tolist() == [1, 1] and nn.tolist() == [4, 4]      # the prompt pass's token + 3 steps
        with pytest.raises(_hip.NeuTTSHipError):
            e.read_finished(0)                               # running in that snapshot
        e.poll_begin()
        st, nn = e.poll_end()
        assert st.tolist() == [2, 2] and nn.tolist() == [N, N]
        ids = [e.read_finished(s) for s in (0, 1)]
        assert ids == [e.read(s)[0] for s in (0, 1)]
        got.append(ids)
        with pytest.raises(_hip.NeuTTSHipError):
            e.release_many([0, 0])                           # repeated slot: nothing is released
        e.release_many([1, 0])                               # one stream operation for the whole set
        st, _ = e.poll()
        assert st.tolist() == [0, 0] and e.free_slots() == 2
    assert got[0] == got[1] == got[2]
    assert got[0][0] == z["bf16_ids_0"][:N].tolist() or sum(a == b for a, b in zip(got[0][0], z["bf16_ids_0"][:N].tolist())) >= N - 2


def test_share_arena_error_paths_and_side_by_side_chains(emu_lib):
    """ntts_backbone_share_arena (ABI 7): refused for an engine that already holds weights, for a donor that is not finalised and for a
    different geometry; no weight load or arena copy into an engine that reads another's arena; and two engines on one arena whose decode
    steps are enqueued ALTERNATELY (bench.py's gang schedule) produce, each, the ids of the same prompts run alone."""
    z, cfg, w = load_fixture("backbone_tiny")
    S, N, eos = int(z["s_len"]), 6, int(z["eos"])
    eng = make_engine(cfg, w, emu_lib, max_batch=2)
    lib = eng.lib
    fresh = _hip.BackboneEngine(eng.cfg, 0, emu_lib)
    assert lib.ntts_backbone_share_arena(eng.h, fresh.h) == -4            # donor not finalised (and eng holds weights)
    other = _hip.BackboneEngine(dict(eng.cfg, max_context=eng.cfg["max_context"] + 32), 0, emu_lib)
    assert lib.ntts_backbone_share_arena(other.h, eng.h) == -1            # EINVAL: another RoPE table length
    assert lib.ntts_backbone_share_arena(eng.h, eng.h) == -1
    tw = eng.twin()
    assert lib.ntts_backbone_share_arena(tw.h, eng.h) == -4               # already sharing
    with pytest.raises(_hip.NeuTTSHipError):
        tw.load_tensor("model.norm.weight", np.ones(cfg.hidden_size, dtype=np.float32))
    buf = np.zeros(tw.arena()[1], dtype=np.uint8)
    with pytest.raises(_hip.NeuTTSHipError):
        tw.arena_copy(buf.ctypes.data, buf.nbytes, True)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    pa = [br.synthetic_prompt(cfg, u, S) for u in (0, 1)]
    pb = [br.synthetic_prompt(cfg, u, S) for u in (2, 3)]
    eng.prefill(pa, [0, 1], [samp] * 2)
    eng.decode(N - 1)
    alone_a = [eng.read(s)[0] for s in (0, 1)]
    eng.release_many([0, 1])
    eng.prefill(pb, [0, 1], [samp] * 2)
    eng.decode(N - 1)
    alone_b = [eng.read(s)[0] for s in (0, 1)]
    eng.release_many([0, 1])
    eng.prefill(pa, [0, 1], [samp] * 2)
    tw.prefill(pb, [0, 1], [samp] * 2)
    for _ in range(N - 1):
        eng.decode(1)
        tw.decode(1)
    assert [eng.read(s)[0] for s in (0, 1)] == alone_a and [tw.read(s)[0] for s in (0, 1)] == alone_b
    tw.close()                                                            # the donor's arena survives its reader
    eng.release_many([0, 1])
    eng.prefill(pa, [0, 1], [samp] * 2)
    eng.decode(N - 1)
    assert [eng.read(s)[0] for s in (0, 1)] == alone_a
    # the arena is reference-counted -- a DONOR destroyed before its readers leaves them on live weights
    tw2, tw3 = eng.twin(), eng.twin()
    tw2._donor = tw3._donor = None
    eng.close()
    for t, p, want in ((tw2, pb, alone_b), (tw3, pa, alone_a)):
        t.prefill(p, [0, 1], [samp] * 2)
        t.decode(N - 1)
        assert [t.read(s)[0] for s in (0, 1)] == want
        t.close()
    # an engine gang: 1..4 engines, a context manager, engine 0 back to its single-chain shape afterwards; more than four are refused
    eng = make_engine(cfg, w, emu_lib, max_batch=2)
    with pytest.raises(ValueError):
        _hip.EngineGang(eng, 5)
    with _hip.EngineGang(eng, 2) as gang:
        assert len(gang.engines) == 2 and len(gang._streams) == 2          # (a lane per engine; the emulator's streams are null handles)
        assert gang.generate(pa + pb, samp) == alone_a + alone_b
    assert len(gang.engines) == 1 and gang.lane(0) is None
    assert eng.generate(pa, samp) == alone_a


def test_read_finished_refuses_a_stale_snapshot(emu_lib):
    """a snapshot entry speaks for the request that occupied the slot when poll_begin was enqueued.  Both sequences
    an external scheduler could run -- (a) poll_begin -> release -> prefill -> poll_end -> read_finished, (b) a completed snapshot that
    showed the slot FINISHED, then release + prefill, then read_finished -- must fail with ESTATE instead of handing out the OLD count
    together with rows the new request is overwriting; a fresh snapshot of the new occupant works again."""
    z, cfg, w = load_fixture("backbone_tiny")
    S, N, eos = int(z["s_len"]), 4, int(z["eos"])
    eng = make_engine(cfg, w, emu_lib, max_batch=2)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    p0, p1 = br.synthetic_prompt(cfg, 0, S), br.synthetic_prompt(cfg, 1, S)
    eng.prefill([p0], [0], [samp])
    eng.decode(N)                                            # slot 0 is finished on the device
    first = eng.read(0)[0]
    # (a) the slot changes hands while the snapshot is open
    eng.poll_begin()
    eng.release(0)
    eng.prefill([p1], [0], [samp])
    st, nn = eng.poll_end()
    assert st[0] == 2 and nn[0] == N                          # the snapshot itself still shows the OLD occupant, finished
    with pytest.raises(_hip.NeuTTSHipError) as ei:
        eng.read_finished(0)
    assert ei.value.code == -4
    # (b) the snapshot completed while the old occupant was there; the slot is refilled afterwards
    eng.decode(N)
    eng.poll_begin()
    st, nn = eng.poll_end()
    assert st[0] == 2
    second = eng.read_finished(0)                            # valid: same occupant as in the snapshot
    assert second == eng.read(0)[0] and second != first
    eng.release(0)
    with pytest.raises(_hip.NeuTTSHipError):
        eng.read_finished(0)                                 # released: free
    eng.prefill([p0], [0], [samp])
    with pytest.raises(_hip.NeuTTSHipError) as ei:
        eng.read_finished(0)                                 # refilled: the last completed snapshot is about somebody else
    assert ei.value.code == -4
    eng.decode(N)
    eng.poll_begin()
    eng.poll_end()
    assert eng.read_finished(0) == first                     # a fresh snapshot of the new occupant
    eng.close()
