"""Backbone parity on a real MI355X through the C-ABI: golden vectors from transformers' Qwen2 (the
dependency the reference calls, ref:neutts/neutts.py:338-347), the CPU oracle, and size-independent
properties at BASELINE.json's full size (NeuTTS-Air geometry, batch 256)."""
import numpy as np
import pytest
import torch

from oracle import backbone_ref as br
from neutts import _hip
from common import bf16_ulp, assert_free_run_matches, assert_varied, assert_walk_exact, load_fixture, make_engine, teacher_forced_compare

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _hip.load_library(hip_lib)
    return hip_lib


def test_tiny_teacher_forced(lib):
    z, cfg, w = load_fixture("backbone_tiny")
    S, N, mn, eos = int(z["s_len"]), int(z["n_new"]), int(z["min_new"]), int(z["eos"])
    eng = make_engine(cfg, w, lib, max_batch=3)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=mn, eos_token_id=eos, do_sample=False)
    eng.prefill([br.synthetic_prompt(cfg, u, S) for u in (0, 1, 2)], [2, 0, 1], [samp] * 3)
    ex, tie = teacher_forced_compare(eng, 2, z["bf16_ids_0"], z["bf16_topv_0"], z["bf16_topi_0"])
    assert ex + tie == N and ex >= N - 4


@pytest.mark.parametrize("graph", [1, 0])
def test_small_walk_exact(lib, graph, monkeypatch):
    """2 kv heads, page-boundary crossings, free-running greedy ids bit-identical to HF's; with and without
    the hipGraph replay of the decode step."""
    monkeypatch.setenv("NTTS_NO_GRAPH", "0" if graph else "1")
    z, cfg, w = load_fixture("backbone_small_walk")
    S, N, eos = int(z["s_len"]), int(z["n_new"]), int(z["eos"])
    eng = make_engine(cfg, w, lib, max_batch=2, max_context=160, bf16_upload=True)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    eng.prefill([br.synthetic_prompt(cfg, 0, S), br.synthetic_prompt(cfg, 1, S)], [0, 1], [samp, samp])
    eng.decode(N - 1)
    for u in (0, 1):
        ids, fin = eng.read(u)
        assert fin
        assert_walk_exact(ids, z[f"bf16_ids_{u}"].tolist())


@pytest.mark.parametrize("knobs", [
    {"NTTS_HEAD_TILE": "0"},                                             # 64 x 64 skinny tile (the default up to batch 64)
    {"NTTS_HEAD_TILE": "1"},                                             # 128 x 128
    {"NTTS_HEAD_TILE": "2"},                                             # 256 x 256, 16 waves
    {"NTTS_HEAD_TILE": "4"}])                                            # natural-order 256 x 288 tile, 12 waves (the batch-256 default)
def test_small_walk_exact_tile_variants(lib, knobs, monkeypatch):
    """Every lm_head tile the large-batch decode path can be switched to (gemm.h: TN = 4 and the natural-order tile with its
    uneven loader split and partial last tile), forced on at batch 2: free-running greedy ids bit-identical to HF's."""
    monkeypatch.setenv("NTTS_SMALL_BATCH", "0")
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    z, cfg, w = load_fixture("backbone_small_walk")
    S, N, eos = int(z["s_len"]), int(z["n_new"]), int(z["eos"])
    eng = make_engine(cfg, w, lib, max_batch=2, max_context=160, bf16_upload=True)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    eng.prefill([br.synthetic_prompt(cfg, 0, S), br.synthetic_prompt(cfg, 1, S)], [0, 1], [samp, samp])
    eng.decode(N - 1)
    for u in (0, 1):
        ids, fin = eng.read(u)
        assert fin, knobs
        assert_walk_exact(ids, z[f"bf16_ids_{u}"].tolist())
    eng.close()


@pytest.mark.parametrize("max_batch", [16, 40, 64, 128, 256, 512, 640])
def test_xcd_row_block_placement(lib, max_batch, monkeypatch):
    """NTTS_XCD_AFFINE=7 at every batch size it applies to (8 / 4 / 2 / 1 XCDs per 64-row m-block): the split-K GEMMs, the norms
    behind them and decode attention place an m-block's rows on one group of XCDs -- a permutation of which workgroup does what.
    Sequences in scattered slots (first, middle and last m-block) give the oracle's free-running ids.  Batch 16 / 40 (no placement
    there) and 64 also run the tile path's attention with the output dimensions split over four / two / two workgroups per
    (sequence, kv-head) (attn_decode.h DS)."""
    monkeypatch.setenv("NTTS_XCD_AFFINE", "7")
    monkeypatch.setenv("NTTS_SMALL_BATCH", "0")
    cfg = br.BackboneConfig(vocab_size=3000, hidden_size=896, intermediate_size=1216, num_layers=2)
    w = br.make_weights(cfg, 29, walk_gain=4.0)
    wd = br.cast_weights(w, torch.bfloat16)
    slots = [3, max_batch // 2 + 5, max_batch - 1]
    prompts = [br.synthetic_prompt(cfg, 70 + i, 30 + 7 * i) for i in range(3)]
    eos = cfg.vocab_size - 1
    N = 12
    eng = make_engine(cfg, w, lib, max_batch=max_batch, max_context=128, max_prefill_tokens=256, bf16_upload=True)
    samp = [_hip.Sampling(max_length=len(p) + N, min_new_tokens=N, eos_token_id=eos, do_sample=False) for p in prompts]
    eng.prefill(prompts, slots, samp)
    eng.decode(N - 1)
    for sl, p in zip(slots, prompts):
        ids, fin = eng.read(sl)
        assert fin and len(ids) == N
        assert_walk_exact(ids, br.generate(cfg, wd, p, len(p) + N, eos_id=eos, min_new_tokens=N).ids)
    eng.close()


def test_prefill_on_cu_masked_side_stream(lib):
    """ntts_backbone_set_prefill_cu_mask: the prompt pass on a side stream restricted to 64 of the 256 CUs (ordered before and
    behind the engine's own stream) and the decode steps that follow give HF's ids; two prefills in a row, mask removed again."""
    z, cfg, w = load_fixture("backbone_small_walk")
    S, N, eos = int(z["s_len"]), int(z["n_new"]), int(z["eos"])
    eng = make_engine(cfg, w, lib, max_batch=2, max_context=160, bf16_upload=True)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    for mask in ([0xffffffff, 0xffffffff], None, [0x0000ffff] * 8):
        eng.set_prefill_cu_mask(mask)
        eng.prefill([br.synthetic_prompt(cfg, 0, S)], [0], [samp])
        eng.prefill([br.synthetic_prompt(cfg, 1, S)], [1], [samp])
        eng.decode(N - 1)
        for u in (0, 1):
            ids, fin = eng.read(u)
            assert fin, mask
            assert_walk_exact(ids, z[f"bf16_ids_{u}"].tolist())
            eng.release(u)


@pytest.mark.parametrize("max_batch", [2, 16])     # small-batch path (o_proj prologue sums the chunk slabs) / tile path (combine pass)
def test_long_context_switches_to_split_attention(lib, monkeypatch, max_batch):
    """Small-batch path, a context that grows past attn_split_ctx (896) mid-generation: the steps before run the single-workgroup
    attention, the steps after the context-split one (second hipGraph, o_proj prologue summing the chunk slabs).  Free-running
    greedy ids on peaked weights: identical to the oracle's, and identical with the split disabled."""
    cfg = br.BackboneConfig(vocab_size=3000, hidden_size=896, intermediate_size=1216, num_layers=3)
    w = br.make_weights(cfg, 33, walk_gain=4.0)
    wd = br.cast_weights(w, torch.bfloat16)
    S, N = 872, 48                                              # positions 872 .. 919: the switch is at 896
    prompt = br.synthetic_prompt(cfg, 5, S)
    eos = cfg.vocab_size - 1
    want = br.generate(cfg, wd, prompt, S + N, eos_id=eos, min_new_tokens=N, keep_logits=True)
    got = {}
    for split in ("8", "0"):
        monkeypatch.setenv("NTTS_ATTN_SPLIT", split)
        eng = make_engine(cfg, w, lib, max_batch=max_batch, max_context=1024, max_prefill_tokens=1024, bf16_upload=True)
        samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
        eng.prefill([prompt], [1], [samp])
        eng.decode(10)                                          # two calls: the second one starts below and ends above the switch
        eng.decode(N - 1 - 10)
        ids, fin = eng.read(1)
        assert fin and len(ids) == N
        got[split] = ids
        eng.close()
    assert_walk_exact(got["8"], want.ids)
    assert got["8"] == got["0"], (got["8"], got["0"])


_TIER_ORACLE = {}


@pytest.mark.parametrize("caps", [("512", "1024"), ("128", "640"), ("512", "512"), ("0", "1024"), ("0", "0")])
def test_prompt_pass_attention_split_by_position(lib, monkeypatch, caps):
    """Prompt-pass attention runs on three kernels, chosen by the query's POSITION alone (attn_prefill.h): below NTTS_PF_RES_CAP (512) the
    resident kernel (pages in LDS, one exp per score, blocks dealt out from both ends of the prompt), below NTTS_PF_DEEP_CAP (1024) the deep one
    (K resident, V^T through a ring, scores packed), from there on the two-sweep kernel.  At NeuTTS-Air's width: a 1100-, a 700-, a 300- and a
    21-token prompt with the default cuts (the longest prompt uses all three kernels), with cuts at 128 / 640, without the deep tier, with
    everything below 1024 on the deep kernel and with everything on the two-sweep kernel -- each row equals the oracle's run of that prompt
    alone, id for id (walk weights)."""
    monkeypatch.setenv("NTTS_PF_RES_CAP", caps[0])
    monkeypatch.setenv("NTTS_PF_DEEP_CAP", caps[1])
    cfg = br.BackboneConfig(vocab_size=3000, hidden_size=896, intermediate_size=1216, num_layers=3)
    w = br.make_weights(cfg, 34, walk_gain=4.0)
    N, eos = 12, cfg.vocab_size - 1
    prompts = [br.synthetic_prompt(cfg, 1, 700), br.synthetic_prompt(cfg, 2, 300), br.synthetic_prompt(cfg, 3, 21), br.synthetic_prompt(cfg, 4, 1100)]
    if "want" not in _TIER_ORACLE:      # the oracle's runs do not depend on the tier setting: once for the five of them
        wd = br.cast_weights(w, torch.bfloat16)
        _TIER_ORACLE["want"] = [br.generate(cfg, wd, p, len(p) + N, eos_id=eos, min_new_tokens=N, keep_logits=True) for p in prompts]
    want = _TIER_ORACLE["want"]
    eng = make_engine(cfg, w, lib, max_batch=4, max_context=1152, max_prefill_tokens=2200, bf16_upload=True)
    samp = [_hip.Sampling(max_length=len(p) + N, min_new_tokens=N, eos_token_id=eos, do_sample=False) for p in prompts]
    eng.prefill(prompts, [2, 0, 3, 1], samp)
    eng.decode(N - 1)
    for s, u in ((2, 0), (0, 1), (3, 2), (1, 3)):
        ids, fin = eng.read(s)
        assert fin and len(ids) == N
        assert_walk_exact(ids, want[u].ids)
    eng.close()


def test_prompt_pass_attention_tiers_logits(lib, monkeypatch):
    """The id-for-id tier tests above run on walk weights, whose next id barely depends on attention (that is their point).  This one is the
    sensitive one: RANDOM-init weights (flat logits, every layer's attention matters), 4 layers at NeuTTS-Air's width, prompts of 1100 / 700 /
    300 tokens, the first-token logits row of each against the oracle's bf16 run -- with the default tiers (the long prompts run through all
    three kernels), with everything below 1024 on the deep kernel, with cuts at 128 / 640, and on the two-sweep kernel alone (the form the HF
    goldens pin).  Every setting must sit as close to the oracle as the two-sweep kernel does, and the settings must agree with each other to
    within the fp32-summation-order freedom (a few bf16 ulps at the top of the distribution)."""
    cfg = br.BackboneConfig(vocab_size=3000, hidden_size=896, intermediate_size=1216, num_layers=4)
    w = br.make_weights(cfg, 36)
    wd = br.cast_weights(w, torch.bfloat16)
    eos = cfg.vocab_size - 1
    prompts = [br.synthetic_prompt(cfg, 1, 1100), br.synthetic_prompt(cfg, 2, 700), br.synthetic_prompt(cfg, 3, 300)]
    ref = [br.generate(cfg, wd, p, len(p) + 1, eos, min_new_tokens=1, keep_logits=True).logits[0].double() for p in prompts]
    rows = {}
    for caps in (("0", "0"), ("512", "1024"), ("0", "1024"), ("128", "640")):
        monkeypatch.setenv("NTTS_PF_RES_CAP", caps[0])
        monkeypatch.setenv("NTTS_PF_DEEP_CAP", caps[1])
        eng = make_engine(cfg, w, lib, max_batch=4, max_context=1152, max_prefill_tokens=2200, bf16_upload=True)
        eng.set_debug(True)
        samp = [_hip.Sampling(max_length=len(p) + 2, min_new_tokens=2, eos_token_id=eos, do_sample=False) for p in prompts]
        eng.prefill(prompts, [0, 1, 2], samp)
        rows[caps] = [torch.from_numpy(eng.read_logits(s)).double() for s in range(3)]
        eng.close()

    def dist(a, b):     # (relative RMS over the row, largest difference at b's top-8 entries in bf16 ulps of those entries)
        fin = torch.isfinite(a) & torch.isfinite(b)                  # (the EOS column is masked to -inf by min_new_tokens)
        a, b = a[fin], b[fin]
        top = torch.topk(b, 8).indices
        ulps = max(abs(float(a[i] - b[i])) / bf16_ulp(float(b[i])) for i in top)
        return float((a - b).norm() / b.norm()), ulps
    base = rows[("0", "0")]
    for u in range(3):
        r0, u0 = dist(base[u], ref[u])
        print(f"prompt {len(prompts[u])}: two-sweep vs oracle rel RMS {r0:.2e}, top-8 within {u0:.1f} ulps")
        for caps, rr in rows.items():
            r1, u1 = dist(rr[u], ref[u])
            r2, u2 = dist(rr[u], base[u])
            print(f"   tiers {caps}: vs oracle {r1:.2e} / {u1:.1f} ulps; vs two-sweep {r2:.2e} / {u2:.1f} ulps")
            assert r1 <= 1.5 * r0 + 1e-3 and u1 <= max(2.0 * u0, 4.0), (caps, u, r1, u1, r0, u0)
            assert r2 <= 2.0 * r0 + 1e-3 and u2 <= max(2.0 * u0, 4.0), (caps, u, r2, u2)
        assert torch.equal(rows[("512", "1024")][u], rows[("0", "1024")][u])    # the resident and the deep kernel are the same arithmetic


@pytest.mark.parametrize("n_prompts", [24, 40, 70])
def test_prompt_pass_attention_heads_spread_over_workgroups(lib, n_prompts):
    """Short prompt passes spread a kv-group's query heads over workgroups while the grid stays within the CUs (attn_prefill_res_launch:
    1 / 2 / 4 / all 7 heads per workgroup).  24 prompts of 200 tokens take 2 heads per workgroup, 40 take 4, 70 all 7; three prompts alone take
    1.  Per query the arithmetic must not depend on how the heads are dealt out: on RANDOM-init weights (NeuTTS-Air's width and head counts)
    the first-token logits rows of three prompts are BIT-identical whether they ran in the big pass or alone (where the GEMMs also take other
    tiles: every tile adds an output element's k-tiles in the same order)."""
    cfg = br.BackboneConfig(vocab_size=3000, hidden_size=896, intermediate_size=1216, num_layers=2)
    w = br.make_weights(cfg, 35)
    eos = cfg.vocab_size - 1
    lens = [200 - 3 * (i % 5) for i in range(n_prompts)]
    prompts = [br.synthetic_prompt(cfg, 10 + i, lens[i]) for i in range(n_prompts)]
    check = [0, n_prompts // 2, n_prompts - 1]
    samp = [_hip.Sampling(max_length=lens[i] + 2, min_new_tokens=2, eos_token_id=eos, do_sample=False) for i in range(n_prompts)]
    eng = make_engine(cfg, w, lib, max_batch=n_prompts, max_context=256, max_prefill_tokens=200 * n_prompts, bf16_upload=True)
    eng.set_debug(True)
    eng.prefill(prompts, list(range(n_prompts)), samp)
    big = [eng.read_logits(i).copy() for i in check]
    eng.close()
    eng = make_engine(cfg, w, lib, max_batch=n_prompts, max_context=256, max_prefill_tokens=200 * n_prompts, bf16_upload=True)
    eng.set_debug(True)
    eng.prefill([prompts[i] for i in check], check, [samp[i] for i in check])
    for k, i in enumerate(check):
        alone = eng.read_logits(i)
        assert np.array_equal(big[k], alone), (n_prompts, i, float(np.nanmax(np.abs(big[k] - alone))))
    eng.close()


@pytest.mark.parametrize("max_batch", [2, 16])
def test_short_sequence_beside_a_long_one_across_the_split_switch(lib, max_batch):
    """the choice between the single-workgroup attention and the context-split one follows the LONGEST running context
    of the batch, so a short sequence (one page: fewer pages than chunks, most of its chunks empty) runs the split kernels
    whenever a long one shares its batch.  The two forms add the softmax denominator and P V in different fp32 orders, i.e. a
    sequence's ids may depend on its batch mates within the summation-order freedom the parity bars allow -- what must hold is
    that EACH row still matches the oracle's run of that sequence alone (identical, or up to a tie of the oracle's own top-2)."""
    cfg = br.BackboneConfig(vocab_size=3000, hidden_size=896, intermediate_size=1216, num_layers=3)
    w = br.make_weights(cfg, 33, walk_gain=4.0)
    wd = br.cast_weights(w, torch.bfloat16)
    N, eos = 24, cfg.vocab_size - 1
    prompts = [br.synthetic_prompt(cfg, 6, 10), br.synthetic_prompt(cfg, 5, 900)]       # contexts 10.. and 900.. (switch at 896)
    want = [br.generate(cfg, wd, p, len(p) + N, eos_id=eos, min_new_tokens=N, keep_logits=True) for p in prompts]
    eng = make_engine(cfg, w, lib, max_batch=max_batch, max_context=1024, max_prefill_tokens=1024, bf16_upload=True)
    samp = [_hip.Sampling(max_length=len(p) + N, min_new_tokens=N, eos_token_id=eos, do_sample=False) for p in prompts]
    eng.prefill(prompts, [0, 1], samp)
    eng.decode(N - 1)
    for s in (0, 1):
        ids, fin = eng.read(s)
        assert fin and len(ids) == N
        assert_walk_exact(ids, want[s].ids)
    eng.close()


def test_continuous_batching_ragged_vs_oracle(lib):
    cfg = br.BackboneConfig(vocab_size=3000, hidden_size=896, intermediate_size=1216, num_layers=3)
    w = br.make_weights(cfg, 21, walk_gain=4.0)
    wd = br.cast_weights(w, torch.bfloat16)
    lens = [5, 33, 64, 17, 40, 100, 1, 65, 31, 32]
    prompts = [br.synthetic_prompt(cfg, 10 + i, n) for i, n in enumerate(lens)]
    # Walk weights emit a different id every step, so a request stops early only through ITS OWN eos id: request i gets the id its
    # free run emits at step 4 .. 12 (odd i; min_new_tokens = 3 masks it before that) or an id it never emits (even i: runs to
    # max_length, 10 .. 28 new tokens) -- ragged stops through both criteria, EOS masking exercised, nothing left to a tie
    free = [br.generate(cfg, wd, p, len(p) + 30, eos_id=cfg.vocab_size - 1, min_new_tokens=30).ids for p in prompts]
    eos = [free[i][4 + (i % 5) * 2] if i % 2 else cfg.vocab_size - 1 for i in range(len(prompts))]
    mlen = [len(p) + 10 + 2 * i for i, p in enumerate(prompts)]
    want = [br.generate(cfg, wd, p, m, eos_id=e, min_new_tokens=3, keep_logits=True) for p, m, e in zip(prompts, mlen, eos)]
    assert [len(r.ids) for r in want] == [10, 7, 14, 11, 18, 5, 22, 9, 26, 13], [len(r.ids) for r in want]
    eng = make_engine(cfg, w, lib, max_batch=4, max_context=256, max_prefill_tokens=256)
    samp = [_hip.Sampling(max_length=m, min_new_tokens=3, eos_token_id=e, do_sample=False) for m, e in zip(mlen, eos)]
    got = eng.generate(prompts, samp, steps_per_poll=5, prefill_token_budget=150)
    for g, ref in zip(got, want):
        assert g == ref.ids, (g, ref.ids)
        assert_varied(g)
    # run-ahead scheduling (the default: one burst queued ahead of the host's bookkeeping, asynchronous snapshots, finished ids read
    # on the copy stream past the queued decode steps) against the blocking poll, and a device-side hand-off from the on_finished hook
    assert eng.generate(prompts, samp, steps_per_poll=5, prefill_token_budget=150, run_ahead=False) == got
    assert eng.generate(prompts, samp, steps_per_poll=2, prefill_token_budget=150) == got
    codes = torch.zeros((len(prompts), 32), dtype=torch.int32, device="cuda")
    lens_out = torch.zeros(len(prompts), dtype=torch.int32, device="cuda")
    seen = {}

    def hook(i, slot, n_new):
        seen[i] = n_new
        eng.export_codes([slot], 0, cfg.vocab_size, codes[i:i + 1].data_ptr(), 32, lens_out[i:i + 1].data_ptr())
    assert eng.generate(prompts, samp, steps_per_poll=5, prefill_token_budget=150, on_finished=hook) == [[] for _ in prompts]
    eng.sync()
    assert seen == {i: len(g) for i, g in enumerate(got)}
    ch, lh = codes.cpu().numpy(), lens_out.cpu().numpy()
    for i, g in enumerate(got):
        assert lh[i] == len(g) and ch[i, :len(g)].tolist() == g
    # the same requests dealt out over a GANG of three engines on one arena (EngineGang: schedulers advanced in turn, decode chains
    # side by side on lane streams): id for id what one engine gives, hook and all
    gang = _hip.EngineGang(eng, 3)
    try:
        assert gang.max_batch == 3 * eng.max_batch and len({e.arena()[0] for e in gang.engines}) == 1
        assert gang.generate(prompts, samp, steps_per_poll=5, prefill_token_budget=150) == got
        seen2 = {}

        def hook2(i, slot, n_new, e):
            seen2[i] = (n_new, e.read(slot)[0])
        assert gang.generate(prompts, samp, steps_per_poll=2, prefill_token_budget=150, on_finished=hook2) == [[] for _ in prompts]
        assert seen2 == {i: (len(g), g) for i, g in enumerate(got)}
    finally:
        gang.close()
    assert eng.generate(prompts, samp, steps_per_poll=5, prefill_token_budget=150) == got     # engine 0 is back on its own stream


@pytest.fixture(scope="module")
def air(lib):
    z, cfg, w = load_fixture("backbone_air")
    eng = make_engine(cfg, w, lib, max_batch=256, max_context=1024, max_prefill_tokens=8192)
    return z, cfg, eng


def test_air_teacher_forced_vs_hf_golden(air):
    """NeuTTS-Air geometry, 500-token prompt, 250 greedy tokens, teacher-forced against transformers' bf16 run.
    (1) Fidelity of the logits themselves (debug tap): at the golden top-4 token ids of every step our logit is within
        a few bf16 ulps of HF's -- both sides round to bf16 ~170 times per token, in different fp32 summation orders.
    (2) Ids: every token equals HF's unless HF's own top-2 logits are within 4 bf16 ulps of each other (then ours must
        be one of HF's top-4 inside that band); random-init weights give flat logits, so such near-ties are frequent."""
    z, cfg, eng = air
    S, N, eos = int(z["s_len"]), int(z["n_new"]), int(z["eos"])
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    eng.set_debug(True)
    stats = []
    try:
        eng.prefill([br.synthetic_prompt(cfg, 0, S)], [7], [samp])
        ex, tie = teacher_forced_compare(eng, 7, z["bf16_ids_0"], z["bf16_topv_0"], z["bf16_topi_0"], max_ulps=4.0,
                                         logit_stats=stats)
    finally:
        eng.release(7)
        eng.set_debug(False)
    err = np.array(stats)
    print(f"logit error at golden top-4 ids, bf16 ulps: mean {err.mean():.3f}  p99 {np.percentile(err, 99):.2f}  max {err.max():.2f}; "
          f"{ex} exact + {tie} near-tie of {N}")
    # Both sides' logits are bf16 values, so the error at a golden value is a whole number of that value's bf16 ulps:
    # measured on MI355X (profiles/r01g_pytest_gpu.log): mean 0.56, p99 2, max 3 -- "<= 3.5" means "at most 3 ulps".
    assert err.mean() <= 0.8 and np.percentile(err, 99) <= 2.5 and err.max() <= 3.5, \
        (err.mean(), np.percentile(err, 99), err.max())
    assert ex + tie == N
    # The steps where HF's OWN top-2 logits lie within 4 bf16 ulps are a property of the fixture (frozen with it); only
    # there may our token differ (teacher_forced_compare asserts the band per step).  Measured: 239 exact + 11 near-ties.
    tv = z["bf16_topv_0"]
    fixture_near = [k for k in range(N) if tv[k][0] - tv[k][1] <= 4.0 * 2.0 ** (np.floor(np.log2(abs(tv[k][0]))) - 7)]
    print(f"fixture near-tie steps (golden top-2 gap <= 4 ulps): {len(fixture_near)} of {N}: {fixture_near}")
    assert tie <= len(fixture_near)
    assert ex >= 235, f"only {ex}/{N} exact ({tie} near-ties); measured 239"


def test_air_logits_vs_fp32_reference(air):
    """The reference as shipped loads fp32 weights (ref:neutts/neutts.py:164 calls from_pretrained without a dtype); this
    engine's contract is bf16 (BASELINE.json).  How far apart are they where it matters for sampling?  First-token
    distribution after the 500-token prompt, engine (bf16) vs the oracle run in fp32: overlap of the top-50 sets (the
    reference's top_k), KL divergence of the softmax, and the same for the oracle's own bf16 run as the yardstick."""
    z, cfg, eng = air
    _, _, w = load_fixture("backbone_air")
    S, eos = int(z["s_len"]), int(z["eos"])
    prompt = br.synthetic_prompt(cfg, 0, S)
    for s in range(256):
        eng.release(s)
    eng.set_debug(True)
    try:
        eng.prefill([prompt], [3], [_hip.Sampling(max_length=S + 2, min_new_tokens=2, eos_token_id=eos, do_sample=False)])
        ours = torch.from_numpy(eng.read_logits(3)).double()
    finally:
        eng.release(3)
        eng.set_debug(False)
    ref32 = br.generate(cfg, w, prompt, S + 1, eos, min_new_tokens=1, keep_logits=True).logits[0].double()
    ref16 = br.generate(cfg, br.cast_weights(w, torch.bfloat16), prompt, S + 1, eos, min_new_tokens=1, keep_logits=True).logits[0].double()
    fin = torch.isfinite(ref32)

    def stats(a):
        p, q = torch.softmax(ref32[fin], 0), torch.softmax(a[fin], 0)
        kl = float((p * (p.log() - q.log())).sum())
        top = len(set(torch.topk(ref32[fin], 50).indices.tolist()) & set(torch.topk(a[fin], 50).indices.tolist()))
        return kl, top, float((a[fin] - ref32[fin]).abs().max())
    kl_o, top_o, d_o = stats(ours)
    kl_h, top_h, d_h = stats(ref16)
    print(f"vs the fp32 run: engine bf16  KL {kl_o:.2e}  top-50 overlap {top_o}/50  max |dlogit| {d_o:.3f};  "
          f"oracle bf16 (= HF bf16)  KL {kl_h:.2e}  top-50 overlap {top_h}/50  max |dlogit| {d_h:.3f}")
    # the engine is as close to fp32 as HF's own bf16 model is (it IS that model up to fp32 summation order)
    assert kl_o <= 2.0 * kl_h + 1e-6 and top_o >= top_h - 3 and kl_o <= 0.05


def test_air_batch1_vs_hf_golden(lib):
    """BASELINE.json configs[1]: NeuTTS-Air bf16, batch 1, 500 prefill / 250 decode, greedy.  A single-slot engine takes
    the small-batch kernel path (wave-per-16-features GEMMs, context-split attention); same golden run, same bars as
    the batch-256 engine: logits within 3 bf16 ulps of HF's at its top-4 ids, ids equal except at HF's own near-ties,
    and the free-running ids follow HF's up to the first such near-tie."""
    z, cfg, w = load_fixture("backbone_air")
    S, N, eos = int(z["s_len"]), int(z["n_new"]), int(z["eos"])
    eng = make_engine(cfg, w, lib, max_batch=1, max_context=1024, max_prefill_tokens=1024)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    prompt = br.synthetic_prompt(cfg, 0, S)
    eng.set_debug(True)
    stats = []
    try:
        eng.prefill([prompt], [0], [samp])
        ex, tie = teacher_forced_compare(eng, 0, z["bf16_ids_0"], z["bf16_topv_0"], z["bf16_topi_0"], max_ulps=4.0,
                                         logit_stats=stats)
    finally:
        eng.release(0)
        eng.set_debug(False)
    err = np.array(stats)
    print(f"batch 1: logit error at golden top-4 ids, bf16 ulps: mean {err.mean():.3f}  p99 {np.percentile(err, 99):.2f}  "
          f"max {err.max():.2f}; {ex} exact + {tie} near-tie of {N}")
    assert err.mean() <= 0.8 and np.percentile(err, 99) <= 2.5 and err.max() <= 3.5
    assert ex + tie == N and ex >= 228       # measured 234-236 (the small-batch kernels sum in another order than the batch-256 tiles: other near-ties flip)
    # free-running (hipGraph replay, no debug tap): follows HF's ids up to the first near-tie of HF's own logits
    eng.prefill([prompt], [0], [samp])
    eng.decode(N - 1)
    ids, fin = eng.read(0)
    g, tv = z["bf16_ids_0"].tolist(), z["bf16_topv_0"]
    assert fin and len(ids) == N
    k = next((i for i in range(N) if ids[i] != g[i]), N)
    print(f"batch 1 free run: identical to HF's ids for {k} of {N} steps")
    if k < N:
        assert tv[k][0] - tv[k][1] <= 4 * 2.0 ** (np.floor(np.log2(abs(tv[k][0]))) - 7), (k, tv[k])
    eng.close()


@pytest.mark.parametrize("chains", [1, 4], ids=["single-chain-shape", "gang-shape"])
def test_air_golden_slot_in_a_full_ragged_dirty_batch(air, chains):
    """VERDICT r4 weak 1 at FULL depth (24 layers, V = 217 488): the golden utterance in slot 137 (third m-block) of a batch whose other
    255 slots hold fillers of ragged lengths (17 ... 700 tokens, page edges included), on pages a previous occupant of every slot
    left dirty: 48 teacher-forced steps against transformers' golden run with the bars of the lone-slot test, AND every step's logits
    row bit-identical to the same utterance decoded alone in an otherwise empty engine (a row's arithmetic may not depend on its
    neighbours, their lengths or what the pages held before).  Once on the single-chain decode shape and once on the gang's
    (ntts_backbone_set_gang(4): 256-row o_proj / down_proj tiles, QKV column blocks per XCD, no row-block placement)."""
    z, cfg, eng = air
    eng.set_gang(chains)
    S, eos, N = int(z["s_len"]), int(z["eos"]), 48
    gold_p = br.synthetic_prompt(cfg, 0, S)
    samp = lambda p, n: _hip.Sampling(max_length=len(p) + n, min_new_tokens=n, eos_token_id=eos, do_sample=False)
    for s in range(256):
        eng.release(s)
    eng.set_debug(True)
    try:
        # (1) alone: the rows to compare with
        eng.prefill([gold_p], [137], [samp(gold_p, N)])
        alone = []
        for k in range(N):
            if k:
                eng.decode(1)
            alone.append(eng.read_logits(137).copy())
            ids, _ = eng.read(137)
            if ids[-1] != int(z["bf16_ids_0"][k]) and k + 1 < N:
                eng.debug_force(137, int(z["bf16_ids_0"][k]))
        eng.release(137)
        # (2) a previous occupant of every slot (other contents, decoded past the positions the run below writes), then the full ragged batch
        lens = [17, 31, 32, 33, 63, 64, 65, 95, 96, 97, 200, 333, 500, 640, 700, 129]
        slots = [s for s in range(256) if s != 137]

        def fill(seed0, extra):
            ps = [br.synthetic_prompt(cfg, seed0 + s, lens[s % len(lens)]) for s in slots]
            i = 0
            while i < len(slots):
                j, used = i, 0
                while j < len(slots) and (j == i or used + len(ps[j]) <= 8000):
                    used += len(ps[j])
                    j += 1
                eng.prefill(ps[i:j], slots[i:j], [samp(p, N + extra) for p in ps[i:j]])
                i = j
        fill(50_000, 4)
        junk = br.synthetic_prompt(cfg, 777, S)
        eng.prefill([junk], [137], [samp(junk, N + 4)])
        eng.decode(N + 2)
        eng.release_many(list(range(256)))
        fill(60_000, 0)
        eng.prefill([gold_p], [137], [samp(gold_p, N)])
        stats = []
        ex, tie = teacher_forced_compare_rows(eng, 137, z["bf16_ids_0"][:N], z["bf16_topv_0"], z["bf16_topi_0"], alone, stats)
    finally:
        eng.release_many(list(range(256)))
        eng.set_debug(False)
        eng.set_gang(1)
    err = np.array(stats)
    print(f"golden slot in a full ragged dirty batch ({chains} chain shape): {ex} exact + {tie} near-tie of {N}; logits error mean {err.mean():.3f} max {err.max():.2f} bf16 ulps; rows bit-identical to the lone run")
    assert ex + tie == N and err.mean() <= 0.8 and err.max() <= 3.5


def teacher_forced_compare_rows(eng, slot, gold_ids, gold_topv, gold_topi, rows_alone, stats):
    """teacher_forced_compare (max_ulps 4) that also demands every step's logits row to equal `rows_alone[k]` bit for bit."""
    n_exact = n_tie = 0
    n = len(gold_ids)
    for k in range(n):
        if k > 0:
            eng.decode(1)
        ids, _ = eng.read(slot)
        assert len(ids) == k + 1
        row = eng.read_logits(slot)
        assert np.array_equal(row, rows_alone[k]), f"step {k}: the row differs from the lone-slot run in {int((row != rows_alone[k]).sum())} logits"
        for i, v in zip(gold_topi[k], gold_topv[k]):
            if np.isfinite(v):
                stats.append(abs(float(row[int(i)]) - float(v)) / bf16_ulp(float(v)))
        if ids[-1] == int(gold_ids[k]):
            n_exact += 1
        else:
            band = 4.0 * bf16_ulp(gold_topv[k][0])
            cand = {int(i): float(v) for i, v in zip(gold_topi[k], gold_topv[k])}
            assert ids[-1] in cand and gold_topv[k][0] - cand[ids[-1]] <= band, (k, ids[-1], cand)
            n_tie += 1
            if k + 1 < n:
                eng.debug_force(slot, int(gold_ids[k]))
    return n_exact, n_tie


def test_air_batch256_invariance_and_golden_prefix(air):
    """Full BASELINE batch: 256 slots filled with 4 distinct golden prompts.  Rows holding the same prompt
    must produce IDENTICAL ids (batch/slot invariance -- each row's result may not depend on its
    neighbours or its pages), and each row follows HF's golden ids up to its first near-tie."""
    z, cfg, eng = air
    S, N, eos = int(z["s_len"]), 64, int(z["eos"])
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    prompts = [br.synthetic_prompt(cfg, u, S) for u in (0, 1, 2, 3)]
    for s in range(256):            # whatever an earlier (failed) test left behind
        eng.release(s)
    for c in range(0, 256, 16):     # prefill in chunks of 16 prompts (8000 tokens)
        eng.prefill([prompts[(c + i) % 4] for i in range(16)], list(range(c, c + 16)), [samp] * 16)
    eng.decode(N - 1)
    rows = [eng.read(s)[0] for s in range(256)]
    for s in range(256):
        assert len(rows[s]) == N
        assert rows[s] == rows[s % 4], f"slot {s} differs from slot {s % 4}"
    for u in range(4):
        g = z[f"bf16_ids_{u}"][:N].tolist()
        tv = z[f"bf16_topv_{u}"]
        k = next((i for i in range(N) if rows[u][i] != g[i]), N)
        if k < N:   # first divergence must be a near-tie of the golden run
            assert tv[k][0] - tv[k][1] <= 4 * 2.0 ** (np.floor(np.log2(abs(tv[k][0]))) - 7), (u, k, tv[k])
    for s in range(256):
        eng.release(s)


def test_air_walk8_free_running_exact_in_a_full_batch256_engine(lib):
    """the batch-256 tile path (fused QKV + RoPE + K append, prologue-free attention, 256 x 288 lm_head tile, XCD
    row-block placement) emitting 250 FREE-RUNNING, ALL-DIFFERENT tokens per utterance that equal transformers' bf16 run id for id --
    no tie clause.  Fixture `backbone_air_walk8` (oracle/gen_golden.py --walk): NeuTTS-Air geometry, 8 utterances of 500 prompt
    tokens, weights whose greedy decoding walks a seeded permutation of the vocabulary (synthetic._make_walk) with every golden
    top-1 / top-2 margin tens of bf16 ulps wide (asserted below from the fixture itself).  Slot s runs utterance s % 8: all four
    64-row m-blocks, every XCD group; one prompt pass of 4 x 64 prompts + 249 graph replays.  The 32 slots of an utterance must hold
    identical rows (batch / slot invariance)."""
    z, cfg, w = load_fixture("backbone_air_walk8")
    S, N, eos = int(z["s_len"]), int(z["n_new"]), int(z["eos"])
    for u in range(8):                                   # the fixture's own margins: no step of any golden run is a near-tie
        tv = z[f"bf16_topv_{u}"]
        ulps = (tv[:, 0] - tv[:, 1]) / 2.0 ** (np.floor(np.log2(np.abs(tv[:, 0]))) - 7)
        assert ulps.min() >= 16.0, (u, float(ulps.min()))
        assert_varied(z[f"bf16_ids_{u}"].tolist(), 0.95)
    eng = make_engine(cfg, w, lib, max_batch=256, max_context=768, max_prefill_tokens=64 * S, bf16_upload=True)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    prompts = [br.synthetic_prompt(cfg, u, S) for u in range(8)]
    for c in range(0, 256, 64):
        eng.prefill([prompts[s % 8] for s in range(c, c + 64)], list(range(c, c + 64)), [samp] * 64)
    eng.decode(N - 1)
    rows = [eng.read(s)[0] for s in range(256)]
    for u in range(8):
        assert_walk_exact(rows[u], z[f"bf16_ids_{u}"].tolist())
        for s in range(u, 256, 8):       # slots 8 j + u: m-blocks 0..3, every XCD group
            assert rows[s] == rows[u], (u, s)
    print(f"batch 256: 8 x {N} free-running ids equal to transformers', distinct ids per utterance: {[len(set(rows[u])) for u in range(8)]}")
    eng.close()


def test_air_walk8_three_engines_side_by_side_exact(lib):
    """bench.py's static schedule at BASELINE's shape: THREE 256-slot engines on one arena (EngineGang: ntts_backbone_share_arena, a lane
    stream per engine), their 249 step graphs replayed ALTERNATELY so that the three decode
    chains run side by side on the GPU, the third engine's prompt passes enqueued while the first two already decode.  768 free-running
    utterances x 250 ids, each id for id transformers' run of its prompt (engine e, slot s runs utterance (s + 3 e) % 8), no tie clause."""
    z, cfg, w = load_fixture("backbone_air_walk8")
    S, N, eos = int(z["s_len"]), int(z["n_new"]), int(z["eos"])
    eng = make_engine(cfg, w, lib, max_batch=256, max_context=768, max_prefill_tokens=64 * S, bf16_upload=True)
    gang = _hip.EngineGang(eng, 3)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    prompts = [br.synthetic_prompt(cfg, u, S) for u in range(8)]

    def fill(k):
        for c in range(0, 256, 64):
            gang.engines[k].prefill([prompts[(s + 3 * k) % 8] for s in range(c, c + 64)], list(range(c, c + 64)), [samp] * 64)
    try:
        fill(0)
        fill(1)
        for _ in range(40):                              # two chains side by side ...
            gang.engines[0].decode(1)
            gang.engines[1].decode(1)
        fill(2)                                          # ... the third engine's prompt passes (on its own lane) beside them ...
        for j in range(N - 1):                           # ... then three
            for k, e in enumerate(gang.engines):
                if k < 2 and j >= N - 1 - 40:
                    continue
                e.decode(1)
        for k, e in enumerate(gang.engines):
            rows = [e.read(s)[0] for s in range(256)]
            for s in range(256):
                if s < 8:
                    assert_walk_exact(rows[s], z[f"bf16_ids_{(s + 3 * k) % 8}"].tolist())
                else:
                    assert rows[s] == rows[s % 8], (k, s)
    finally:
        gang.close()
        eng.close()


@pytest.mark.parametrize("max_batch", [1, 8, 32])
def test_air_walk8_free_running_exact_small_batches(lib, max_batch):
    """The same golden runs on the other decode paths: batch 1 (BASELINE configs[1]) and batch 8 take the small-batch GEMV kernels
    (fused norm prologues, output-dimension-split attention), batch 32 the tile path with the attention split four ways over the
    output dimensions.  250 free-running ids per utterance, id for id."""
    z, cfg, w = load_fixture("backbone_air_walk8")
    S, N, eos = int(z["s_len"]), int(z["n_new"]), int(z["eos"])
    n = min(max_batch, 8)
    eng = make_engine(cfg, w, lib, max_batch=max_batch, max_context=768, max_prefill_tokens=n * S, bf16_upload=True)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    slots = [(3 * u + 1) % max_batch for u in range(n)] if max_batch > 8 else list(range(n))
    eng.prefill([br.synthetic_prompt(cfg, u, S) for u in range(n)], slots, [samp] * n)
    eng.decode(N - 1)
    for u, sl in enumerate(slots):
        ids, fin = eng.read(sl)
        assert fin
        assert_walk_exact(ids, z[f"bf16_ids_{u}"].tolist())
    eng.close()


def test_air_prefix_sharing_identical_to_plain_prefill(air):
    """Prefix sharing (ntts_backbone_prefill_shared) at NeuTTS-Air geometry: 16 utterances of one
    "speaker" -- a common 200-token beginning, then their own 300 tokens -- give bit-identical first-token logits and
    identical greedy ids whether each prompt is computed in full or 15 of them re-use the first one's KV pages (192
    tokens = 6 pages each); the shared pages outlive their donor."""
    z, cfg, eng = air
    eos = int(z["eos"])
    for s in range(256):
        eng.release(s)
    rng = np.random.default_rng(42)
    head = rng.integers(0, cfg.vocab_size - 1, 200).tolist()
    prompts = [head + rng.integers(0, cfg.vocab_size - 1, 300).tolist() for _ in range(16)]
    N = 40
    samp = [_hip.Sampling(max_length=500 + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)] * 16
    slots = list(range(16))

    def run(donors, release_donor_early):
        eng.set_debug(True)
        try:
            eng.prefill(prompts, slots, samp, donors)
            t_pf = eng.last_timing()[0]
            logits = [eng.read_logits(s).copy() for s in (0, 1, 15)]
            eng.decode(8)
            first = eng.read(0)[0]
            if release_donor_early:
                eng.release(0)
            eng.decode(N - 1 - 8)
            ids = [first] + [eng.read(s)[0] for s in slots[1:]]
        finally:
            for s in slots:
                eng.release(s)
            eng.set_debug(False)
        return logits, ids, t_pf

    st0 = eng.kv_stats()
    want_logits, want_ids, t_plain = run(None, False)
    got_logits, got_ids, t_shared = run([None] + [(0, 200)] * 15, True)
    st1 = eng.kv_stats()
    assert st1["prompt_tokens_shared"] - st0["prompt_tokens_shared"] == 15 * 192
    assert st1["free_pages"] == st1["total_pages"]
    for a, b in zip(got_logits, want_logits):
        assert np.array_equal(a, b)
    assert [g[:9] for g in got_ids[:1]] == [w[:9] for w in want_ids[:1]] and got_ids[1:] == want_ids[1:]
    print(f"prefill of 16 x 500 tokens: plain {t_plain:.2f} ms, 15 prompts sharing 192 tokens {t_shared:.2f} ms")


def test_air_sampling_topk50_full_vocab(air):
    """The reference's own call (do_sample=True, temperature=1.0, top_k=50) at NeuTTS-Air geometry, V = 217 488, 256 slots:
    each sampled token lies in the top-50 set of that step's logits (TopKLogitsWarper semantics, read back through the
    debug tap), and the first-token frequencies over 1024 independent seeds follow softmax(top-50 logits)."""
    z, cfg, eng = air
    S, eos = int(z["s_len"]), int(z["eos"])
    p = br.synthetic_prompt(cfg, 0, S)
    for s in range(256):
        eng.release(s)
    eng.set_debug(True)
    try:
        from oracle.sampling_ref import sample_topk
        counts = {}
        ref_row = None
        exact = 0
        for rep in range(4):
            for c in range(0, 256, 16):
                samp = [_hip.Sampling(max_length=S + 4, min_new_tokens=4, eos_token_id=eos, do_sample=True, top_k=50,
                                      temperature=1.0, seed=77_000 * rep + c + i) for i in range(16)]
                eng.prefill([p] * 16, list(range(c, c + 16)), samp)
            ids, _ = eng.read_all()
            row = eng.read_logits(0)
            if ref_row is None:
                ref_row = row
            assert np.array_equal(row, ref_row)                       # same prompt -> same logits, every slot / repeat
            kth = np.sort(row)[-50]
            for s in range(256):
                t = ids[s][0]
                assert row[t] >= kth, (s, t, row[t], kth)
                counts[t] = counts.get(t, 0) + 1
                # token for token against the draw's specification (oracle/sampling_ref.py): candidate-group scan over the 2 268 lm_head
                # maxima, radix select, tie handling, token order, Philox draw, inverse CDF
                want, margin = sample_topk(row, 50, 1.0, 77_000 * rep + s, 0)
                if margin > 1e-5:
                    assert t == want, (rep, s, t, want, margin)
                    exact += 1
            if rep == 0:   # a few decode steps: membership against each step's own logits
                for step in range(3):
                    eng.decode(1)
                    ids2, _ = eng.read_all()
                    for s in (0, 17, 255):
                        r = eng.read_logits(s)
                        assert r[ids2[s][-1]] >= np.sort(r)[-50], (step, s)
                        want, margin = sample_topk(r, 50, 1.0, s, len(ids2[s]) - 1)
                        assert margin <= 1e-5 or ids2[s][-1] == want, (step, s, ids2[s][-1], want, margin)
            for s in range(256):
                eng.release(s)
        top = np.where(ref_row >= np.sort(ref_row)[-50])[0]          # bf16 logits tie: everything >= the 50th value is kept
        pr = np.exp(ref_row[top] - ref_row[top].max())
        pr /= pr.sum()
        got = np.array([counts.get(int(t), 0) for t in top]) / 1024.0
        assert abs(got.sum() - 1.0) < 1e-9
        assert np.abs(got - pr).max() < 0.05, (got, pr)
        assert exact >= 1000, exact                                  # (of 1024 draws; the rest sat within 1e-5 of a boundary)
    finally:
        eng.set_debug(False)


def test_twin_engine_and_async_snapshots(lib):
    """BackboneEngine.twin(): a second engine that READS THE FIRST ONE'S ARENA (ntts_backbone_share_arena, ABI 7: what bench.py's engine gangs
    run on) and one filled by a device-to-device copy of it (share=False) generate the same ids as the engine they were made from; and the asynchronous snapshot calls (poll_begin / poll_end /
    read_finished, ABI 5) report what the blocking poll / read report, refuse to be opened twice and refuse rows that were not finished."""
    z, cfg, w = load_fixture("backbone_tiny")
    S, N, mn, eos = int(z["s_len"]), 8, int(z["min_new"]), int(z["eos"])
    eng = make_engine(cfg, w, lib, max_batch=2)
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


def test_engine_on_a_caller_provided_stream(lib):
    """SURVEY.md 8b / VERDICT r3 missing 6: ntts_backbone_set_stream -- prompt pass, graph-replayed decode steps and the device-side code
    export enqueued on a stream the CALLER owns (a torch stream here): a torch op put on that stream right behind them sees their result
    without any engine-side synchronisation, the ids are those of the engine's own stream, and NULL hands the engine its stream back."""
    z, cfg, w = load_fixture("backbone_small_walk")
    S, N, eos = int(z["s_len"]), 24, int(z["eos"])
    eng = make_engine(cfg, w, lib, max_batch=2, max_context=160, bf16_upload=True)
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=False)
    prompts = [br.synthetic_prompt(cfg, 0, S), br.synthetic_prompt(cfg, 1, S)]
    mine = torch.cuda.Stream()
    eng.set_stream(mine.cuda_stream)
    codes = torch.zeros((2, 32), dtype=torch.int32, device="cuda")
    lens = torch.zeros(2, dtype=torch.int32, device="cuda")
    eng.prefill(prompts, [0, 1], [samp, samp])
    eng.decode(N - 1)
    eng.export_codes([0, 1], 0, cfg.vocab_size, codes.data_ptr(), 32, lens.data_ptr())
    with torch.cuda.stream(mine):                 # ordered behind the engine's work by the stream alone
        got = (codes + 0).cpu().numpy()
        n = lens.cpu().numpy()
    for u in (0, 1):
        assert n[u] == N
        assert_walk_exact(got[u, :N].tolist(), z[f"bf16_ids_{u}"][:N].tolist())
    eng.release_many([0, 1])
    eng.set_stream(None)                          # back on the engine's own stream
    eng.prefill(prompts[:1], [0], [samp])
    eng.decode(N - 1)
    assert eng.read(0)[0] == z["bf16_ids_0"][:N].tolist()
    eng.close()
