"""The engine's GENERAL attention geometry (round 6): head_dim 128 and / or Qwen3-style per-head q/k RMSNorm -- what the
reference's AutoModelForCausalLM dispatch (ref:neutts/neutts.py:164) hands over for a Qwen3-based backbone -- at the logits level against the
oracle (oracle/backbone_ref.py, pinned bit for bit to transformers.Qwen3ForCausalLM: tests/test_oracle_pin.py).  Shared by the emulator suite
(small widths) and the GPU suite (Qwen3-0.6B's widths).  RANDOM-init weights, ragged prompts that straddle the 32-token page edges, slots on
pages a previous occupant left dirty, teacher-forced decode steps: every step's fp32 logits row at the oracle's top-4 ids, in bf16 ulps of the
oracle's value (the bars of tests/test_gpu_parity_matrix.py: mean <= 0.8, max <= 3.5).
hf:models/qwen3/modeling_qwen3.py Qwen3Attention.forward (q_norm / k_norm before RoPE), eager_attention_forward (scaling = head_dim ** -0.5)."""
import numpy as np
import torch

from oracle import backbone_ref as br
from neutts import _hip
from common import bf16_ulp, make_engine


def run_case(lib, cfg, lens, max_batch, n_steps, seed=41, knobs_tag=""):
    eos = cfg.vocab_size - 1
    w = br.make_weights(cfg, seed)
    wd = br.cast_weights(w, torch.bfloat16)
    golds = []
    for i, n in enumerate(lens):
        p = br.synthetic_prompt(cfg, 300 + i, n)
        r = br.generate(cfg, wd, p, n + n_steps, eos, min_new_tokens=n_steps, keep_logits=True)
        top = [torch.topk(lg, 4) for lg in r.logits]
        golds.append((p, list(r.ids), [t.values.numpy() for t in top], [t.indices.numpy() for t in top]))
    B = max_batch
    samp = lambda p, n: _hip.Sampling(max_length=len(p) + n, min_new_tokens=n, eos_token_id=eos, do_sample=False)
    eng = make_engine(cfg, w, lib, max_batch=B, max_context=((max(lens) + n_steps + 2 + 31) // 32) * 32, max_prefill_tokens=4096, bf16_upload=True)
    try:
        eng.set_debug(True)
        rounds = [list(range(len(lens)))] if B >= len(lens) else [[i] for i in range(len(lens))][:max(1, 2 // B + 1)]
        stats, exact, tie, rows = [], 0, 0, 0
        for which in rounds:
            gs = [golds[i] for i in which]
            step = max(1, (B - 1) // max(1, len(gs) - 1)) if len(gs) > 1 else 1
            slots = [min(B - 1, k * step) for k in range(len(gs))]          # spread over the batch: first ... last row
            assert len(set(slots)) == len(slots)
            # a previous occupant of every checked slot (other contents, same lengths), decoded past the positions the checked run writes
            junk = [br.synthetic_prompt(cfg, 900 + i, len(g[0])) for i, g in enumerate(gs)]
            eng.prefill(junk, slots, [samp(p, n_steps + 2) for p in junk])
            eng.decode(n_steps + 1)
            eng.release_many(slots)
            eng.prefill([g[0] for g in gs], slots, [samp(g[0], n_steps) for g in gs])
            for k in range(n_steps):
                if k:
                    eng.decode(1)
                for s, (p, ids_g, topv, topi) in zip(slots, gs):
                    ids, _ = eng.read(s)
                    assert len(ids) == k + 1
                    row = eng.read_logits(s)
                    for i, v in zip(topi[k], topv[k]):
                        if np.isfinite(v):
                            stats.append(abs(float(row[int(i)]) - float(v)) / bf16_ulp(float(v)))
                    rows += 1
                    if ids[-1] == ids_g[k]:
                        exact += 1
                    else:
                        band = 2.0 * bf16_ulp(topv[k][0])
                        cand = {int(i): float(v) for i, v in zip(topi[k], topv[k])}
                        assert ids[-1] in cand and topv[k][0] - cand[ids[-1]] <= band, (s, k, ids[-1], ids_g[k], cand)
                        tie += 1
                        if k + 1 < n_steps:
                            eng.debug_force(s, ids_g[k])
            eng.release_many(slots)
        st = np.asarray(stats)
        print(f"general attention path {knobs_tag}(head_dim {cfg.head_dim}, qk_norm {cfg.qk_norm}, batch {B}): {exact} exact + {tie} near-tie of {rows} steps; "
              f"logits error at the oracle's top-4: mean {st.mean():.3f}, max {st.max():.2f} bf16 ulps")
        assert exact + tie == rows and exact >= 0.8 * rows
        assert st.mean() <= 0.8 and st.max() <= 3.5, (st.mean(), st.max())
        stt = eng.kv_stats()
        assert stt["free_pages"] == stt["total_pages"]
    finally:
        eng.close()
