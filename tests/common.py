"""Shared helpers for the parity tests (oracle = checker only)."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import backbone_ref as br
from neutts import _hip

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    cfg = br.BackboneConfig(**{k: v for k, v in z["cfg"]})
    walk = dict(walk_gain=float(z["walk_gain"]), walk_scale=float(z["walk_scale"])) if "walk_gain" in z.files else {}
    w = br.make_weights(cfg, int(z["seed"]), init=str(z["init"]), peak_sigma=float(z["peak_sigma"]), **walk)
    return z, cfg, w


def engine_cfg(cfg: br.BackboneConfig, **kw):
    d = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
             num_layers=cfg.num_layers, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
             head_dim=cfg.head_dim, qk_norm=getattr(cfg, "qk_norm", False), attention_bias=cfg.attention_bias,
             tie_word_embeddings=cfg.tie_word_embeddings)
    d.update(kw)
    return d


def make_engine(cfg, w, lib, max_batch=2, max_context=128, max_prefill_tokens=512, bf16_upload=False, **kw):
    eng = _hip.BackboneEngine(engine_cfg(cfg, max_batch=max_batch, max_context=max_context,
                                         max_prefill_tokens=max_prefill_tokens, **kw), 0, lib)
    sd = {k: (v.to(torch.bfloat16) if bf16_upload else v.numpy()) for k, v in w.items()}
    eng.load_state_dict(sd, inv_freq=br.rope_inv_freq(cfg).numpy())
    return eng


def bf16_ulp(x: float) -> float:
    x = abs(float(x))
    if x == 0:
        return 2.0 ** -133
    return 2.0 ** (np.floor(np.log2(x)) - 7)


def teacher_forced_compare(eng, slot, gold_ids, gold_topv, gold_topi, max_ulps=2.0, logit_stats=None):
    """Step the engine one token at a time against a golden greedy run.

    A token must equal the golden one unless the golden top-1/top-2 logits are within `max_ulps`
    bf16 ulps of each other (then fp32 summation order legitimately decides; our token must then be
    among the golden top-4 with a logit inside that band).  After such a step the golden token is
    forced so that later steps stay comparable.  Returns (n_exact, n_near_tie).
    With `logit_stats` (a list) and the engine's debug tap enabled, the absolute error of OUR logits at the golden
    top-4 token ids is appended per step, in bf16 ulps of the golden value: the direct measure of numerical fidelity."""
    n_exact = n_tie = 0
    n = len(gold_ids)
    for k in range(n):
        if k > 0:
            eng.decode(1)
        ids, fin = eng.read(slot)
        assert len(ids) == k + 1, (k, len(ids))
        tok = ids[-1]
        if logit_stats is not None:
            row = eng.read_logits(slot)
            for i, v in zip(gold_topi[k], gold_topv[k]):
                if np.isfinite(v):
                    logit_stats.append(abs(float(row[int(i)]) - float(v)) / bf16_ulp(float(v)))
        if tok == int(gold_ids[k]):
            n_exact += 1
        else:
            band = max_ulps * bf16_ulp(gold_topv[k][0])
            cand = {int(i): float(v) for i, v in zip(gold_topi[k], gold_topv[k])}
            assert tok in cand and gold_topv[k][0] - cand[tok] <= band, (
                f"step {k}: got {tok}, golden {int(gold_ids[k])}, golden top4 {cand}, band {band}")
            n_tie += 1
            if k + 1 < n:
                eng.debug_force(slot, int(gold_ids[k]))
    return n_exact, n_tie


assert_free_run_matches = br.assert_free_run_matches


def assert_varied(ids, frac=0.8):
    """A free-running comparison only says something if the run is not a fixed point: `peak_sigma` weights collapse into 1-12
    distinct ids per 250 steps; the walk weights (synthetic._make_walk) emit a new id every step."""
    assert len(set(ids)) >= max(2, int(frac * len(ids))), f"degenerate run: {len(set(ids))} distinct ids in {len(ids)} steps"


def assert_walk_exact(got, want):
    """Walk weights: every top-1 / top-2 margin is tens of bf16 ulps wide, so free-running ids are compared id for id -- no tie clause."""
    got, want = list(got), list(want)
    assert got == want, f"first difference at step {next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))} of {len(want)}"
    assert_varied(got)


# ---------------------------------------------------------------------------------------------- codec
def load_codec_fixture(name):
    from oracle import codec_ref as cr
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    d = {k: v for k, v in z["cfg"]}
    d["levels"] = tuple(d["levels"])
    cfg = cr.CodecConfig(**d)
    return z, cfg, cr.make_weights(cfg, int(z["seed"]))


def make_codec_engine(cfg, w, lib, max_frames=64, max_rows=512, precision="fp16"):
    eng = _hip.CodecEngine(dict(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                num_layers=cfg.num_layers, num_heads=cfg.num_heads, quantization_dim=cfg.quantization_dim,
                                levels=list(cfg.levels), hop_length=cfg.hop_length, rms_eps=cfg.rms_eps,
                                max_frames=max_frames, max_rows=max_rows, precision=precision), 0, lib)
    eng.load_state_dict({k: v.numpy() for k, v in w.items()})
    return eng


def rms(x):
    return float(np.sqrt(np.mean(np.square(np.asarray(x, dtype=np.float64)))))


# ------------------------------------------------------------------------------------------------ reference encoder (f4)
def load_encoder_fixture(name):
    import synthetic as syn
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    d = {k: v for k, v in z["cfg"]}
    d["ratios"], d["levels"] = tuple(d["ratios"]), tuple(d["levels"])
    cfg = syn.EncoderConfig(**d)
    return z, cfg, syn.make_encoder_weights(cfg, int(z["seed"]))


def make_encoder_engine(cfg, w, lib, max_samples=16000):
    d = cfg.to_dict()
    d["max_samples"] = max_samples
    eng = _hip.EncoderEngine(d, 0, lib)
    eng.load_state_dict({k: v.numpy() for k, v in w.items()})
    return eng


def check_encoder_codes(cfg, got_codes, got_lat, gold_codes, gold_lat, lat_tol, tag=""):
    """Integer parity of an FSQ index stream computed in floating point: the twice-bounded latents agree within `lat_tol`,
    every code whose gold latents all sit farther than 2 * lat_tol from a rounding boundary is IDENTICAL, and a code that does
    differ differs only in digits whose gold latent is that close to a boundary (by one level)."""
    assert got_codes.shape == gold_codes.shape and got_codes.dtype == np.int32
    err = np.abs(got_lat - gold_lat).max()
    dist = np.abs(gold_lat - np.floor(gold_lat) - 0.5)            # distance of each gold latent to the nearest x.5
    safe = (dist > 2 * lat_tol).all(axis=1)
    same = got_codes == gold_codes
    print(f"encoder{tag}: latents max |err| {err:.2e} (bound {lat_tol:.0e}); codes equal {same.sum()}/{same.size}; "
          f"{(~safe).sum()} frames within {2 * lat_tol:.0e} of a rounding boundary")
    assert err <= lat_tol, err
    assert same[safe].all(), np.nonzero(~same & safe)[0]
    levels = np.array(cfg.levels)
    basis = np.cumprod(np.concatenate([[1], levels[:-1]]))
    for t in np.nonzero(~same)[0]:
        dg, dr = (got_codes[t] // basis) % levels, (gold_codes[t] // basis) % levels
        bad = np.nonzero(dg != dr)[0]
        assert (np.abs(dg[bad] - dr[bad]) == 1).all() and (dist[t, bad] <= 2 * lat_tol).all(), (t, dg, dr, dist[t])
    return err


# ------------------------------------------------------------------------------------------------ streaming (ref:neutts/neutts.py:373-465)
def reference_stream_chunks(ref_codes, new_codes, dec, hop, chunk=25, lookforward=5, lookback=50, overlap=1):
    """The reference's streaming algorithm (ref:neutts/neutts.py:385-465) restated on a FINISHED code sequence: the chunks `infer_stream`
    must yield, in order.  `dec(codes) -> waveform` is the codec (the oracle's, in the tests).  A window becomes decodable once
    chunk + lookforward undecoded tokens exist; it spans lookback + overlap frames of left context (the reference voice's codes first),
    the chunk and the lookforward, of which chunk + 2 * overlap frames starting at the first undecoded one are kept and cross-faded
    (linear_overlap_add) at a stride of `chunk` frames; the tail takes one overlap frame BEFORE its first undecoded token (asymmetric
    in the reference: kept)."""
    from oracle import codec_ref as cr
    cache, audio, out = list(ref_codes), [], []
    n_tok, n_samp = len(ref_codes), 0
    for c in new_codes:
        cache.append(c)
        if len(cache) - n_tok >= chunk + lookforward:
            t0 = max(n_tok - (lookback + overlap), 0)
            s0 = (n_tok - t0) * hop
            audio.append(dec(cache[t0:n_tok + chunk + lookforward + overlap])[s0:s0 + (chunk + 2 * overlap) * hop])
            mixed = cr.linear_overlap_add(audio, chunk * hop)
            out.append(mixed[n_samp:len(audio) * chunk * hop])
            n_samp = len(audio) * chunk * hop
            n_tok += chunk
    rem = len(cache) - n_tok
    if rem > 0:
        t0 = max(len(cache) - (lookback + overlap + rem), 0)
        audio.append(dec(cache[t0:])[(len(cache) - t0 - rem - overlap) * hop:])
        out.append(cr.linear_overlap_add(audio, chunk * hop)[n_samp:])
    return out
