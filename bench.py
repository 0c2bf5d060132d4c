#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: codec-tokens/s (+ real-time factor), NeuTTS-Air bf16, batch 256 per GPU.

One "step" = one pass of the synthesis hot path over one batch of synthetic input: 256 utterances per GPU,
each a 500-token prompt (prefill) followed by 250 greedy tokens (1 from the prefill logits + 249 decode
steps, EOS masked so every run does identical work), through libneutts_hip.so.  Inputs (prompt ids,
weights) are resident / uploaded before the timed region; `value` = tokens of ALL ranks / max-over-ranks
wall time of the K timed steps.

    python bench.py                      # N=1, K=3, W=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        # one rank per GPU, RCCL weight broadcast, no step collective

Extra legs on rank 0 at N=1: `roofline` (dominant kernel of the decode step, hipEvent-timed on the engine
stream at mid-generation slot state, cycling over the layers so operands are HBM-cold as inside the step) and
`cpu_baseline` (the reference's CPU path for this hot path, one utterance, on the box's host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "neutts-air_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(cfg, w, prompt, n_new, eos, codec_cfg, codec_w, max_seconds=30.0):
    """The reference's own CPU path for this hot path on the box's host cores: transformers
    Qwen2ForCausalLM.generate called as in ref:neutts/neutts.py:338-347 (fp32, greedy so the work is fixed;
    kind "reference") followed by the NeuCodec-decoder restatement (oracle/codec_ref.decode_code) on the
    produced codes.  Thread count: a short sweep over {8, 16, 32, 64} (bounded probes of the same call) picks the
    fastest -- HF's tiny per-token ops crawl when spread over hundreds of threads, and crawl on too few -- and the
    reported sample runs at that count.  BOUNDED: every generate() gets max_time, so the sample is one utterance or
    the part of it that fits; falls back to the oracle port when transformers is missing (kind "port")."""
    from oracle import backbone_ref as br     # the ONLY place bench.py touches oracle/: the reported CPU baseline
    from oracle import codec_ref as cr
    ncpu = os.cpu_count() or 1
    forced = os.environ.get("NTTS_CPU_BASELINE_THREADS")
    cand = [int(forced)] if forced else sorted({min(c, ncpu) for c in (8, 16, 32, 64)})
    t0 = time.time()
    sweep = {}
    try:
        from oracle.gen_golden import hf_backbone
        m = hf_backbone(cfg, w, torch.float32)
        ids_t = torch.tensor([prompt])

        def gen(n, tmax):
            return m.generate(ids_t, max_length=len(prompt) + n, eos_token_id=eos, pad_token_id=eos, do_sample=False,
                              use_cache=True, min_new_tokens=n, max_time=tmax)
        if len(cand) > 1:
            for c in cand:                      # probe: the prompt pass + 12 decode steps
                torch.set_num_threads(c)
                tp = time.time()
                gen(min(12, n_new), 10.0)
                sweep[c] = round(time.time() - tp, 3)
            cores = min(sweep, key=sweep.get)
        else:
            cores = cand[0]
        torch.set_num_threads(cores)
        t1 = time.time()
        out = gen(n_new, max_seconds)
        ids = out[0, len(prompt):].tolist()
        kind = "reference"
    except Exception as ex:  # transformers missing on this box
        log(f"[cpu_baseline] transformers path unavailable ({type(ex).__name__}: {ex}); timing the oracle port")
        cores = min(ncpu, 32)
        torch.set_num_threads(cores)
        wd = br.cast_weights(w, torch.float32)
        t1 = time.time()
        ids = br.generate(cfg, wd, prompt, len(prompt) + min(n_new, 32), eos, min_new_tokens=min(n_new, 32)).ids
        kind = "port"
    t2 = time.time()
    n = len(ids)
    codes = torch.tensor([i % 65536 for i in ids], dtype=torch.long)[None, None, :]
    wav = cr.decode_code(codec_cfg, codec_w, codes)
    t3 = time.time()
    assert wav.shape[-1] == codec_cfg.hop_length * n
    dt = t3 - t1
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": n / dt, "unit": "codec-tokens/s", "cores": cores, "kind": kind,
            "sample": f"1 utterance, {len(prompt)} prefill + {n} greedy tokens "
                      f"({'transformers Qwen2ForCausalLM.generate' if kind == 'reference' else 'oracle/backbone_ref.generate'}"
                      f", fp32 torch CPU) {t2 - t1:.1f}s + NeuCodec-decoder restatement on the {n} codes {t3 - t2:.2f}s "
                      f"(+{t1 - t0:.1f}s model build and thread sweep, untimed); host has {ncpu} logical cores",
            "thread_sweep_probe_s": sweep, "torch_num_threads": torch.get_num_threads(), "cpu_model": cpu_model,
            "backbone_tokens_per_s": n / (t2 - t1), "codec_frames_per_s": n / (t3 - t2),
            "rtf": dt / (n / 50.0)}


ROCPROF_SYMBOLS = [   # rocprofv3 symbol prefix -> the step's logical kernels it serves
    ("attn_decode_kernel<", ["attn_decode_kernel"]),
    ("qkv_rope_kernel<", ["gemm_qkv"]),                                         # QKV projection + bias + RoPE + K append (qkv_rope.h)
    ("gemm_kernel<4, 1, 1, 2, 4,", ["gemm_o_proj_splitk", "gemm_down_splitk"]),
    ("gemm_kernel<8, 1, 2, 2, 3,", ["gemm_o_proj_splitk", "gemm_down_splitk"]),   # the gang's 256-row tile (ntts_backbone_set_gang): under the profiler the four chains run
                                                                                  # one after the other, so this symbol's average is the tile ALONE (slower than the 64-row one)
    ("gemm_kernel<4, 2, 2, 1, 3,", ["gemm_gate_up_silu"]),
    # the WIDE decode shape (engines of 512+ slots: bench.py's default since round 6): gate/up on 256 x 192, down_proj on 128 x 128 / 8 waves / 4 K slices,
    # o_proj on whole-K 64 x 64 tiles with the residual add in the epilogue
    ("gemm_kernel<4, 3, 4, 1, 2,", ["gemm_gate_up_silu"]),
    ("gemm_kernel<4, 2, 2, 2, 3,", ["gemm_down_splitk"]),
    ("gemm_kernel<4, 1, 1, 6, 4,", ["gemm_o_proj_splitk"]),
    ("gemm_kernel<4, 3, 4, 3, 2, 0, 64, true", ["gemm_lm_head_argmax"]),     # 256 x 288 natural-order tile (NTTS_HEAD_XL=4, the default)
    ("gemm_kernel<4, 4, 4, 3, 2, 0, 64, true", ["gemm_lm_head_argmax"]),     # 256 x 256 tile (NTTS_HEAD_XL=1)
    ("add_rmsnorm_row_kernel", ["add_rmsnorm_kernel"]),
]


def engine_slots_for(steps, batch, gang, cap=1024):
    """Decode slots per engine of the static gang for a timed region of `steps` batches of `batch` utterances: the batches are spread evenly over the
    fewest gang steps whose engines stay within `cap` slots (a gang step of 4 x 1024 utterances takes 3.7 s: a batch still finishes sooner than the 5 s of
    audio it carries), slots rounded up to whole 64-row tiles and never below one batch.  20 batches -> two gang steps of ten = 640 slots on four engines."""
    n_gs = -(-(steps * batch) // (gang * cap))            # gang steps
    per_gs = -(-steps // n_gs)                            # batches per gang step
    return max(batch, -(-(per_gs * batch) // (gang * 64)) * 64)


def latest_profile(suffix):
    """The newest committed profiles/rNN*<suffix> (files are named per round and session: r03i_..., r04b_...)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*" + suffix)))
    return c[-1] if c else os.path.join(ROOT, "profiles", "missing" + suffix)


def mfma_util_table(path):
    """Matrix-core utilisation per kernel symbol from the committed counter summary (tools/mfma_util_summary.py over a
    `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16` pass of bench.py):
    {"source": file, "command": its header line, "kernels": {symbol: {"mfma_util_pct", "launches", "bf16_gflop_per_launch"}}}."""
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return None
    out = {}
    for ln in lines:
        if ln.startswith("#") or ln.startswith("kernel "):
            continue
        parts = ln.rsplit(None, 5)
        if len(parts) != 6:
            continue
        try:
            out[parts[0].strip()] = {"launches": int(parts[1]), "mfma_util_pct": float(parts[4]), "bf16_gflop_per_launch": float(parts[5])}
        except ValueError:
            continue
    if not out:
        return None
    return {"source": os.path.relpath(path, ROOT), "command": lines[0].lstrip("# ") if lines and lines[0].startswith("#") else None,
            "peak": "2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md); MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)",
            "kernels": {k: v for k, v in out.items() if v["mfma_util_pct"] > 0}}


def continuous_leg(static_value, a):
    import subprocess
    req = int(os.environ.get("NTTS_BENCH_CONT_REQUESTS", "32768"))    # (16 generations of the gang's 2048 slots, ~ 48 s: the finite job's ramp and drain cost 4 % of half that job, 1 / 4 of an eighth's time)
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "continuous", "--requests", str(req), "--steps", "1", "--warmup", "1", "--warmup-requests", str(req // 8),
           "--no-cpu-baseline", "--no-roofline", "--gang", str(max(1, a.gang)), "--prefill", str(a.prefill), "--decode", str(a.decode)]
    if os.environ.get("NTTS_BENCH_CONT_SLOTS"):
        cmd += ["--engine-slots", os.environ["NTTS_BENCH_CONT_SLOTS"]]
    t0 = time.time()
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=float(os.environ.get("NTTS_BENCH_CONT_TIMEOUT", "240")))
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
        r = json.loads(line)
        ss = (r.get("phase_ms") or {}).get("steady_state_tokens_per_s")
        return {"value": r["value"], "unit": r["unit"], "ratio_to_static": r["value"] / static_value, "requests": req,
                "steady_state_value": ss, "steady_state_ratio_to_static": (ss / static_value) if ss else None,
                "steady_state_note": "tokens of the requests finished between the 25 % and 75 % marks of the job over that time: the finite job's ramp and drain left out",
                "ms_per_step": r["ms_per_step"], "workload": r["config"]["workload"][:400], "command": " ".join(cmd[1:]),
                "wall_s_incl_startup": round(time.time() - t0, 1), "phase_ms": r.get("phase_ms")}
    except Exception as ex:  # noqa: BLE001  (a sub-record must never cost the headline line)
        return {"error": repr(ex)[:300], "command": " ".join(cmd[1:])}


def rocprof_symbols(path, live):
    """live: logical kernel -> (ms per launch, algorithmic bytes per launch, launches per step) from this run's HIP events."""
    try:
        lines = open(path).read().splitlines()
    except OSError:
        return None
    out = []
    for ln in lines:
        for prefix, names in ROCPROF_SYMBOLS:
            if ln.startswith(prefix) and all(n in live for n in names):
                f = ln.split()
                calls, total_ms, avg_us, share = int(f[-6]), float(f[-5]), float(f[-4]), float(f[-1])
                nl = sum(live[n][2] for n in names)
                alg = sum(live[n][1] * live[n][2] for n in names) / nl
                live_us = sum(live[n][0] * live[n][2] for n in names) / nl * 1e3
                out.append({"symbol": prefix.rstrip(",< ") , "serves": names, "calls": calls, "share_pct": share, "avg_us": avg_us,
                            "launches_per_step": nl, "mean_alg_bytes": alg, "GBps": alg / (avg_us * 1e-6) / 1e9,
                            "frac_of_hbm_peak": alg / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "live_avg_us": live_us,
                            "agree": abs(avg_us - live_us) <= 0.15 * live_us})
    if not out:
        return None
    out.sort(key=lambda r: -r["share_pct"])
    return {"summary": os.path.relpath(path, ROOT), "command": "NTTS_BENCH_PRIME_STEPS=2 rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 0 --engine-slots 640 "
            "--no-cpu-baseline --no-roofline", "dominant_symbol_by_share": out[0]["symbol"], "symbols": out}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)       # (the driver's own --steps 20 --warmup 5: the no-flag run is the same measurement)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["air-bf16", "nano-fp8", "nano-bf16"], default="air-bf16",
                    help="air-bf16: NeuTTS-Air bf16 (BASELINE.json configs[1..3], the headline metric); nano-fp8: the assumed NeuTTS-Nano "
                         "geometry with fp8 weights / GEMM inputs, batch 512 (configs[4])")
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default 256; 512 for --config nano-fp8)")
    ap.add_argument("--engine-slots", type=int, default=None,
                    help="static (one gang) and continuous mode: decode slots of every engine of the gang.  An engine steps ALL its rows in lock-step "
                         "(one chain; from 512 rows its wide decode shape, GEMMs of that many rows), so a gang step holds gang x slots utterances = "
                         "gang x slots / batch batches of the contract (a `step` stays ONE batch of --batch utterances).  Default for NeuTTS-Air with "
                         "--gang > 1: static -- the --steps K batches are spread evenly over the fewest gang steps of at most 1024 slots per engine "
                         "(K = 20: two gang steps of 10 batches, 640 slots; K = 16: one of 16, 1024 slots; K <= 4: 256 slots); continuous -- 512.  "
                         "Round 6, 4 engines, codec-tokens/s: 256 slots 166 k, 512 191 k, 640 190 k, 768 195 k, 1024 196 k (profiles/r06h_*).")
    ap.add_argument("--prefill", type=int, default=500)
    ap.add_argument("--decode", type=int, default=250)
    ap.add_argument("--vocab", type=int, default=None, help="default: 217488 (NeuTTS-Air), 142080 (assumed Nano)")
    ap.add_argument("--prefill-chunk", type=int, default=64, help="prompts per prefill call")
    ap.add_argument("--mode", choices=["static", "continuous", "stream"], default="static",
                    help="static (default, BASELINE's shape): one batch of equal-length utterances per step.  continuous: --requests "
                         "ragged requests (prompts 0.7-1.3 x --prefill, lengths 0.6-1.4 x --decode) through the continuous-batching "
                         "scheduler -- slots recycled as utterances finish, prefill chunks between decode bursts.  stream: the batch as "
                         "concurrent NeuTTS.infer_stream utterances (ref:neutts/neutts.py:373-465 windows, 0.5 s chunks): the codec pass "
                         "of chunk k on the codec engine's stream beside the decode graph of chunk k + 1 (BASELINE.json configs[4]); "
                         "reports time to first audio and chunk cadence next to tokens/s")
    ap.add_argument("--sample", action="store_true",
                    help="the reference's own sampling call (ref:neutts/neutts.py:338-347: do_sample=True, top_k=50, temperature=1.0; seeded) "
                         "instead of greedy: radix select + Philox multinomial on the bf16 logits rows")
    ap.add_argument("--warmup-requests", type=int, default=None, help="continuous mode: a warm-up step runs the first that many requests only (default: all of them)")
    ap.add_argument("--requests", type=int, default=None, help="continuous mode: requests per step (default 4 x batch)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="static mode: ONE backbone engine, batches strictly one after the other (round-2 shape).  Default: a gang of --gang engines "
                         "on one weight arena, --gang batches in flight at a time")
    ap.add_argument("--pipe-head", type=int, default=100,
                    help="--gangs 2 only: decode steps of gang k enqueued BEFORE gang k + 1's prompt passes (0 = prompt passes first); "
                         "the prompt passes then run beside the later, longer-context steps (profiles/r03i_sweep_pipeline_schedule.txt)")
    ap.add_argument("--gang", type=int, default=4,
                    help="static and continuous mode: that many engines of --batch slots each, reading ONE copy of the weights, each on a lane "
                         "stream (= hardware queue) of its own; their step graphs are replayed ALTERNATELY from the one launching thread.  A decode "
                         "chain is latency-bound (a quarter of the HBM peak alone), four of them fill each other's launch gaps and first round "
                         "trips (tools/probe_two_chains.py: 1.57 -> 0.96-0.99 ms per 256-row step; the runtime has four hardware queues, more "
                         "chains than that run one after the other).  1 = one engine (asynchronous codec pass only)")
    ap.add_argument("--gangs", type=int, default=1, choices=[1, 2],
                    help="pipelined static mode: 2 = two gangs take turns (gang k + 1's prompt passes beside gang k's decode steps, prompt and codec "
                         "passes on one shared stream); 1 = ONE gang, every engine on a lane of its own for all of its work: prompt passes, decode "
                         "chains and codec passes of the gang's batches each run side by side, phase after phase")
    ap.add_argument("--speech-range-head", action="store_true",
                    help="OPT-IN serving option, NEVER the headline: lm_head over 65 536 speech ids + EOS only (ntts_backbone_set_logits_range); "
                         "a separate line, labelled as such")
    ap.add_argument("--stream-gang", action=argparse.BooleanOptionalAction, default=False,
                    help="stream mode: the streams dealt out over --gang engines of batch / gang slots (fp8 Nano-sized model, 512 streams: 150.2 k against "
                         "144.2 k on one engine; NeuTTS-Air bf16 loses: 80.3 k against 100.5 k at 256 streams -- off by default)")
    ap.add_argument("--stream-admit", type=int, default=0, help="stream mode on a gang: streams per admission group (one device-side stream set each); 0 = one group per engine")
    ap.add_argument("--codec-precision", choices=["fp16", "bf16", "high"], default="fp16",
                    help="NeuCodec GEMM operand format: fp16 (the engine's default: ~9.5e-4 relative rms of the fp32 decoder), bf16 (rounds 1-5: 7e-3), "
                         "high (split bf16: ~7e-4 at 3x the matrix-core work)")
    ap.add_argument("--park", type=int, default=None,
                    help="continuous mode: PARKING rows per engine (ABI 9 park_slots): prompts are admitted in waves into them and move into a decode "
                         "slot the moment one is released, so no decode row idles waiting for a wave; default 32 in continuous mode, 0 otherwise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-codec", action="store_true", help="backbone only (profiling aid; not the headline metric)")
    ap.add_argument("--tiny", action="store_true", help="tiny model geometry (plumbing tests only; not a benchmark)")
    a = ap.parse_args()
    # Plumbing-test hook (tests/test_dist_gloo.py): run the launch / shard / broadcast / timing logic on CPU
    # ranks over gloo against the SIMT-emulator build of the library.  Never set for a measurement.
    emu_lib = os.environ.get("NTTS_BENCH_EMU_LIB")

    from neutts import _hip, dist as ndist
    import synthetic as syn                 # model geometry, seeded random weights and prompts (plain data, not the oracle)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        log(f"[bench] WORLD_SIZE={world} but --gpus {a.gpus}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
        if world == 1 and a.gpus > 1:
            sys.exit(2)
    if not emu_lib:
        assert torch.cuda.is_available(), "bench.py measures the HIP path: it needs the MI355X (no CPU fallback)"
        torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emu_lib:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    import importlib.util
    spec = importlib.util.spec_from_file_location("ntts_build", os.path.join(PKG, "build.py"))
    bmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bmod)
    # start-up timeline per rank (read on the first real multi-GPU run): [rank r] stage: seconds since process start
    t_proc = time.time()

    def stage(what):
        log(f"[bench] rank {rank}/{world} +{time.time() - t_proc:6.1f}s  {what}")

    # ONE process per node builds (a stale .so on a fresh box must not start 8 hipcc runs on one output file); the others wait
    if emu_lib:
        lib = emu_lib
    else:
        if local == 0:
            lib = bmod.build(verbose=False)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        if local != 0:
            lib = bmod.LIB
    stage("library ready")
    # the launching thread of each rank on its own share of the host cores (8 ranks x torch's default thread pools otherwise
    # oversubscribe rank 0's weight synthesis and every rank's launch loop)
    if world > 1 and hasattr(os, "sched_setaffinity") and os.environ.get("NTTS_BENCH_AFFINITY", "1") != "0":
        cpus = sorted(os.sched_getaffinity(0))
        nloc = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        per = max(1, len(cpus) // nloc)
        mine = cpus[local * per:(local + 1) * per] or cpus
        try:
            os.sched_setaffinity(0, mine)
            torch.set_num_threads(max(1, min(len(mine), 32)))
        except OSError:
            pass

    nano = a.config.startswith("nano")          # the assumed Nano geometry
    fp8 = a.config == "nano-fp8"
    if a.batch is None:
        a.batch = 512 if nano else 256
    if a.tiny:
        cfg = syn.BackboneConfig.tiny(vocab_size=512, num_layers=1)
    elif nano:
        cfg = syn.BackboneConfig.neutts_nano_like(a.vocab or 142080)
    else:
        cfg = syn.BackboneConfig.neutts_air(a.vocab or 217488)
    fp8_scales = syn.default_fp8_input_scales(cfg) if fp8 else None
    ccfg = syn.CodecConfig.tiny() if a.tiny else syn.CodecConfig.neucodec()
    n_codes = int(np.prod(ccfg.levels))
    Q, S, N = a.batch, a.prefill, a.decode          # Q = the batch of BASELINE's metric = one `step` of the contract
    cont = a.mode == "continuous"
    strm = a.mode == "stream"
    # B = decode slots per engine.  An engine of the gang steps all its rows in lock-step, so it may hold more than one batch of the contract:
    # static mode spreads the K batches of the timed region evenly over the fewest gang steps (at most 1024 slots per engine)
    wide_ok = Q > 1 and a.gang > 1 and (cont or (a.mode == "static" and not a.no_pipeline and a.gangs == 1))
    B = Q
    if wide_ok and a.engine_slots is not None:
        B = max(1, a.engine_slots)
    elif wide_ok and not nano and not a.tiny:
        if cont:
            B = 2 * Q
        else:
            B = engine_slots_for(a.steps, Q, a.gang)
    R = a.requests or 4 * B * (max(1, a.gang) if (a.mode == "continuous" and B > 1) else 1)
    S_max, N_max = (int(S * 1.3) + 1, int(N * 1.4) + 1) if cont else (S, N)
    dev = 0 if emu_lib else local
    eos = cfg.vocab_size - 1
    tts = None
    # stream mode on a gang (round 5): the B streams dealt out over --gang engines of B / gang slots each, admitted in groups of --stream-admit
    Gs = a.gang if (strm and a.stream_gang and a.gang > 1 and B % a.gang == 0 and B // a.gang >= 8 and not emu_lib) else 1
    if strm:
        # Streaming goes through the product's own streaming code (NeuTTS._infer_stream_batch_hip: window / cross-fade semantics of
        # ref:neutts/neutts.py:401-465 for every utterance of the batch, ONE batched codec pass per 25-token chunk on the codec
        # engine's stream while the backbone's next 25 graph replays are already enqueued).  Every rank synthesises its own weights
        # here (a single-GPU configuration: no broadcast leg).
        from neutts import NeuTTS
        w = syn.make_weights(cfg, 0)
        cw = syn.make_codec_weights(ccfg, 0)
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):     # (the class prints its loading messages like the reference; stdout carries the ONE JSON line)
          tts = NeuTTS(
            backbone_repo={"config": dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                          num_layers=cfg.num_layers, num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                          max_context=((S + N + 31) // 32) * 32, max_prefill_tokens=a.prefill_chunk * S,
                                          weight_dtype="fp8" if fp8 else "bf16"),
                           "state_dict": {k: v.numpy() for k, v in w.items()}, "inv_freq": syn.rope_inv_freq(cfg).numpy(),
                           "input_scales": fp8_scales, "tokenizer": None, "speech_base": 0, "eos_token_id": eos},
            backbone_device=f"cuda:{dev}",
            codec_repo={"config": dict(hidden_size=ccfg.hidden_size, intermediate_size=ccfg.intermediate_size, num_layers=ccfg.num_layers,
                                       num_heads=ccfg.num_heads, quantization_dim=ccfg.quantization_dim, levels=list(ccfg.levels),
                                       hop_length=ccfg.hop_length, rms_eps=ccfg.rms_eps, max_frames=128, max_rows=B * 96),
                        "state_dict": {k: v.numpy() for k, v in cw.items()}},
            codec_device=f"cuda:{dev}", do_sample=a.sample, max_batch=B // Gs, engines=Gs, lib_path=lib, codec_precision=a.codec_precision)
        if a.stream_admit > 0:
            tts.stream_admit = a.stream_admit
        tts.stream_on_gang = Gs > 1
        tts.watermarker = None
        tts._ids_to_codes = lambda ids: [int(i) % n_codes for i in ids]     # random weights do not stay in the speech range
        tts._stream_modulo = n_codes                                         # ... the same rule for the device-side streaming path
        tts.stream_on_device = not os.environ.get("NTTS_STREAM_HOST")        # (A/B aid: the round-3 host path)
        tts._ids_to_codes_array = lambda ids: (np.asarray(ids, dtype=np.int64) % n_codes).astype(np.int32)
        tts.min_new_tokens, tts.max_context = N, S + N
    eng = tts.backbone if strm else _hip.BackboneEngine(dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
                                   intermediate_size=cfg.intermediate_size, num_layers=cfg.num_layers,
                                   num_heads=cfg.num_heads, num_kv_heads=cfg.num_kv_heads, rms_eps=cfg.rms_eps,
                                   max_context=((S_max + N_max + 31) // 32) * 32, max_batch=B,
                                   max_prefill_tokens=a.prefill_chunk * S, weight_dtype="fp8" if fp8 else "bf16",
                                   park_slots=(a.park if a.park is not None else (32 if a.mode == "continuous" and B > 1 else 0))), dev, lib)
    codec = tts.codec.engine if strm else None
    if not a.no_codec and not strm:
        codec_cfg_d = dict(hidden_size=ccfg.hidden_size, intermediate_size=ccfg.intermediate_size,
                           num_layers=ccfg.num_layers, num_heads=ccfg.num_heads,
                           quantization_dim=ccfg.quantization_dim, levels=list(ccfg.levels),
                           hop_length=ccfg.hop_length, rms_eps=ccfg.rms_eps, max_frames=N_max,
                           max_rows=B * (N_max + 6), precision=a.codec_precision)
        codec = _hip.CodecEngine(codec_cfg_d, dev, lib)
    if not strm:
        w = cw = None
    t0 = time.time()
    stage("engines created")
    if rank == 0 and not strm:
        w = syn.make_weights(cfg, 0)           # synthetic N(0,1/fan_in) weights at the exact NeuTTS-Air shapes
        eng.load_state_dict({k: v.numpy() for k, v in w.items()}, inv_freq=syn.rope_inv_freq(cfg).numpy(), input_scales=fp8_scales)
        cw = syn.make_codec_weights(ccfg, 0)         # synthetic NeuCodec-decoder weights (xcodec2 parameter names)
    if rank == 0:
        stage("weights synthesised and uploaded (rank 0)")
    if strm:
        pass
    elif world > 1:
        tdev = torch.device("cpu") if emu_lib else None
        ndist.broadcast_weights(eng, src=0, device=tdev)  # RCCL over xGMI: packed backbone arena, one broadcast
        if codec is not None:
            codec_sd = ndist.broadcast_state_dict(cw, src=0, device=tdev)
            codec.load_state_dict(codec_sd)
        stage("weights received (one arena broadcast + one packed codec buffer)")
    elif codec is not None:
        codec_sd = {k: v.numpy() for k, v in cw.items()}
        codec.load_state_dict(codec_sd)
    if a.speech_range_head:
        if strm or cfg.vocab_size < 65536 + 2:
            raise SystemExit("--speech-range-head: static / continuous mode at a vocabulary that holds 65 536 speech ids")
        sp_lo = cfg.vocab_size - 16 - 65536              # the speech ids sit behind the 151 936 base ids, the specials last
        eng.set_logits_range(sp_lo, sp_lo + 65536, eos)
        stage(f"OPT-IN restricted lm_head: ids [{sp_lo}, {sp_lo + 65536}) + eos {eos}")
    # static mode: a TWIN engine (same configuration, its own KV pool and slot state, the arena copied device to device) so that
    # consecutive batches can overlap on the GPU: while engine A replays batch k's decode graphs, engine B runs batch k + 1's prompt pass
    pipe = (not cont) and (not strm) and (not a.no_pipeline) and B > 1
    G = max(1, a.gang) if pipe else 1
    engs = [eng]
    codecs = [codec]
    if pipe:
        NG = a.gangs
        for _ in range(NG * G - 1):
            engs.append(eng.twin(share=os.environ.get("NTTS_BENCH_TWIN_COPY") != "1"))   # (A/B aid: twins with arena copies of their own)
        if codec is not None:
            for _ in range(NG * G - 1):                 # one codec engine (stream, activations, pinned output) per backbone engine
                c2 = _hip.CodecEngine(codec_cfg_d, dev, lib)
                c2.load_state_dict(codec_sd)
                codecs.append(c2)
        stage(f"{len(engs) - 1} twin engine(s) ready (reading engine 0's arena)")
        # Streams -> hardware queues.  The HIP runtime multiplexes all streams of the process onto four hardware queues (least-loaded
        # at creation); two decode chains in one queue run one after the other, and a prompt pass in a chain's queue stalls it.  So:
        # `gang` lane streams, created back to back (torch's pool: containers / streams only), engine i of either gang decodes on
        # lane i (the gangs take turns), and ONE more stream carries every prompt pass and codec pass (matrix-core-bound, they
        # time-slice with each other anyway).  profiles/r04o_*: with the engines' own streams a gang of three had two chains in one queue.
        lanes = None
        if not emu_lib and os.environ.get("NTTS_BENCH_LANES", "1") != "0":
            lanes = [torch.cuda.Stream(device=dev) for _ in range(G + (1 if NG == 2 else 0))]
            for j, e in enumerate(engs):
                e.set_stream(lanes[j % G].cuda_stream)
                if NG == 2:
                    e.set_prefill_stream(lanes[G].cuda_stream)
            for j, c2 in enumerate(codecs):
                if c2 is not None:
                    c2.set_stream(lanes[G if NG == 2 else j % G].cuda_stream)      # (one gang: engine j's codec pass on its own lane)
        if G > 1 and os.environ.get("NTTS_BENCH_GANG_SHAPE", "1") != "0":
            for e in engs:
                e.set_gang(G)       # round 5 (ABI 8): the decode step's tiles / XCD placement chosen for G chains side by side (A/B aid: =0 keeps the single-chain shape)
    if os.environ.get("NTTS_BENCH_PRIME", "1") != "0" and not cont:
        for e2 in engs[1:]:
            e2.warm_up(int(os.environ.get("NTTS_BENCH_PRIME_STEPS", str(max(2, N - 1)))))
        eng.warm_up(int(os.environ.get("NTTS_BENCH_PRIME_STEPS", str(max(2, N - 1)))))   # start-up: graph capture + runtime pools (sized by one decode call of real length), before any request
    stage(f"warm-up done (weights ready in {time.time() - t0:.1f}s)")

    # --sample: the reference's own call (ref:neutts/neutts.py:338-347): top-k 50, temperature 1.0, one Philox key per utterance
    samp = _hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=a.sample, top_k=50, temperature=1.0, seed=20260930)
    samps = [_hip.Sampling(max_length=S + N, min_new_tokens=N, eos_token_id=eos, do_sample=a.sample, top_k=50, temperature=1.0,
                           seed=20260930 + 7919 * (rank * B + i)) for i in range(B)] if a.sample else None
    lo = rank * B
    prompts = [syn.synthetic_prompt(cfg, lo + i, S) for i in range(B)]   # seed 1234 + utterance index

    # device buffers of the id -> code hand-off (torch = tensor container only; the emulator's "device" is host memory)
    # (one pair per engine: batch k's codec pass reads its pair while batch k + 1's export fills the other)
    codes_bufs, lens_bufs, codes_ptrs, lens_ptrs = [], [], [], []
    for _ in engs:
        if emu_lib:
            cb, lb = np.zeros((B, N), dtype=np.int32), np.zeros(B, dtype=np.int32)
            codes_ptrs.append(cb.ctypes.data); lens_ptrs.append(lb.ctypes.data)
        else:
            cb = torch.zeros((B, N), dtype=torch.int32, device=f"cuda:{dev}")
            lb = torch.zeros(B, dtype=torch.int32, device=f"cuda:{dev}")
            codes_ptrs.append(cb.data_ptr()); lens_ptrs.append(lb.data_ptr())
        codes_bufs.append(cb); lens_bufs.append(lb)
    codes_buf, lens_buf, codes_ptr, lens_ptr = codes_bufs[0], lens_bufs[0], codes_ptrs[0], lens_ptrs[0]

    if cont:   # ragged requests, seeded: prompt lengths 0.7 .. 1.3 x S, generated lengths 0.6 .. 1.4 x N (means S and N)
        rng = np.random.default_rng(4321 + rank)
        # (NTTS_BENCH_PSPREAD / _GSPREAD: diagnostic only -- 0 makes the prompts / the generated lengths all equal, to price raggedness itself)
        psp, gsp = float(os.environ.get("NTTS_BENCH_PSPREAD", "0.3")), float(os.environ.get("NTTS_BENCH_GSPREAD", "0.4"))
        r_plen = rng.integers(int(S * (1 - psp)), int(S * (1 + psp)) + 1, size=R)
        r_glen = rng.integers(int(N * (1 - gsp)), int(N * (1 + gsp)) + 1, size=R)
        r_prompts = [syn.synthetic_prompt(cfg, lo * 8 + i, int(r_plen[i])) for i in range(R)]
        r_samp = [_hip.Sampling(max_length=int(r_plen[i] + r_glen[i]), min_new_tokens=int(r_glen[i]), eos_token_id=eos, do_sample=False)
                  for i in range(R)]
    cont_tokens = [0]

    # continuous mode: finished utterances leave through the device-side hand-off as they finish (on_finished hook: export_codes
    # into a staging buffer before the slot is released); every B of them go through ONE codec pass on the codec engine's stream
    # while the backbone keeps decoding the others.  With --gang > 1 the requests are dealt out over that many engines on one arena
    # (neutts._hip.EngineGang: the engines' schedulers advance in turn, their decode chains run side by side), each with a codec
    # engine and staging buffers of its own.
    Gc = max(1, a.gang) if (cont and B > 1) else 1
    gangc = _hip.EngineGang(eng, Gc, lanes=not emu_lib) if Gc > 1 else None
    cengs = gangc.engines if gangc else [eng]
    ccodecs = [codec]
    if cont and codec is not None:
        for _ in range(Gc - 1):
            c2 = _hip.CodecEngine(codec_cfg_d, dev, lib)
            c2.load_state_dict(codec_sd)
            ccodecs.append(c2)
        if gangc is not None and gangc.lane(0):
            for k, c2 in enumerate(ccodecs):
                c2.set_stream(gangc.lane(k))                # engine k's codec passes in engine k's hardware queue (four queues, four lanes)
    if gangc is not None and os.environ.get("NTTS_BENCH_PRIME", "1") != "0":
        for e2 in cengs[1:]:
            e2.warm_up(max(2, N - 1))
    if cont and codec is not None:
        if emu_lib:
            stage_codes = [[np.zeros((B, N_max), dtype=np.int32) for _ in range(2)] for _ in cengs]
            stage_lens = [[np.zeros(B, dtype=np.int32) for _ in range(2)] for _ in cengs]
        else:
            stage_codes = [[torch.zeros((B, N_max), dtype=torch.int32, device=f"cuda:{dev}") for _ in range(2)] for _ in cengs]
            stage_lens = [[torch.zeros(B, dtype=torch.int32, device=f"cuda:{dev}") for _ in range(2)] for _ in cengs]

    def one_step_continuous(collect=False, last=False, nreq=None):
        """One pass over R ragged requests: continuous batching (BackboneEngine.generate: admission by free slots, KV pages and
        prefill budget; ONE burst of decode steps always queued ahead of the host's bookkeeping; finished slots exported on the
        device, released and refilled, new prompts admitted 24 at a time and the engines of a gang in waves -- EngineGang.generate), the codec over every B finished utterances on its own
        stream beside the decode steps of the others."""
        ph = {"generate_wall": 0.0, "codec_tail_wall": 0.0, "codec_passes": 0}
        c0 = {k: sum(e.counters[k] for e in cengs) for k in eng.counters}
        # where the launching thread's time goes: blocked on a snapshot (the GPU is ahead of the host: good) vs inside the enqueueing calls
        host = {"poll_end": 0.0, "poll_begin": 0.0, "decode": 0.0, "prefill": 0.0, "activate": 0.0}
        undo = []

        def timed(e, name):
            f = getattr(e, name)

            def g(*aa, **kk):
                t = time.perf_counter()
                try:
                    return f(*aa, **kk)
                finally:
                    host[name] += time.perf_counter() - t
            setattr(e, name, g)
            undo.append((e, name))
        for e in cengs:
            for name in list(host):
                timed(e, name)
        t1 = time.time()
        st8 = [{"n": 0, "buf": 0, "lens": np.zeros(B, dtype=np.int32), "wavs": None, "busy": False} for _ in cengs]
        which = {id(e): k for k, e in enumerate(cengs)}
        tokens = [0]

        def ptr(x, row):
            return x[row:row + 1].ctypes.data if emu_lib else x[row:row + 1].data_ptr()

        def flush(k, nrows):
            s8 = st8[k]
            if s8["busy"]:
                ccodecs[k].sync()                             # the previous pass has left this codec engine's pinned output buffer
            cb = stage_codes[k][s8["buf"]]
            wv = ccodecs[k].decode_device(cb.ctypes.data if emu_lib else cb.data_ptr(), N_max, s8["lens"][:nrows].copy(), producer_stream=cengs[k].stream())
            s8["busy"], s8["wavs"] = True, wv
            ph["codec_passes"] += 1
            s8["buf"] ^= 1
            s8["n"] = 0

        done_at = []                                      # (wall time, tokens) of every finished request: the steady-state rate below
        codec_rows = min(B, int(os.environ.get("NTTS_BENCH_CODEC_ROWS", str(B))))       # finished utterances per codec pass
        codec_on_wave = int(os.environ.get("NTTS_BENCH_CODEC_ON_WAVE", "0"))           # > 0: also flush that many or more at an engine's admission

        def hook(i, slot, n_new, e=eng):
            t = time.perf_counter()
            try:
                hook_body(i, slot, n_new, e)
            finally:
                host["on_finished"] = host.get("on_finished", 0.0) + time.perf_counter() - t

        def hook_body(i, slot, n_new, e):
            assert n_new == int(r_glen[i]), "continuous run did not produce the expected tokens"
            tokens[0] += n_new
            done_at.append((time.time(), n_new, sum(e_.counters["decode_steps"] for e_ in cengs)))
            if codec is None:
                return
            k = which[id(e)]
            s8 = st8[k]
            row = s8["n"]
            e.export_codes([slot], 0, n_codes, ptr(stage_codes[k][s8["buf"]], row), N_max, ptr(stage_lens[k][s8["buf"]], row), modulo=True)
            s8["lens"][row] = n_new
            s8["n"] = row + 1
            if s8["n"] >= codec_rows:
                flush(k, s8["n"])

        # (burst length: 4 steps per poll on one engine, profiles/r02i_sweep_continuous_sched.txt; 1 on a gang -- the engines' bursts are
        #  enqueued in turn, and the shorter the turn the closer the chains run side by side: 144.5 k at 1-2, 132.0 k at 4, 106.7 k at 8
        #  steps per poll, profiles/r04s_sweep_continuous_gang_sched.txt; with the engines' admissions in waves 149.8 k at 1, 146.3-147.0 k
        #  at 2 against 144.3-145.6 k for independent admissions, profiles/r05m_sweep_continuous_admission_*.txt)
        kw = dict(steps_per_poll=int(os.environ.get("NTTS_BENCH_POLL", "1" if Gc > 1 else "4")), prefill_token_budget=a.prefill_chunk * S,
                  min_admit=int(os.environ.get("NTTS_BENCH_MIN_ADMIT", "24")), on_finished=hook,
                  run_ahead=os.environ.get("NTTS_BENCH_RUN_AHEAD", "1") != "0")
        if gangc is not None:
            kw["admit"] = os.environ.get("NTTS_BENCH_ADMIT", "wave")
            if codec is not None and codec_on_wave > 0:
                # a codec pass holds its engine's lane like a prompt pass does: put it next to the admission wave
                kw["on_admit"] = lambda e, n_prompts: flush(which[id(e)], st8[which[id(e)]]["n"]) if st8[which[id(e)]]["n"] >= codec_on_wave else None      # EngineGang.generate: how the engines' prompt passes are placed against each other
        try:
            nreq = len(r_prompts) if nreq is None else min(nreq, len(r_prompts))       # (a warm-up pass may take the first requests only)
            (gangc or eng).generate(r_prompts[:nreq], r_samp[:nreq], **kw)
        finally:
            for e, name in undo:
                delattr(e, name)                              # (the instance attribute shadowing the method)
        ph["generate_wall"] = (time.time() - t1) * 1e3
        ph["host_ms"] = {k: round(v * 1e3, 1) for k, v in host.items()}
        ph["host_ms"]["python_between_calls"] = round(ph["generate_wall"] - sum(host.values()) * 1e3, 1)
        t1 = time.time()
        wavs = None
        if codec is not None:
            for k, s8 in enumerate(st8):
                if s8["n"]:
                    flush(k, s8["n"])
            for k, s8 in enumerate(st8):
                ccodecs[k].sync()
                if s8["wavs"] is not None:
                    wavs = s8["wavs"][:1, :4000].copy()
        else:
            for e in cengs:
                e.sync()
        ph["codec_tail_wall"] = (time.time() - t1) * 1e3
        cont_tokens[0] = tokens[0]
        assert tokens[0] == int(r_glen[:nreq].sum())
        ph.update({k: sum(e.counters[k] for e in cengs) - c0[k] for k in c0})       # scheduler diagnostics: decode steps issued, prompt passes and their sizes
        ph["slot_occupancy"] = tokens[0] / max(1, ph["decode_steps"] * B)
        ph["park_slots_per_engine"] = eng.park_slots
        # steady state: the tokens of the requests that finished between the moments 25 % and 75 % of all requests were done, over that time
        # -- the finite job's ramp (every slot waits for the first prompt passes) and drain (the last generation thins out) left out
        if len(done_at) >= 64:
            done_at.sort()
            a25, a75 = done_at[len(done_at) // 4], done_at[(3 * len(done_at)) // 4]
            mid = sum(x[1] for x in done_at if a25[0] < x[0] <= a75[0])
            if a75[2] > a25[2]:                                             # ... and how full the decode rows were over the same stretch
                ph["steady_state_slot_occupancy"] = mid / ((a75[2] - a25[2]) * B)
            if a75[0] > a25[0]:
                ph["steady_state_tokens_per_s"] = mid / (a75[0] - a25[0])
        return ph, None, wavs

    pending = []                                          # codec passes in flight: (codec engine, the lens buffer its export filled)
    pstate = {"cur": 0, "ready": [0, 0]}                  # pipelined static mode: whose turn it is, how many prefilled batches each gang holds
    async_codec = os.environ.get("NTTS_BENCH_ASYNC_CODEC", "1") != "0"
    gangs = ([list(range(0, G)), list(range(G, 2 * G))] if NG == 2 else [list(range(G)), list(range(G))]) if pipe else [[0], [0]]     # engine indices

    def finish_pending():
        """Wait for the codec passes enqueued by the previous step (each is ordered behind its batch's decode loop and code export, so
        those are done too) and check that every utterance produced its N tokens: export_codes wrote the counts next to the codes."""
        while pending:
            cdc, lb, *rows_ = pending.pop(0)
            cdc.sync()
            got = lb if emu_lib else lb.cpu().numpy()
            assert (np.asarray(got)[:rows_[0] if rows_ else B] == N).all(), "bench run did not produce the expected tokens"

    def prefill_all(e, rows=None):
        """prompt passes of the first `rows` slots of engine e (default: all of its B slots), --prefill-chunk prompts per call"""
        rows = B if rows is None else rows
        each = []
        for c in range(0, rows, a.prefill_chunk):
            n = min(a.prefill_chunk, rows - c)
            tc0 = time.time()
            e.prefill(prompts[c:c + n], list(range(c, c + n)), samps[c:c + n] if samps else [samp] * n)
            each.append(round((time.time() - tc0) * 1e3, 1))
        return each

    def decode_gang(es, n):
        """n decode steps on every engine of the gang, the graph replays enqueued alternately (engine 0's step j, engine 1's step j, ...):
        each engine's chain runs on its own stream, the GPU fills one chain's launch gaps and first round trips with the other's kernels."""
        if n <= 0:
            return
        if len(es) == 1:
            es[0].decode(n)
            return
        for _ in range(n):
            for e in es:
                e.decode(1)

    def gang_step(nb, nb_next):
        """`nb` batches (one per engine of the current gang) through the pipeline: prompts -> codec-token ids -> 24 kHz waveforms.  The
        launching thread enqueues the NEXT gang's prompt passes (`nb_next` batches, the other gang's streams) between this gang's decode
        graphs (after --pipe-head of the 249), and this gang's codec passes (codec engines' streams) before the next iteration.  A timed
        region is self-contained: its first gang runs its own prompt passes un-overlapped (nothing is ready), its last gang starts no
        further batch -- K batches hold K prompt passes, K decode loops, K codec passes."""
        i = pstate["cur"]
        cur, other = [engs[j] for j in gangs[i][:nb]], [engs[j] for j in gangs[1 - i][:nb_next]]
        ph = {}
        tw = [time.time()]
        each = []
        for e in cur[pstate["ready"][i]:]:
            each += prefill_all(e)                                 # pipeline ramp
        nd = N - 1
        head = min(max(a.pipe_head, 0), nd) if other else nd
        ph["host_wall_prefill_each"] = each
        tw.append(time.time())
        decode_gang(cur, head)                                     # gang k: a head start of `head` decode steps ...
        for e in other:
            prefill_all(e)                                         # ... then gang k + 1's prompt passes (the other gang's streams) ...
        pstate["ready"][1 - i] = len(other)
        decode_gang(cur, nd - head)                                # ... beside the rest of gang k's decode steps
        tw.append(time.time())
        wavs = None
        if codec is not None:
            for j, e in zip(gangs[i], cur):
                e.export_codes(list(range(B)), 0, n_codes, codes_ptrs[j], N, lens_ptrs[j], modulo=True)
        for e in cur:
            st, n_new = e.poll()                                    # blocking: this batch's decode (+ export) done
            assert (n_new == N).all() and (st == 2).all(), "bench run did not produce the expected tokens"
        if codec is not None:
            finish_pending()                                        # gang k - 1's waveforms have left the codec engines' pinned buffers
            for j, e in zip(gangs[i], cur):
                wavs = codecs[j].decode_device(codes_ptrs[j], N, np.full(B, N, dtype=np.int32), producer_stream=e.stream())
                pending.append((codecs[j], lens_bufs[j]))
                assert wavs.shape == (B, ccfg.hop_length * N)
        tw.append(time.time())
        for e in cur:
            e.release_many(list(range(B)))
        pstate["ready"][i] = 0
        pstate["cur"] = 1 - i
        tw.append(time.time())
        ph["host_wall_prefill_calls"] = (tw[1] - tw[0]) * 1e3
        ph["host_wall_decode_call"] = (tw[2] - tw[1]) * 1e3
        ph["host_wall_wait_and_codec"] = (tw[3] - tw[2]) * 1e3
        ph["host_wall_release"] = (tw[4] - tw[3]) * 1e3
        ph["host_wall_total"] = (tw[4] - tw[0]) * 1e3
        return ph, None, wavs

    def rows_of(nbq):
        """`nbq` batches of Q utterances dealt out over the gang's engines, B rows each: the rows each engine holds (the last one may be part-filled)"""
        total = nbq * Q
        return [min(B, total - k * B) for k in range((total + B - 1) // B)]

    def gang_step_single(nb, nb_next):
        """--gangs 1: the one gang's `nb` batches (dealt out B rows per engine, in lock-step there), phase after phase, every engine on its own lane:
        [prompt passes side by side] (enqueued at the end of the previous step, behind its codec passes) [decode chains side by side]
        [export + codec passes side by side]."""
        rows = rows_of(nb)
        cur = engs[:len(rows)]
        ph = {}
        tw = [time.time()]
        each = []
        for e, r in list(zip(cur, rows))[pstate["ready"][0]:]:
            each += prefill_all(e, r)
        ph["host_wall_prefill_each"] = each
        tw.append(time.time())
        decode_gang(cur, N - 1)
        tw.append(time.time())
        wavs = None
        if codec is not None:
            finish_pending()                                        # the previous step's waveforms have left the pinned buffers (enqueued a whole decode phase ago)
            for j, (e, r) in enumerate(zip(cur, rows)):
                e.export_codes(list(range(r)), 0, n_codes, codes_ptrs[j], N, lens_ptrs[j], modulo=True)
                wavs = codecs[j].decode_device(codes_ptrs[j], N, np.full(r, N, dtype=np.int32), producer_stream=e.stream())
                pending.append((codecs[j], lens_bufs[j], r))
        for e, r in zip(cur, rows):
            st, n_new = e.poll()                                    # blocking: this batch's decode (+ export) done
            assert (n_new[:r] == N).all() and (st[:r] == 2).all(), "bench run did not produce the expected tokens"
        tw.append(time.time())
        for e, r in zip(cur, rows):
            e.release_many(list(range(r)))
        nxt = rows_of(nb_next)
        for e, r in zip(engs, nxt):
            prefill_all(e, r)                                      # the next step's prompt passes, behind this step's codec passes on each lane
        pstate["ready"][0] = len(nxt)
        tw.append(time.time())
        ph["host_wall_prefill_calls"] = (tw[1] - tw[0]) * 1e3
        ph["host_wall_decode_call"] = (tw[2] - tw[1]) * 1e3
        ph["host_wall_wait_and_codec"] = (tw[3] - tw[2]) * 1e3
        ph["host_wall_release"] = (tw[4] - tw[3]) * 1e3
        ph["host_wall_total"] = (tw[4] - tw[0]) * 1e3
        return ph, None, wavs

    def run_pipelined(K):
        """K batches through the two-gang pipeline; returns per-gang-step [batches, host wall ms, phases]."""
        walls = []
        done = 0
        cap = (G * B) // Q if NG == 1 else G                       # batches per gang step
        while done < K:
            nb = min(cap, K - done)
            nb_next = min(cap, K - done - nb)
            ts = time.time()
            ph_t = (gang_step if NG == 2 else gang_step_single)(nb, nb_next)[0]
            walls.append((nb, round((time.time() - ts) * 1e3, 2), ph_t))
            done += nb
        return walls

    def one_step_static(collect=False, last=False):
        """One pass of the hot path over one batch: prompts -> codec-token ids -> 24 kHz waveforms."""
        assert collect or not pipe, "pipelined static mode runs through run_pipelined()"
        if pipe:                                                     # the untimed phase split runs serially on engine 0: the pipeline is drained
            finish_pending()                                         # (run_pipelined's last gang starts no further batch)
            assert pstate["ready"] == [0, 0], pstate
            pstate["cur"] = 0
        ph = {"prefill": 0.0, "decode": 0.0, "codec": 0.0, "handoff_host": 0.0, "codec_call_wall": 0.0}
        tw = [time.time()]                                   # host wall-clock stamps (reported as phase_ms.host_wall_*)
        each = []
        for c in range(0, B, a.prefill_chunk):
            n = min(a.prefill_chunk, B - c)
            tc0 = time.time()
            eng.prefill(prompts[c:c + n], list(range(c, c + n)), samps[c:c + n] if samps else [samp] * n)
            each.append(round((time.time() - tc0) * 1e3, 1))
            if collect:
                ph["prefill"] += eng.last_timing()[0]
        ph["host_wall_prefill_each"] = each
        tw.append(time.time())
        eng.decode(N - 1)
        tw.append(time.time())
        wavs = None
        if codec is None:
            eng.sync()
            if collect:
                ph["decode"] = eng.last_timing()[1]
            st, n_new = eng.poll()
            assert (n_new == N).all() and (st == 2).all(), "bench run did not produce the expected tokens"
        else:
            # id -> code hand-off on the device (random weights do not stay in the speech range -> code = id mod 65536),
            # enqueued behind the decode steps; the codec pass is ordered behind it on its own stream; only the waveforms
            # (into the codec engine's pinned buffer) and 2 x B counters cross PCIe
            finish_pending()                                    # the previous batch's codec pass reads codes_buf / lens_buf and fills the pinned
                                                                # output: it finished ~0.4 s ago; formally ordered before they are reused
            eng.export_codes(list(range(B)), 0, n_codes, codes_ptr, N, lens_ptr, modulo=True)
            th = time.time()
            st, n_new = eng.poll()                              # blocking: decode + export done
            if collect:
                ph["decode"] = eng.last_timing()[1]
            assert (n_new == N).all() and (st == 2).all(), "bench run did not produce the expected tokens"
            # (Measured and not kept, profiles/r02k_ab_bench_async_modes.txt: with the poll dropped too -- export, codec pass, slot
            #  releases and the next batch's prompt pass are all stream-ordered behind the decode steps, so the host can enqueue a whole
            #  batch ahead -- the step got SLOWER, 583 vs 573 ms: the launching thread then runs into a full hardware queue (its fourth
            #  prompt-pass call blocks 165 ms) and the decode loop behind it loses 10 ms.)
            ph["handoff_host"] = (time.time() - th) * 1e3
            tc = time.time()
            wavs = codec.decode_device(codes_ptr, N, np.full(B, N, dtype=np.int32), producer_stream=eng.stream())
            if async_codec and not collect:
                # batch k's codec pass and the D2H of its waveforms run on the codec engine's stream while the host releases the slots
                # and enqueues batch k + 1's prompt pass (independent data; the pass is waited for before its buffers are reused and
                # inside the closing barrier, so every waveform has landed when the clock stops)
                pending.append((codec, lens_buf))
            else:
                codec.sync()
                ph["codec"] = codec.last_timing()
            ph["codec_call_wall"] = (time.time() - tc) * 1e3
            assert wavs.shape == (B, ccfg.hop_length * N)
        tw.append(time.time())
        eng.release_many(list(range(B)))
        tw.append(time.time())
        ph["host_wall_prefill_calls"] = (tw[1] - tw[0]) * 1e3     # host time inside the (asynchronous) prefill calls
        ph["host_wall_decode_call"] = (tw[2] - tw[1]) * 1e3       # host time enqueueing the decode graphs
        ph["host_wall_wait_and_codec"] = (tw[3] - tw[2]) * 1e3    # blocking: decode done, codec pass, D2H
        ph["host_wall_release"] = (tw[4] - tw[3]) * 1e3
        ph["host_wall_total"] = (tw[4] - tw[0]) * 1e3
        ids = None
        return ph, ids, wavs

    stream_stats = {}

    def one_step_stream(collect=False, last=False):
        """One pass over the batch as B concurrent streams: time to first audio per utterance, the times at which utterance 0's
        0.5 s chunks arrive (every utterance of the batch gets its chunk in the same burst), all waveform samples received."""
        refs = [[int(t) % n_codes for t in p[-372:]] for p in prompts]      # reference codes as long as ref:samples/dave.pt (372)
        if not emu_lib:
            torch.cuda.synchronize()
        t1 = time.time()
        first, burst, n_samp = {}, [], 0
        for i, chunk in tts._infer_stream_batch_hip([list(p) for p in prompts], refs):
            now = (time.time() - t1) * 1e3
            first.setdefault(i, now)
            if i == 0:
                burst.append(now)
            n_samp += len(chunk)
        total = (time.time() - t1) * 1e3
        # (the reference's windows carry one overlap frame past the last chunk: N or N + 1 frames of audio per utterance)
        assert B * N * ccfg.hop_length <= n_samp <= B * (N + 2) * ccfg.hop_length, (n_samp, B * N * ccfg.hop_length)
        tt = np.array(sorted(first.values()))
        ph = {"ttfa_ms_first": float(tt[0]), "ttfa_ms_median": float(np.median(tt)), "ttfa_ms_last": float(tt[-1]), "total_ms": total,
              "chunks_per_utterance": len(burst), "chunk_arrival_ms_utt0": [round(x, 1) for x in burst],
              "mean_chunk_period_ms": (burst[-2] - burst[0]) / max(len(burst) - 2, 1) if len(burst) > 2 else None,
              "audio_s_per_utterance": n_samp / B / (50.0 * ccfg.hop_length)}
        stream_stats.update(ph)
        return ph, None, None

    one_step = one_step_stream if strm else one_step_continuous if cont else one_step_static

    def barrier():
        finish_pending()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        if not emu_lib:
            torch.cuda.synchronize()

    staticpipe = pipe and one_step is one_step_static
    if staticpipe and os.environ.get("NTTS_BENCH_PRIME", "1") != "0":
        # start-up, like warm_up(): every engine pair (backbone + codec) sees one prompt-pass chunk, a decode step, an export and a codec
        # pass of the real shapes once (pinned output buffers, runtime pools), whatever --warmup is -- the warm-up steps below go
        # through gangs, i.e. through the first engines only
        for j, e in enumerate(engs):
            n = min(a.prefill_chunk, B)
            e.prefill(prompts[:n], list(range(n)), samps[:n] if samps else [samp] * n)
            e.decode(min(2, N - 1))
            if codec is not None:
                e.export_codes(list(range(n)), 0, n_codes, codes_ptrs[j], N, lens_ptrs[j], modulo=True)
                codecs[j].decode_device(codes_ptrs[j], N, np.full(B, N, dtype=np.int32), producer_stream=e.stream())
                codecs[j].sync()
            e.sync()
            e.release_many(list(range(n)))
        stage("every engine primed")
    if staticpipe:
        run_pipelined(a.warmup)
    for k in range(0 if staticpipe else a.warmup):
        if cont and a.warmup_requests:
            one_step(last=(k == a.warmup - 1), nreq=a.warmup_requests)
        else:
            one_step(last=(k == a.warmup - 1))
    barrier()
    t0 = time.time()
    step_wall = []                                   # per-step host wall time (diagnostic; `value` uses the barrier-bracketed total)
    step_host = []                                   # static mode: host wall of [prefill calls, decode enqueue, wait + codec] per step
    if staticpipe:                                   # K batches = K "steps" of the contract, `gang` of them in flight per gang step
        for nb, wall_ms, ph_t in run_pipelined(a.steps):
            step_wall.append([nb, wall_ms])
            step_host.append([ph_t["host_wall_prefill_each"]] + [round(ph_t[k], 1) for k in ("host_wall_prefill_calls", "host_wall_decode_call", "host_wall_wait_and_codec")])
    for k in range(0 if staticpipe else a.steps):
        ts = time.time()
        ph_t = one_step(collect=bool(os.environ.get("NTTS_BENCH_STEP_PHASES")), last=(k == a.steps - 1))[0]   # (diagnostic: per-step GPU phases add syncs)
        step_wall.append(round((time.time() - ts) * 1e3, 2))
        if "host_wall_prefill_calls" in ph_t:
            step_host.append([ph_t["host_wall_prefill_each"]] + [round(ph_t[k], 1) for k in ("host_wall_prefill_calls", "host_wall_decode_call", "host_wall_wait_and_codec")]
                             + ([round(ph_t[k], 1) for k in ("prefill", "decode", "codec")] if os.environ.get("NTTS_BENCH_STEP_PHASES") else []))
    barrier()
    dt = time.time() - t0
    ranks_seen, host_stats = world, None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if emu_lib else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # 8-GPU hedges: every rank must be IN the timed job (a rank that silently fell out of the group would make
        # `value` = world x tokens a lie), and the launching threads' host time per step -- eight ranks' launch threads and pollers share one
        # host -- is reported as max / mean over ranks so that a host-bound rank shows in the driver's line
        one = torch.ones(1, dtype=torch.float64, device=t.device)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(one.item())))
        assert ranks_seen == world == dist.get_world_size(), f"{ranks_seen} ranks took part in the timed region, WORLD_SIZE says {world}"
        mine = [float(np.mean([sum(x[1:4]) if isinstance(x[0], list) else sum(x[:3]) for x in step_host])) if step_host else 0.0,
                float(np.mean([w_[1] if isinstance(w_, list) else w_ for w_ in step_wall])) if step_wall else 0.0]
        hm = torch.tensor(mine, dtype=torch.float64, device=t.device)
        hmax, hsum = hm.clone(), hm.clone()
        dist.all_reduce(hmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(hsum, op=dist.ReduceOp.SUM)
        host_stats = {"host_ms_in_calls_per_step": {"max_over_ranks": float(hmax[0]), "mean_over_ranks": float(hsum[0]) / world},
                      "step_wall_ms": {"max_over_ranks": float(hmax[1]), "mean_over_ranks": float(hsum[1]) / world}}

    # ---- untimed extra legs (phase split, roofline of the dominant kernel, CPU baseline)
    gang_shape = pipe and G > 1 and os.environ.get("NTTS_BENCH_GANG_SHAPE", "1") != "0"
    if gang_shape:
        eng.set_gang(1)      # the serial pass and the per-kernel replays below describe ONE engine alone: its single-chain decode shape
    ph, ids, wavs = one_step(collect=True)
    if wavs is not None:
        assert np.isfinite(wavs[:4]).all(), "non-finite waveform"
    roof = None
    step_info = None
    if rank == 0 and not a.no_roofline and not cont and not strm:
        # slot state at mid-generation (mean context S + N/2 ~ 625): prefill everything, decode N/2 steps
        for c in range(0, B, a.prefill_chunk):
            n = min(a.prefill_chunk, B - c)
            eng.prefill(prompts[c:c + n], list(range(c, c + n)), samps[c:c + n] if samps else [samp] * n)
        eng.decode(N // 2)
        eng.sync()
        step_bytes = eng.step_bytes()
        eng.decode(8)
        eng.sync()
        step_ms = eng.last_timing()[1] / 8
        rows = []
        for k, name in enumerate(_hip.BackboneEngine.KERNELS):
            ms, nbytes, nl = eng.time_kernel(k, 48)   # 2 sweeps over the 24 layers: HBM-cold, like the step
            rows.append((ms * nl, name, ms, nbytes, nl))
            log(f"[roofline] {name:24s} {ms * 1e3:8.1f} us/launch x {nl:3d} = {ms * nl:7.3f} ms/step   "
                f"{nbytes / 1e6:9.2f} MB/launch   {nbytes / (ms * 1e-3) / 1e9:7.0f} GB/s")
        # the same step with the gang's other engines decoding beside it (what the timed region runs): G chains at the same context,
        # step graphs replayed alternately; wall time of 16 steps each over G x 16 (HIP events see one stream, the chains run on G)
        gang_step_ms = None
        side = {}
        if gang_shape:
            eng.sync()
            eng.set_gang(G)  # ... and from here on the step as the timed region runs it
        if pipe and G > 1:
            for e2 in engs[1:G]:
                for c in range(0, B, a.prefill_chunk):
                    n = min(a.prefill_chunk, B - c)
                    e2.prefill(prompts[c:c + n], list(range(c, c + n)), samps[c:c + n] if samps else [samp] * n)
                e2.decode(N // 2 + 8)
            for e2 in engs[:G]:
                e2.sync()
            decode_gang(engs[:G], 2)
            for e2 in engs[:G]:
                e2.sync()
            tg = time.perf_counter()
            decode_gang(engs[:G], 16)
            for e2 in engs[:G]:
                e2.sync()
            gang_step_ms = (time.perf_counter() - tg) * 1e3 / (16 * G)
            # ... and the dominant kernels replayed on all G engines AT THE SAME TIME (one host thread per engine, ctypes releases the GIL;
            # rocprofv3's kernel trace serialises the chains, HIP events per stream do not): per-launch time under G-way contention and
            # the bytes per second the G launches move together -- what the kernel delivers inside the gang's decode phase
            import threading
            for k, name in enumerate(_hip.BackboneEngine.KERNELS):
                if emu_lib or name not in ("attn_decode_kernel", "gemm_gate_up_silu", "gemm_down_splitk", "gemm_o_proj_splitk", "gemm_qkv"):
                    continue                              # (the SIMT emulator runs one launch at a time)
                res = [None] * G
                bar = threading.Barrier(G)

                span = [None] * G

                def run_k(j, k=k):
                    bar.wait()
                    t0_ = time.perf_counter()
                    res[j] = engs[j].time_kernel(k, 240)
                    span[j] = (t0_, time.perf_counter())
                th = [threading.Thread(target=run_k, args=(j,)) for j in range(G)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                # bytes of all G x 240 launches over the wall time from the first thread's start to the last one's end (the threads' windows need not
                # coincide exactly: summing the per-stream rates would over-count where they do not)
                wall_s = max(t[1] for t in span) - min(t[0] for t in span)
                gbps = sum(r[1] * 240 for r in res) / wall_s / 1e9
                side[name] = {"us_per_launch": [round(r[0] * 1e3, 2) for r in res],
                              "GBps_total": gbps, "frac_of_hbm_peak_total": gbps / HBM_PEAK_GBPS,
                              "window_overlap": round(sum(r[0] * 1e-3 * 240 for r in res) / G / wall_s, 3)}
            for e2 in engs[1:G]:
                e2.release_many(list(range(B)))
        rows.sort(reverse=True)
        live = {r[1]: (r[2], r[3], r[4]) for r in rows}
        std_cfg = not nano and B == 640 and S == 500 and not a.speech_range_head          # the configuration the committed rocprofv3 / PMC passes were taken on (round 6: the driver's --steps 20 = engines of 640 slots)
        # rocprofv3 view of the same command (committed summary of the same configuration): per SYMBOL, since one gemm
        # template serves two launches per layer.  `agree` = its average duration is within 15 % of this run's HIP events.
        rocprof = rocprof_symbols(latest_profile("_bench_kernel_stats.txt"), live) if std_cfg else None
        # HBM bytes per launch from the committed PMC passes (tools/gpu_round.sh pmc -> tools/pmc_to_json.py), per logical kernel, and
        # the over-fetch ratio traffic / algorithmic bytes (> 1: re-reads through the fabric -- the first thing to fix where it costs time)
        pmc, pmc_src = {}, None
        try:
            if not std_cfg:
                raise OSError("no PMC pass for this configuration")
            pm_path = latest_profile("_pmc_traffic.json")
            with open(pm_path) as fh:
                pm = json.load(fh)
            pmc_src = os.path.relpath(pm_path, ROOT) + ": " + pm["source"]
            for prefix, names in ROCPROF_SYMBOLS:
                for kname, rec in pm["kernels"].items():
                    if kname.startswith(prefix.rstrip(", ")) and all(n in live for n in names):
                        tr = rec["fetch_bytes_per_launch"] + rec.get("write_bytes_per_launch_uncorrected", 0.0)
                        nl_ = sum(live[n][2] for n in names)
                        alg = sum(live[n][1] * live[n][2] for n in names) / nl_
                        pmc["+".join(names)] = {"traffic_bytes_per_launch": tr, "alg_bytes_per_launch": alg, "overfetch": tr / alg}
        except (OSError, KeyError, ValueError, TypeError):
            pass
        # ... and the same counters taken on the GANG's decode shape (NTTS_TALL=3 NTTS_XCD_AFFINE=0 NTTS_QKV_WSTAT=1 on one engine: 256-row
        # o_proj / down_proj tiles with a K slice per XCD pair, QKV column blocks dealt to XCDs) -- what each chain of the timed region moves
        pmc_gang = {}
        try:
            if not std_cfg:
                raise OSError("no PMC pass for this configuration")
            pg_path = latest_profile("_pmc_traffic_gang_shape.json")
            with open(pg_path) as fh:
                pg = json.load(fh)
            for prefix, names in ROCPROF_SYMBOLS:
                for kname, rec in pg["kernels"].items():
                    if kname.startswith(prefix.rstrip(", ")) and all(n in live for n in names):
                        tr = rec["fetch_bytes_per_launch"] + rec.get("write_bytes_per_launch_uncorrected", 0.0)
                        nl_ = sum(live[n][2] for n in names)
                        alg = sum(live[n][1] * live[n][2] for n in names) / nl_
                        pmc_gang["+".join(names)] = {"traffic_bytes_per_launch": tr, "alg_bytes_per_launch": alg, "overfetch": tr / alg, "symbol": kname[:48]}
            pmc_gang["source"] = os.path.relpath(pg_path, ROOT)
        except (OSError, KeyError, ValueError, TypeError):
            pass
        # The dominant kernel: by rocprofv3 SYMBOL share of the same command when the committed summary covers this configuration
        # (VERDICT r3 item 5), else by this run's ms per step.  A symbol that serves several logical kernels (the split-K template:
        # o_proj + down_proj) is reported as their launch-weighted mean.
        names = [rows[0][1]]
        choice = "largest ms/step among the step's logical kernels (HIP events, this run)"
        if rocprof:
            names = rocprof["symbols"][0]["serves"]
            choice = f"largest rocprofv3 symbol share of the same command ({rocprof['summary']}: {rocprof['symbols'][0]['symbol']} {rocprof['symbols'][0]['share_pct']} %)"
        nl = sum(live[n][2] for n in names)
        ms = sum(live[n][0] * live[n][2] for n in names) / nl
        nbytes = sum(live[n][1] * live[n][2] for n in names) / nl
        name = "+".join(names)
        ach = nbytes / (ms * 1e-3) / 1e9
        traffic = pmc.get(name, {}).get("traffic_bytes_per_launch")
        step_frac = step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        wb_ = 1.0 if fp8 else 2.0
        weight_bytes = wb_ * (cfg.num_layers * ((cfg.num_heads + 2 * cfg.num_kv_heads) * cfg.head_dim * cfg.hidden_size + cfg.hidden_size * cfg.num_heads * cfg.head_dim
                                                + 3 * cfg.intermediate_size * cfg.hidden_size) + cfg.vocab_size * cfg.hidden_size)   # W_layers + W_head
        roof = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": pmc_src if traffic is not None else None, "avg_launch_us": ms * 1e3,
                "alg_bytes_per_launch": nbytes, "launches_per_step": nl,
                # the quantity BASELINE's target is stated on ("decode step at >= 60 % of the HBM roofline"): the WHOLE step, not its best kernel
                "step_frac": step_frac, "step_ms": step_ms, "step_alg_bytes": step_bytes,
                # ... and the step as the timed region runs it: G engines' chains side by side (each chain's algorithmic bytes counted
                # in full, the weights too -- every chain streams them, the second and third find most of a layer in the memory-side cache)
                "gang_step": ({"chains": G, "rows_per_chain": B, "ms_per_chain_step": gang_step_ms, "ms_per_256_row_step": gang_step_ms * 256.0 / B, "frac_of_hbm_peak": step_bytes / (gang_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                               # BOTH accountings: `frac_of_hbm_peak` charges every chain its own copy of the weights (G x step_alg_bytes per
                               # gang step); SURVEY 8(d)'s formula at the B = G x 256 rows the GPU actually holds charges them ONCE:
                               "alg_bytes_weights_once": weight_bytes + G * (step_bytes - weight_bytes),
                               "frac_of_hbm_peak_weights_once": (weight_bytes + G * (step_bytes - weight_bytes)) / (G * gang_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                               "counters_note": "rocprofv3 --pmc over gang steps serialises the dispatches (sum of durations / busy time = 1.00 in the kernel trace of the "
                                                "same pass: profiles/r06c_pmc_gang_concurrent_*.txt): per dispatch each chain moves what pmc_per_kernel_gang_shape "
                                                "lists; how much of the second to fourth chain's weight stream the memory-side cache serves is not observable with this tool",
                               "measured": "host wall clock over 16 alternately replayed steps per engine at context ~S + N/2 + 10",
                               # the step's kernels launched on all G engines at the same time: [us per launch on each engine], bytes/s of the G launches together
                               "kernels_side_by_side": side or None}
                              if gang_step_ms else None),
                "dominant_choice": choice,
                "pmc_per_kernel": pmc or None,
                "pmc_per_kernel_gang_shape": pmc_gang or None,
                "mfma_util": mfma_util_table(latest_profile("_mfma_util.txt")) if std_cfg else None,
                "rocprof": rocprof,
                "per_kernel": [{"kernel": r[1], "us": r[2] * 1e3, "launches_per_step": r[4], "alg_bytes": r[3],
                                "GBps": r[3] / (r[2] * 1e-3) / 1e9, "frac": r[3] / (r[2] * 1e-3) / 1e9 / HBM_PEAK_GBPS} for r in rows]}
        step_info = {"ms": step_ms, "alg_bytes": step_bytes, "achieved_GBps": step_bytes / (step_ms * 1e-3) / 1e9,
                     "frac_of_hbm_peak": step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "sum_of_isolated_kernels_ms": sum(r[0] for r in rows),
                     "gang": ({"chains": G, "rows_per_chain": B, "ms_per_256_row_step": gang_step_ms * 256.0 / B, "achieved_GBps": step_bytes / (gang_step_ms * 1e-3) / 1e9,
                               "frac_of_hbm_peak": step_bytes / (gang_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS} if gang_step_ms else None)}
        for s in range(B):
            eng.release(s)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(cfg, w, prompts[0], N, eos, ccfg, cw)

    if rank == 0:
        tokens = world * (cont_tokens[0] if cont else Q * N) * a.steps
        value = tokens / dt
        stages = "backbone prefill + decode loop" + (f" + NeuCodec decoder ({a.codec_precision} GEMM operands, fp32 accumulate / residual stream) to 24 kHz waveform (D2H included)"
                                                     if codec is not None else " (codec skipped: --no-codec)")
        if nano:
            workload = (f"NeuTTS-Nano (ASSUMED geometry: hidden {cfg.hidden_size}, {cfg.num_layers} layers, {cfg.num_heads}:{cfg.num_kv_heads} "
                        f"heads, FFN {cfg.intermediate_size}, V {cfg.vocab_size}) "
                        + ("fp8 e4m3 weights + GEMM inputs on the fp8 MFMA, bf16 KV / attention, " if fp8 else "bf16, ")
                        + f"{world}xMI355X batch={B}, {S} prefill / {N} decode tokens (BASELINE.json configs[4])")
        elif B == 1 and world == 1:
            workload = f"NeuTTS-Air bf16 1xMI355X, batch=1, {S} prefill / {N} decode tokens (BASELINE.json configs[1])"
        else:
            one_eng = (B * N / ((ph["prefill"] + ph["decode"] + ph["codec"]) * 1e-3)) if (pipe and all(k in ph for k in ("prefill", "decode", "codec")) and ph["decode"] > 0) else None
            head = (f"NeuTTS-Air bf16 {world}xMI355X: {(G * B) // Q} batches of {Q} IN FLIGHT per GPU ({G * B} resident"
                    + (f" on {G} engines of {B} slots, each stepping its rows in lock-step" if B > Q else "") + f"; ONE {B}-slot engine alone = "
                    f"{one_eng / 1e3:.1f}k tok/s" + ("; 4 x 256 in flight, round 5's shape: 166.2 k, profiles/r06f_bench.json" if (B > Q and Q == 256 and G == 4) else "") + "), "
                    if (pipe and G > 1 and one_eng) else f"NeuTTS-Air bf16 {world}xMI355X ")
            workload = (head + f"batch={Q} synthetic prompts per GPU, {S} prefill / {N} decode tokens, "
                        f"STATIC batch (all {B} slots of the continuous-batching engine filled at once, every utterance {N} tokens; the ragged "
                        f"scheduler line is --mode continuous)" + ((f", {(G * B) // Q} batches at a time on {G} {B}-slot engines reading ONE copy of the weights: their prompt passes, decode chains (step graphs replayed alternately, one stream = one hardware queue each) and codec passes side by side" + (f", two such gangs taking turns ({2 * G} engines)" if NG == 2 else "") if G > 1 else ", consecutive batches pipelined over two engines") if pipe else "") + f" + hipGraph decode (BASELINE.json configs[{2 if world == 1 else 3}])")
        workload += ", sampling as the reference calls generate (do_sample, top_k=50, temperature=1.0, seeded)" if a.sample else ", greedy"
        if a.speech_range_head:
            workload = ("OPT-IN --speech-range-head (NOT the headline, not the reference's arithmetic outside the range): lm_head over 65 536 speech ids + EOS "
                        "instead of the whole vocabulary; " + workload)
        if strm:
            workload = (f"STREAM mode: {B} concurrent infer_stream utterances per GPU (27-frame windows every 25 tokens, 0.5 s chunks, "
                        f"ref:neutts/neutts.py:401-465), "
                        + (f"dealt out over a gang of {Gs} engines of {B // Gs} slots (one arena, a lane each) in admission groups of {a.stream_admit or B // Gs} streams: per engine "
                           f"turn [windows -> codec pass -> chunks] [next group's prompt pass] [next decode burst], the other engines' bursts beside it; "
                           if Gs > 1 else "codec pass of chunk k on the codec engine's stream beside the decode graph of chunk k + 1; ")
                        + workload)
        if cont:
            workload = (f"CONTINUOUS mode (not BASELINE's static shape): {R} ragged requests per GPU through "
                        + (f"{Gc} engines of {B} decode slots each (one arena, schedulers advanced in turn, decode chains side by side), prompts " if Gc > 1 else f"{B} decode slots, prompts ")
                        + f"{int(S * 0.7)}-{int(S * 1.3)} tokens, {int(N * 0.6)}-{int(N * 1.4)} generated tokens each, slots recycled as utterances finish; "
                        + workload)
        rec = {
            "metric": "codec-tokens/s", "value": value, "unit": "codec-tokens/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8" if fp8 else "bf16", "data": "synthetic",
            "config": {"workload": workload,
                       "batch_per_gpu": Q, "engine_slots": B, "prefill_tokens": S, "decode_tokens": N, "vocab_size": cfg.vocab_size,
                       "stages": stages, "codec_operands": (a.codec_precision if codec is not None else None),
                       "parallelism": f"independent shards x{world}, RCCL weight broadcast only"},
            "tokens_per_s_per_gpu": value / world,
            "rtf": dt / (tokens / 50.0),
            "phase_ms": ph, "step_wall_ms": step_wall, "step_host_wall_ms": step_host,
            "ranks_in_timed_region": ranks_seen, "host_wall_over_ranks": host_stats,
            "roofline": roof, "decode_step": step_info, "cpu_baseline": cpu,
            "timed_region": ((f"warm_up() before timing; {len(engs)} backbone engines of {B} slots on one weight arena, each with a codec engine, on {G} lane streams; one "
                              f"launching thread; a gang step = {(G * B) // Q if NG == 1 else G} batches of {Q} ({B} utterances per engine): [prompt passes] [{N - 1} decode steps per engine, the engines' step graphs "
                              f"replayed alternately] [code export + codec pass + D2H per engine]"
                              + ("; the next gang step's prompt passes are enqueued behind this one's codec passes" if NG == 1 else
                                 "; two gangs take turns: gang k + 1's prompt passes (one shared stream with the codec passes) between gang k's decode graphs")
                              + f"; the timed region is self-contained (K prompt passes, K decode loops, K codec passes for --steps K, the last gang step takes "
                              f"K mod {(G * B) // Q if NG == 1 else G} batches if that is not zero (its last engine part-filled), nothing prefetched before the clock starts, every waveform landed before it stops; a "
                              f"step of the contract = one batch of {Q}); phase_ms is a separate serial pass of {B} utterances on ONE engine") if pipe else
                             "warm_up() before timing, codec pass + D2H of batch k asynchronous under the prompt pass of batch k + 1 (static mode, one engine)"),
            # the same batch through ONE engine, phase after phase (the untimed serial pass behind phase_ms): what a single 256-slot engine
            # delivers on this box, next to the gang's line above
            "one_engine_serial": ({"ms_per_batch": ph["prefill"] + ph["decode"] + ph["codec"],
                                   "codec_tokens_per_s": B * N / ((ph["prefill"] + ph["decode"] + ph["codec"]) * 1e-3)}
                                  if pipe and all(k in ph for k in ("prefill", "decode", "codec")) and ph["decode"] > 0 else None),
            "pipeline": {"engines": len(engs), "gang": G, "gangs": NG, "overlap": ("prefill x gang | decode x gang (chains side by side) | codec x gang" if NG == 1 else "prefill(gang k+1) | decode(gang k: its batches side by side) | codec(gang k-1)"),
                         "decode_head_start_steps": a.pipe_head if NG == 2 else None, "utterances_resident": len(engs) * B} if pipe else None,
        }
        if strm:
            rec["stream"] = stream_stats
        # BASELINE.json configs[2] literally says "continuous batching": the ragged-request scheduler measured in the SAME driver
        # invocation (its own process and engines -- the ragged requests need a longer context than the static engines were created
        # with), as a sub-record with its ratio to the static line above.  Rank 0 of a 1-GPU run only; NTTS_BENCH_CONT_LEG=0 skips it.
        if (world == 1 and not cont and not strm and not nano and not a.tiny and Q == 256 and not emu_lib and not a.no_roofline
                and not a.speech_range_head and os.environ.get("NTTS_BENCH_CONT_LEG", "1") != "0"):
            rec["continuous"] = continuous_leg(value, a)
        print(json.dumps(rec), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
