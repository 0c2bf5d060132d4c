"""N>1 path on CPU: world_size 2 over gloo, emulator library (ROOT cause of any failure on the 8-GPU run
would be host logic: sharding, the one-time arena broadcast, max-over-ranks timing -- all exercised here)."""
import json
import os
import socket
import subprocess
import sys

from neutts import dist as ndist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(nproc, script_args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_shard_range_partitions():
    for n in (0, 1, 5, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            parts = [ndist.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_broadcast_and_sharded_generate_world2(emu_lib):
    r = _torchrun(2, [os.path.join(ROOT, "tests", "_dist_worker.py"), emu_lib])
    assert r.returncode == 0 and "DIST_OK 2" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_contract_world2(emu_lib):
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--tiny", "--batch", "2",
                      "--prefill", "12", "--decode", "4", "--prefill-chunk", "2", "--no-roofline"],
                  {"NTTS_BENCH_EMU_LIB": emu_lib})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert abs(rec["value"] - 2 * 2 * 4 / (rec["ms_per_step"] / 1e3)) < 1e-6 * rec["value"] + 1e-9
    # static mode runs its batches four at a time, one per engine of a gang on one arena (prompt passes, decode chains and codec passes
    # side by side, phase after phase); every batch asserts its token counts from the device-exported lengths: the three steps are
    # a gang step of three batches, the warm-up batch before them went through the first engine alone
    assert rec["pipeline"]["engines"] == 4 and rec["pipeline"]["gang"] == 4 and rec["pipeline"]["gangs"] == 1 and rec["steps"] == 3


def test_bench_contract_world8(emu_lib):
    """The launch the driver uses on an 8-GPU node (`torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8`), on CPU ranks
    over gloo against the emulator build: rank-0-only library step + barrier, rank-local CPU affinity, weight synthesis on rank 0
    -> ONE arena broadcast + one packed codec buffer -> 8 contiguous shards with no collective in the step, max-over-ranks
    timing, exactly one JSON line -- and every rank's start-up timeline on stderr."""
    r = _torchrun(8, [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--tiny", "--batch", "2",
                      "--prefill", "12", "--decode", "4", "--prefill-chunk", "2", "--no-roofline"],
                  {"NTTS_BENCH_EMU_LIB": emu_lib}, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "weak" and rec["cpu_baseline"] is None
    assert abs(rec["value"] - 8 * 2 * 4 / (rec["ms_per_step"] / 1e3)) < 1e-6 * rec["value"] + 1e-9
    for rank in range(8):          # the per-rank start-up timeline (library, engines, weights received, warm-up)
        assert f"rank {rank}/8" in r.stderr and "warm-up done" in r.stderr
    assert r.stderr.count("weights received") == 8 and r.stderr.count("weights synthesised and uploaded (rank 0)") == 1
    # round 5: the job counts its ranks inside the timed region and reports the launching threads' host time over ranks
    assert rec["ranks_in_timed_region"] == 8
    hw = rec["host_wall_over_ranks"]
    for k in ("host_ms_in_calls_per_step", "step_wall_ms"):
        assert hw[k]["max_over_ranks"] >= hw[k]["mean_over_ranks"] > 0


def test_bench_two_gangs_taking_turns_emulator(emu_lib):
    """`bench.py --gang 2 --gangs 2` (the schedule DESIGN.md section 4j measures against the default): two gangs of two engines take turns,
    gang k + 1's prompt passes enqueued between gang k's decode graphs; five batches = two full gang steps and one of a single batch, every
    batch's token counts asserted from the device-exported lengths."""
    import subprocess
    import sys
    env = dict(os.environ, NTTS_BENCH_EMU_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tiny", "--gang", "2", "--gangs", "2", "--batch", "2", "--prefill", "12",
                        "--decode", "4", "--prefill-chunk", "2", "--steps", "5", "--warmup", "1", "--no-roofline", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["pipeline"]["engines"] == 4 and rec["pipeline"]["gangs"] == 2
    assert [nb for nb, _ in rec["step_wall_ms"]] == [2, 2, 1]
    assert abs(rec["value"] - 2 * 4 / (rec["ms_per_step"] / 1e3)) < 1e-6 * rec["value"] + 1e-9


def test_bench_engines_wider_than_a_batch_emulator(emu_lib):
    """`bench.py --engine-slots`: engines of the gang that hold more than one batch of the contract and step their rows in lock-step (round 6: the
    default on the GPU is gang x 640 slots for the driver's --steps 20).  Two engines of 5 slots at --batch 2: a gang step takes 5 batches, seven
    batches = one full gang step and one of two batches (one engine, part-filled); `steps`, `value` and `ms_per_step` stay in batches of --batch."""
    import subprocess
    import sys
    env = dict(os.environ, NTTS_BENCH_EMU_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tiny", "--gang", "2", "--batch", "2", "--engine-slots", "5", "--prefill", "12",
                        "--decode", "4", "--prefill-chunk", "2", "--steps", "7", "--warmup", "1", "--no-roofline", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["steps"] == 7 and rec["config"]["batch_per_gpu"] == 2 and rec["config"]["engine_slots"] == 5
    assert rec["pipeline"]["engines"] == 2 and rec["pipeline"]["utterances_resident"] == 10
    assert [nb for nb, _ in rec["step_wall_ms"]] == [5, 2]
    assert abs(rec["value"] - 2 * 4 / (rec["ms_per_step"] / 1e3)) < 1e-6 * rec["value"] + 1e-9


def test_bench_continuous_mode_emulator(emu_lib):
    """`bench.py --mode continuous` end to end on the emulator build: ragged requests through the run-ahead scheduler, every finished
    utterance exported on the "device" from the on_finished hook, one codec pass per `batch` finished utterances + the ragged tail,
    token counts asserted per request inside the step."""
    import subprocess
    import sys
    env = dict(os.environ, NTTS_BENCH_EMU_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tiny", "--mode", "continuous", "--batch", "2", "--requests", "7",
                        "--prefill", "12", "--decode", "6", "--prefill-chunk", "2", "--steps", "1", "--warmup", "0", "--no-roofline",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["value"] > 0 and rec["phase_ms"]["codec_passes"] == 4      # 7 utterances = 3 full codec batches of 2 + the tail
