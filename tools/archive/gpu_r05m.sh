#!/bin/bash
# round 5, session m: continuous mode on the final decode shape -- admission threshold, burst length and how the engines' prompt passes
# are placed against each other (EngineGang.generate admit = independent / spaced:R / wave), 8192 ragged requests each
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/sweep_continuous_admission_${TAG:-a}.txt; : > $LOG
run() {   # env assignments as arguments
    echo "== $*" | tee -a $LOG
    env "$@" timeout 200 python bench.py --mode continuous --requests 8192 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | \
        python -c 'import sys, json; r = json.loads(sys.stdin.read()); p = r.get("phase_ms") or {}; print(json.dumps({"value": r["value"], "steady": p.get("steady_state_tokens_per_s"), "occupancy": p.get("slot_occupancy"), "decode_steps": p.get("decode_steps"), "prefill_calls": p.get("prefill_calls"), "generate_wall": p.get("generate_wall")}))' | tee -a $LOG
}
run NTTS_BENCH_MIN_ADMIT=24
run NTTS_BENCH_CODEC_ROWS=128
run NTTS_BENCH_CODEC_ROWS=64
run NTTS_BENCH_CODEC_ON_WAVE=32
run NTTS_BENCH_CODEC_ON_WAVE=64
run NTTS_BENCH_CODEC_ON_WAVE=128
run NTTS_BENCH_CODEC_ON_WAVE=16
run NTTS_BENCH_MIN_ADMIT=24
