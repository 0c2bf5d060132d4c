#!/bin/bash
# round 5, session o: stream mode over engine gangs -- config [4] (fp8 Nano-sized, 512 streams) and NeuTTS-Air bf16 (256 streams): gang (--stream-gang, one admission
# group per engine) against one engine; edit the `run` lines for other group sizes (--stream-admit) / gang sizes (--gang)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/sweep_stream_gang_${TAG:-a}.txt; : > $LOG
run() {
    echo "== $*" | tee -a $LOG
    timeout 300 python bench.py --mode stream --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | grep '^{' | \
        python -c 'import sys, json; r = json.loads(sys.stdin.read()); p = r.get("phase_ms") or {}; print(json.dumps({"value": r["value"], "ms_per_step": r["ms_per_step"], "ttfa": {k: v for k, v in p.items() if "first" in k or "ttfa" in k}}))' | tee -a $LOG
}
run --config nano-fp8 --stream-gang
run --config nano-fp8
run --config air-bf16 --stream-gang
run --config air-bf16
run --config air-bf16 --stream-gang --gang 2
run --config air-bf16 --batch 512 --stream-gang
run --config air-bf16 --batch 512
