cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for cfgs in "16 1" "16 8" "16 16" "16 32" "32 16" "32 32" "8 16"; do set -- $cfgs
NTTS_BENCH_POLL=$1 NTTS_BENCH_MIN_ADMIT=$2 timeout 300 python bench.py --mode continuous --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cont.json 2> gpurun_out/bench_cont.err; echo "poll=$1 min_admit=$2 rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_cont.json")); print(round(d["value"]), round(d["ms_per_step"],1), {k:round(v,1) for k,v in d["phase_ms"].items()})
PY
done
