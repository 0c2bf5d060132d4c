cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for ps in 2 249; do
  NTTS_BENCH_PRIME_STEPS=$ps timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_w1.json 2> gpurun_out/bench_w1.err; echo "prime_steps=$ps rc=$?"; tail -2 gpurun_out/bench_w1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_w1.json")); print(round(d["value"]), d["step_wall_ms"], d["step_host_wall_ms"])
PY
done
