cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "rc=$?"; tail -12 gpurun_out/bench_final.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_final.json")); print(round(d["value"]), d["ms_per_step"], d["step_wall_ms"], {k:round(v,1) for k,v in d["phase_ms"].items() if not k.startswith("host")}, d["decode_step"]["ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print([ (s["symbol"], s["share_pct"], s["avg_us"], s["agree"]) for s in d["roofline"]["rocprof"]["symbols"]])
PY
