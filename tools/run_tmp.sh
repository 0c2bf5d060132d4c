cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for a in 0 3; do
NTTS_ASYM=$a timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_asym$a.json 2> gpurun_out/bench_asym$a.err; echo "asym=$a rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_asym$a.json")); print(d["value"], d["ms_per_step"], d["step_wall_ms"], {k:round(v,1) for k,v in d["phase_ms"].items() if not k.startswith("host")})
PY
done
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
