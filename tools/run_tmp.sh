cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for a in 0 3 0 3; do
NTTS_ASYM=$a timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_asym$a.json 2> gpurun_out/bench_asym$a.err; echo "asym=$a rc=$?"
grep "lm_head\|gate_up" gpurun_out/bench_asym$a.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_asym$a.json")); print(d["decode_step"]["ms"], {k:round(v,1) for k,v in d["phase_ms"].items() if not k.startswith("host")})
PY
done
