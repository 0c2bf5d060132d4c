cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for a in 0 1 0 1; do
NTTS_PF_ROPE_VEC=$a timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_rv$a.json 2> gpurun_out/bench_rv$a.err; echo "rope_vec=$a rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_rv$a.json")); print(round(d["value"]), d["step_wall_ms"], {k:round(v,1) for k,v in d["phase_ms"].items() if not k.startswith("host")})
PY
done
timeout 900 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_variants.py tests/test_gpu_neutts_class.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
