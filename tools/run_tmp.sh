cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "silu" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_silu.json 2> gpurun_out/bench_silu.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/bench_silu.json")); print(round(d["value"]), d["step_wall_ms"], {k:round(v,1) for k,v in d["phase_ms"].items() if not k.startswith("host")})
PY
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
