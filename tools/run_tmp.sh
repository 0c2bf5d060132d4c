cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -q -rA -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "^PASSED" gpurun_out/pytest_gpu.log | tail -12; grep -n "fp8 slot\|fp8 free\|fp8 gemm" gpurun_out/pytest_gpu.log | head -20
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('gpurun_out/bench.json'));print('VALUE',d['value'],d['ms_per_step'],d['phase_ms'],d['decode_step'])"; tail -3 gpurun_out/bench.err
