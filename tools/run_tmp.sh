#!/bin/bash
# scratch command file for one-off gpurun calls:  /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/run_tmp.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; echo "bench rc=$?"
