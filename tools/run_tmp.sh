#!/bin/bash
# scratch: one-off GPU experiment of the moment
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_neutts_class.py -m gpu -q -s -x -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pytest_encoder.log | tail -25
echo "== probe"; timeout 500 python tools/encode_probe.py --secs 3,10,30 --cpu 2>&1 | grep -v amdgpu.ids | tee gpurun_out/encode_probe.log | tail -8
echo "== rocprof"; rm -rf gpurun_out/prof_enc; timeout 400 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_enc -o enc -- python tools/encode_probe.py --secs 10 > /dev/null 2> gpurun_out/prof_enc.err; python tools/prof_summary.py gpurun_out/prof_enc 2>&1 | head -16 | tee gpurun_out/prof_enc_summary.txt; find gpurun_out/prof_enc -name '*kernel_trace.csv' -size +20M -delete
