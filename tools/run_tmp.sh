#!/bin/bash
# scratch: one-off GPU experiment of the moment
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 500 python tools/sweep_decode.py --knobs '[["NTTS_GU_TILE",[1,3,4,1]]]' > gpurun_out/sweep_gu.log 2>&1; grep -v "^\[sweep\] weights" gpurun_out/sweep_gu.log | cut -c1-420 | tail -7
