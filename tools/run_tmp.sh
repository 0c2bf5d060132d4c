cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -f gpurun_out/sweep.jsonl
timeout 800 python tools/sweep_decode.py --knobs '[["NTTS_ATTN_VAR",[1,17,23,1,17]]]' 2>&1 | grep -v "amdgpu.ids\|Perth" | tee gpurun_out/sweep_attn_nw8.log | cut -c1-200
