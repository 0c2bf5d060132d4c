cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -f gpurun_out/sweep.jsonl
timeout 600 python tools/sweep_decode.py --knobs '[["NTTS_ATTN_VAR",[1,9,1,9]]]' 2>&1 | grep -v "amdgpu.ids\|Perth" | tee gpurun_out/sweep_attn_bt.log | cut -c1-330
