cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -f gpurun_out/sweep.jsonl
timeout 800 python tools/sweep_decode.py --knobs '[["NTTS_PF_W",[0,1,3,7,2,0,3]]]' 2>&1 | grep -v "amdgpu.ids\|Perth" | tee gpurun_out/sweep_pfw.log | cut -c1-330
