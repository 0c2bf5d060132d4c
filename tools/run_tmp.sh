cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
for nw in 4 16; do echo "== B=1 timeline NW=$nw"; TL_BATCH=1 NTTS_ATTN_NW_SMALL=$nw timeout 120 python tools/attn_timeline.py 2>&1 | grep -v amdgpu.ids | tail -12; done
timeout 300 python tools/sweep_decode.py --batch 1 --mid 100 --steps 40 --knobs '[["NTTS_ATTN_NW_SMALL",[4]]]' > gpurun_out/sweep_b1.log 2>&1; grep -v "^\[sweep\] weights" gpurun_out/sweep_b1.log | cut -c1-400 | tail -4
