cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_backbone.py -m gpu -q -rA -p no:cacheprovider 2>&1 | grep -v "^PASSED" | grep "fp32 run\|passed\|failed\|batch 1" | head
timeout 300 python tools/sweep_decode.py --batch 1 --mid 100 --steps 40 --knobs '[["NTTS_ATTN_NW_SMALL",[16,4,16]]]' > gpurun_out/sweep_b1.log 2>&1; grep -v "^\[sweep\] weights" gpurun_out/sweep_b1.log | cut -c1-330 | tail -6
echo "== NW=16 timeline"; TL_BATCH=1 NTTS_ATTN_NW_SMALL=16 timeout 120 python tools/attn_timeline.py 2>&1 | grep -v amdgpu.ids | tail -11
