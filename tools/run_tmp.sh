cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
PP_LOG=1 PP_BATCHES=4 PP_CUS=96 PP_LAYOUT=0 timeout 400 python tools/pipeline_probe.py 2>&1 | grep -v "amdgpu.ids\|Perth"
