cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out

timeout 300 python tools/ubench_gemm.py --wreg 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ubench_wreg.txt
