cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf gpurun_out/pmc_mfma
NTTS_NO_GRAPH=1 timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 -f csv -d gpurun_out/pmc_mfma -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --prefill 605 --decode 40 > gpurun_out/pmc_mfma.json 2> gpurun_out/pmc_mfma.err; echo "rc=$?"
python tools/mfma_util_summary.py gpurun_out/pmc_mfma | tee gpurun_out/mfma_util_summary.txt
find gpurun_out/pmc_mfma -name '*.csv' -size +8M -delete
