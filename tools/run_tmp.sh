cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_backbone.py tests/test_gpu_variants.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/sweep_decode.py --batch 256 --knobs '[["NTTS_XCD_SPLIT",[0,1,0,1]]]' > gpurun_out/sweep_xcd.log 2>&1; grep -v "^\[sweep\] weights" gpurun_out/sweep_xcd.log | cut -c1-330 | tail -7
rm -rf gpurun_out/pmc_FETCH_SIZE; NTTS_NO_GRAPH=1 timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d gpurun_out/pmc_FETCH_SIZE -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-codec --prefill 605 --decode 8 --batch 256 > gpurun_out/pmc_bench.json 2> gpurun_out/pmc_FETCH_SIZE.err; python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE | grep -i "4, 1, 1\|4, 2, 2\|decode\|row"
find gpurun_out/pmc_FETCH_SIZE -name '*.csv' -size +8M -delete
