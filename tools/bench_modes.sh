#!/bin/bash
# The other bench lines of a round in one gpurun call (gpu_round.sh covers the default line, tests, rocprofv3 and PMC):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/bench_modes.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
run() { name=$1; shift; timeout 500 python bench.py "$@" --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name rc=$?"; cut -c1-300 $OUT/bench_$name.json; }
run b1 --batch 1 --steps 6 --warmup 2
run sample --sample --steps 6 --warmup 1 --no-roofline
run continuous --mode continuous --requests 16384 --steps 1 --warmup 1 --no-roofline
run continuous_one_engine --mode continuous --gang 1 --requests 4096 --steps 1 --warmup 1 --no-roofline
run nano-fp8 --config nano-fp8 --steps 3 --warmup 1
run nano-fp8_stream --config nano-fp8 --mode stream --steps 2 --warmup 1 --no-roofline
run nano-fp8_stream_batch32 --config nano-fp8 --mode stream --batch 32 --steps 2 --warmup 1 --no-roofline
run no_pipeline --no-pipeline --steps 6 --warmup 1 --no-roofline
