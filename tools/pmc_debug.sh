#!/bin/bash
# Why does the PMC pass stall?  Variants under rocprofv3 --pmc; on timeout SIGUSR1 makes faulthandler print the Python stack.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; OUT=gpurun_out; mkdir -p $OUT
run() { tag=$1; tmo=$2; args=$3; shift 3
  rm -rf $OUT/pmcdbg_$tag
  env "$@" NTTS_NO_GRAPH=1 NTTS_BENCH_PRIME_STEPS=2 timeout -k 5 -s USR1 $tmo rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmcdbg_$tag -o pmc -- \
    python -X faulthandler -c "
import faulthandler, signal, sys
faulthandler.register(signal.SIGUSR1, all_threads=True)
sys.argv = ['bench.py', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-roofline', '--no-codec', '--no-pipeline'] + '$args'.split()
import runpy; runpy.run_path('bench.py', run_name='__main__')
" > $OUT/pmcdbg_$tag.json 2> $OUT/pmcdbg_$tag.err; echo "$tag rc=$? after $SECONDS s"; grep -v "simple_timer\|^import\|^faulthandler\|^sys.argv\|^exec\|::" $OUT/pmcdbg_$tag.err | tail -12 | cut -c1-160; find $OUT/pmcdbg_$tag -name '*counter_collection.csv' | head -2; find $OUT/pmcdbg_$tag -name '*.csv' -size +8M -delete 2>/dev/null; }
run small 60 "--batch 64 --prefill 100 --decode 4" X=1
run mid 100 "--batch 256 --prefill 605 --decode 4" X=1
run nograph0 100 "--batch 256 --prefill 605 --decode 40" NTTS_NO_GRAPH=0
