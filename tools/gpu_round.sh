#!/bin/bash
# One gpurun call = everything we want from a GPU box, each leg under its own timeout, logs in gpurun_out/.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_round.sh [legs...]'
# legs: smoke tests bench b1 nano prof profcont pmc pmcgang mfma sweep   (default: smoke tests bench prof)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT
LEGS="${*:-smoke tests bench prof}"
echo "== legs: $LEGS"; rocm-smi --showproductname 2>/dev/null | head -8; nproc
for leg in $LEGS; do
  case $leg in
    smoke) timeout 420 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 $OUT/smoke.log;;
    tests) timeout ${TESTS_TIMEOUT:-900} python -m pytest tests -m gpu -q -rA --durations=10 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -45 $OUT/pytest_gpu.log;;
    bench) timeout 900 python bench.py --gpus 1 --steps ${BENCH_STEPS:-20} --warmup ${BENCH_WARMUP:-5} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -25 $OUT/bench.err;;
    b1)    timeout 300 python bench.py --batch 1 --steps ${BENCH_STEPS:-3} --warmup 1 --no-cpu-baseline > $OUT/bench_b1.json 2> $OUT/bench_b1.err; echo "bench b1 rc=$?"; cat $OUT/bench_b1.json | cut -c1-1500; tail -12 $OUT/bench_b1.err;;
    prof)  rm -rf $OUT/prof; NTTS_BENCH_PRIME_STEPS=2 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof -o bench -- python bench.py --steps ${PROF_STEPS:-10} --warmup 0 --engine-slots ${PMC_BATCH:-640} --no-cpu-baseline --no-roofline > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc=$?"
           python tools/prof_summary.py $OUT/prof > $OUT/prof_summary.txt 2>&1; head -40 $OUT/prof_summary.txt; cat $OUT/prof_bench.json; tail -3 $OUT/prof.err
           find $OUT/prof -name '*kernel_trace.csv' -size +20M -delete;;
    pmc)   # HBM traffic counters, each in its own pass (FETCH_SIZE and WRITE_SIZE do not fit one pass); hipGraph replay off
           # (counter collection crashed rocprofv3 under graph replay); --kernel-trace only, as the pool rules require
           # (a counter pass takes ~10 s when it runs; on some attempts the profiled process stalls right after start-up -- seen with either
           #  counter, never without the profiler -- so each pass gets a short timeout and up to three attempts)
           for ctr in FETCH_SIZE WRITE_SIZE; do export NTTS_BENCH_PRIME=0   # (no warm-up: its one-slot decode steps would dilute the per-launch means; 8 decode steps per pass -- a pass with 40 hangs under the profiler since round 3, with 4-8 it takes 8-16 s)
             for attempt in 1 2 3; do
               rm -rf $OUT/pmc_$ctr; env ${PMC_EXTRA_ENV:-} NTTS_NO_GRAPH=1 timeout 90 rocprofv3 --kernel-trace --pmc $ctr -f csv -d $OUT/pmc_$ctr -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-codec --prefill ${PMC_PREFILL:-621} --decode ${PMC_DECODE:-8} --no-pipeline --batch ${PMC_BATCH:-640} > $OUT/pmc_bench.json 2> $OUT/pmc_$ctr.err; rc=$?; echo "pmc $ctr attempt $attempt rc=$rc"
               [ $rc -eq 0 ] && break
             done
             python tools/pmc_summary.py $OUT/pmc_$ctr > $OUT/pmc_${ctr}_summary.txt 2>&1; head -14 $OUT/pmc_${ctr}_summary.txt
           done
           # the record bench.py quotes as roofline.traffic (one engine of PMC_BATCH slots = the engines of the default run: the driver's --steps 20 -> 640)
           python tools/pmc_to_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --prefill=${PMC_PREFILL:-621} --decode=${PMC_DECODE:-8} --batch=${PMC_BATCH:-640} > $OUT/pmc_traffic.json 2> $OUT/pmc_to_json.err; head -c 600 $OUT/pmc_traffic.json; echo
           find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE -name '*.csv' -size +8M -delete;;
    pmcgang) # HBM traffic counters over GANG steps -- four 256-slot engines on one arena, step graphs off, the chains' launches enqueued
           # alternately on their lanes as in the timed region (one pass per counter; --kernel-trace only, as the pool rules require).  Whether the chains'
           # kernels really overlapped under the profiler is read from the kernel trace of the same pass (tools/pmc_gang_summary.py).
           for ctr in FETCH_SIZE WRITE_SIZE; do export NTTS_BENCH_PRIME=0
             for attempt in 1 2 3; do
               rm -rf $OUT/pmcg_$ctr; NTTS_NO_GRAPH=1 timeout 150 rocprofv3 --kernel-trace --pmc $ctr -f csv -d $OUT/pmcg_$ctr -o pmc -- python bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-roofline --no-codec --prefill ${PMC_PREFILL:-621} --decode ${PMC_DECODE:-8} --batch 256 > $OUT/pmcg_bench.json 2> $OUT/pmcg_$ctr.err; rc=$?; echo "pmcgang $ctr attempt $attempt rc=$rc"
               [ $rc -eq 0 ] && break
             done
             python tools/pmc_gang_summary.py $OUT/pmcg_$ctr $ctr > $OUT/pmcg_${ctr}_summary.txt 2>&1; head -30 $OUT/pmcg_${ctr}_summary.txt; find $OUT/pmcg_$ctr -name '*.csv' -size +8M -delete
           done;;
    mfma)  # matrix-core utilisation per kernel (north_star: "MFMA utilisation against gfx950 peak"): SQ busy cycles / GRBM cycles / MFMA op counts
           # in ONE counter pass (8 SQ slots, 2 GRBM), graph replay off, one engine, 8 decode steps (longer profiled passes hang since round 3)
           for attempt in 1 2 3; do
             rm -rf $OUT/pmc_mfma; NTTS_NO_GRAPH=1 NTTS_BENCH_PRIME=0 timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 -f csv -d $OUT/pmc_mfma -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-pipeline --prefill 605 --decode 8 --batch ${PMC_BATCH:-640} > $OUT/pmc_mfma_bench.json 2> $OUT/pmc_mfma.err; rc=$?; echo "mfma attempt $attempt rc=$rc"
             [ $rc -eq 0 ] && break
           done
           { echo "# NTTS_NO_GRAPH=1 NTTS_BENCH_PRIME=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-pipeline --prefill 605 --decode 8 --batch ${PMC_BATCH:-640}"
             echo "# python tools/mfma_util_summary.py <dir>   (formula and calibration in the tool's header; counter passes run at a lower clock than the bench)"
             python tools/mfma_util_summary.py $OUT/pmc_mfma; } > $OUT/mfma_util.txt 2>&1; head -16 $OUT/mfma_util.txt; find $OUT/pmc_mfma -name '*.csv' -size +8M -delete;;
    sweep)  # SWEEP_KNOBS='[["NTTS_ATTN_DEPTH",[2]]]'
           timeout ${SWEEP_TIMEOUT:-400} python tools/sweep_decode.py --knobs "${SWEEP_KNOBS:-[[\"NTTS_ATTN_DEPTH\",[2]]]}" > $OUT/sweep.log 2>&1; echo "sweep rc=$?"; grep -v '^\[sweep\] weights' $OUT/sweep.log | tail -12;;
    profcont) # where the ragged scheduler's time goes against the static schedule's: kernel traces of both, classed by phase over the middle of the run
           for mode in static continuous; do
             rm -rf $OUT/profcont_$mode
             if [ $mode = static ]; then args="--steps 12 --warmup 0"; else args="--mode continuous --requests ${PROFCONT_REQUESTS:-4096} --steps 1 --warmup 0 --gang 4"; fi
             NTTS_BENCH_PRIME_STEPS=2 timeout 600 rocprofv3 --kernel-trace -f csv -d $OUT/profcont_$mode -o t -- python bench.py $args --no-cpu-baseline --no-roofline > $OUT/profcont_$mode.json 2> $OUT/profcont_$mode.err; echo "profcont $mode rc=$?"
             { echo "# rocprofv3 --kernel-trace -- python bench.py $args --no-cpu-baseline --no-roofline"; python tools/trace_classes.py $OUT/profcont_$mode 0.35 0.85; } > $OUT/profcont_${mode}_classes.txt 2>&1
             cat $OUT/profcont_${mode}_classes.txt | cut -c1-170; cut -c1-600 $OUT/profcont_$mode.json; tail -3 $OUT/profcont_$mode.err
             find $OUT/profcont_$mode -name '*.csv' -size +1M -delete
           done;;
    nano)  for cfg in nano-fp8 nano-bf16; do timeout 300 python bench.py --config $cfg --steps ${NANO_STEPS:-20} --warmup 1 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; echo "bench $cfg rc=$?"; cut -c1-1200 $OUT/bench_$cfg.json; tail -12 $OUT/bench_$cfg.err; done;;
    *) echo "unknown leg $leg";;
  esac
done
