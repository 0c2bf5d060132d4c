#!/bin/bash
# temporary: A/B of the prompt-pass attention + counters of the resident kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT
SWEEP_KNOBS='[["NTTS_PF_RES_CAP",[512,0,512]]]' SWEEP_TIMEOUT=500 bash tools/gpu_round.sh sweep 2>&1 | grep -o 'NTTS_PF_RES_CAP.: [0-9]*}, .step_ms.: [0-9.]*, .last_prefill_chunk_ms.: [0-9.]*'
export NTTS_BENCH_PRIME=0 NTTS_NO_GRAPH=1
CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-codec --prefill 500 --decode 2 --no-pipeline --batch 64"
rm -rf $OUT/pfk2; timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $OUT/pfk2 -o k -- $CMD > /dev/null 2>&1
python tools/prof_summary.py $OUT/pfk2 | grep -i 'attn_prefill'
i=0
for set in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
  i=$((i+1)); rm -rf $OUT/pfp_$i
  timeout 120 rocprofv3 --kernel-trace --pmc $set -f csv -d $OUT/pfp_$i -o p -- $CMD > /dev/null 2>&1
  python tools/pmc_summary.py $OUT/pfp_$i | grep attn_prefill | awk '{print $1,$2,$3,$4}'
done
find $OUT/pfp_1 $OUT/pfp_2 $OUT/pfk2 -name '*.csv' -size +8M -delete
