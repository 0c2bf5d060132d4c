#!/bin/bash
# temporary: counters of the prompt-pass attention kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 NTTS_BENCH_PRIME=0 NTTS_NO_GRAPH=1
OUT=gpurun_out; mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-codec --prefill 500 --decode 2 --no-pipeline --batch 64"
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/pmc_avail_sq.txt
for cap in 512 0; do
  export NTTS_PF_RES_CAP=$cap
  rm -rf $OUT/pfk_$cap; timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $OUT/pfk_$cap -o k -- $CMD > /dev/null 2> $OUT/pfk_$cap.err; echo "trace cap=$cap rc=$?"
  python tools/prof_summary.py $OUT/pfk_$cap 2>/dev/null | grep -i "attn_prefill\|rope_kv" | head -4
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
    i=$((i+1)); rm -rf $OUT/pfp_${cap}_$i
    timeout 120 rocprofv3 --kernel-trace --pmc $set -f csv -d $OUT/pfp_${cap}_$i -o p -- $CMD > /dev/null 2> $OUT/pfp_${cap}_$i.err; echo "pmc cap=$cap set=$i rc=$?"
    python tools/pmc_summary.py $OUT/pfp_${cap}_$i 2>/dev/null | grep "attn_prefill" | awk '{print $1, $2, $3, $4, $5}' | cut -c1-160
    find $OUT/pfp_${cap}_$i -name '*.csv' -size +8M -delete
  done
  find $OUT/pfk_$cap -name '*.csv' -size +8M -delete
done
