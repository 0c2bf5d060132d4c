#!/bin/bash
# round 5, session h: big-GEMM tiles with more rows (DMA bytes per FLOP); tall tiles with one K slice per XCD pair (counters + step); batch-1 timelines
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 600 python tools/ubench_gemm.py --prefill-big > $OUT/ubench_prefill_big.txt 2>&1; cat $OUT/ubench_prefill_big.txt
timeout 300 python tools/sweep_gang.py --settings '[{}, {"NTTS_TALL_XCD_SPLIT": 0}, {}, {"NTTS_TALL_XCD_SPLIT": 0}, {"NTTS_TALL": 0, "NTTS_XCD_AFFINE": 7, "NTTS_QKV_WSTAT": 0}]' 2>&1 | grep '^{' > $OUT/sweep_gang_r05h.log; cut -c1-200 $OUT/sweep_gang_r05h.log
PMC_EXTRA_ENV="NTTS_TALL=3 NTTS_XCD_AFFINE=0 NTTS_QKV_WSTAT=1" bash tools/gpu_round.sh pmc > $OUT/pmc_gang_shape_final.log 2>&1; grep -h "qkv_rope\|8, 1, 2, 2, 3\|4, 2, 2, 1, 3" $OUT/pmc_FETCH_SIZE_summary.txt $OUT/pmc_WRITE_SIZE_summary.txt
for f in FETCH_SIZE WRITE_SIZE; do cp $OUT/pmc_${f}_summary.txt $OUT/pmc_${f}_summary_gang_shape_final.txt; done
python tools/pmc_to_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --prefill=621 --decode=8 > $OUT/pmc_traffic_gang_shape_final.json
TL_BATCH=1 timeout 200 python tools/gemv_timeline.py > $OUT/gemv_timeline_b1.txt 2>&1; tail -40 $OUT/gemv_timeline_b1.txt
TL_BATCH=1 timeout 200 python tools/attn_timeline.py > $OUT/attn_timeline_b1.txt 2>&1; tail -15 $OUT/attn_timeline_b1.txt
bash tools/gpu_round.sh b1 2>&1 | tail -4 | cut -c1-600
