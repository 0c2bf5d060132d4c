#!/bin/bash
# round 5, session g: QKV W-stationary mapping under the gang (step time + counters); stream mode again (burst = what the nearest window needs); new tests
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 300 python tools/sweep_gang.py --kernels --settings '[{"NTTS_TALL": 3, "NTTS_XCD_AFFINE": 0}, {"NTTS_TALL": 3, "NTTS_XCD_AFFINE": 0, "NTTS_QKV_WSTAT": 1}, {"NTTS_TALL": 3, "NTTS_XCD_AFFINE": 0}, {"NTTS_TALL": 3, "NTTS_XCD_AFFINE": 0, "NTTS_QKV_WSTAT": 1}, {}]' 2>&1 | grep '^{' > $OUT/sweep_gang_r05g.log; cat $OUT/sweep_gang_r05g.log | cut -c1-420
PMC_EXTRA_ENV="NTTS_TALL=3 NTTS_XCD_AFFINE=0 NTTS_QKV_WSTAT=1" bash tools/gpu_round.sh pmc > $OUT/pmc_gang_shape_wstat.log 2>&1; grep "qkv_rope\|8, 1, 2, 2, 3" $OUT/pmc_gang_shape_wstat.log
for f in FETCH_SIZE WRITE_SIZE; do cp $OUT/pmc_${f}_summary.txt $OUT/pmc_${f}_summary_gang_shape_wstat.txt; done
python tools/pmc_to_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --prefill=621 --decode=8 > $OUT/pmc_traffic_gang_shape_wstat.json
timeout 600 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity_matrix.py tests/test_gpu_neutts_class.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
for spec in "512 4 64" "512 4 128" "512 2 128" "32 4 64"; do set -- $spec
  timeout 400 python bench.py --config nano-fp8 --mode stream --batch $1 --gang $2 --stream-admit $3 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_nano-fp8_stream_b$1_g$2_a$3.json 2> $OUT/bench_nano-fp8_stream_b$1_g$2_a$3.err; echo "stream b=$1 gang=$2 admit=$3 rc=$?"; python -c "
import json;r=json.loads(open('$OUT/bench_nano-fp8_stream_b$1_g$2_a$3.json').read().strip().splitlines()[-1]);s=r['stream'];print(round(r['value']), {k: (round(v,1) if isinstance(v,float) else v) for k,v in s.items() if k.startswith('ttfa') or k in ('total_ms','mean_chunk_period_ms')})"; done
