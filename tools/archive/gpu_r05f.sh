#!/bin/bash
# round 5, session f: codec high precision + class-level stream tests; PMC passes on the GANG's decode shape; opt-in speech-range head line; stream mode on the gang vs one engine
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_neutts_class.py tests/test_gpu_variants.py -m gpu -q -s -rA -p no:cacheprovider > $OUT/pytest_codec_class.log 2>&1; echo "pytest rc=$?"; grep -a "relative rms\|18x\|codec pass\|high precision\|passed\|failed\|FAILED" $OUT/pytest_codec_class.log | tail -20
PMC_EXTRA_ENV="NTTS_TALL=3 NTTS_XCD_AFFINE=0" bash tools/gpu_round.sh pmc > $OUT/pmc_gang_shape.log 2>&1; tail -30 $OUT/pmc_gang_shape.log
for f in FETCH_SIZE WRITE_SIZE; do cp $OUT/pmc_${f}_summary.txt $OUT/pmc_${f}_summary_gang_shape.txt; done
python tools/pmc_to_json.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE --prefill=621 --decode=8 > $OUT/pmc_traffic_gang_shape.json
timeout 400 python bench.py --speech-range-head --steps 12 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/bench_speech_range_head.json 2> $OUT/bench_speech_range_head.err; echo "srh rc=$?"; cut -c1-400 $OUT/bench_speech_range_head.json; tail -3 $OUT/bench_speech_range_head.err
timeout 400 python bench.py --config nano-fp8 --mode stream --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_nano-fp8_stream.json 2> $OUT/bench_nano-fp8_stream.err; echo "stream gang rc=$?"; python -c "
import json;r=json.loads(open('$OUT/bench_nano-fp8_stream.json').read().strip().splitlines()[-1]);print(r['value'], r['stream'])"; tail -3 $OUT/bench_nano-fp8_stream.err
timeout 400 python bench.py --config nano-fp8 --mode stream --gang 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_nano-fp8_stream_one_engine.json 2> $OUT/bench_nano-fp8_stream_one_engine.err; echo "stream one rc=$?"; python -c "
import json;r=json.loads(open('$OUT/bench_nano-fp8_stream_one_engine.json').read().strip().splitlines()[-1]);print(r['value'], r['stream'])"
timeout 300 python bench.py --config nano-fp8 --mode stream --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_nano-fp8_stream_batch32.json 2> $OUT/bench_nano-fp8_stream_batch32.err; echo "stream b32 rc=$?"; python -c "
import json;r=json.loads(open('$OUT/bench_nano-fp8_stream_batch32.json').read().strip().splitlines()[-1]);print(r['value'], r['stream'])"
