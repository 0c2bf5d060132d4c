cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for ma in 32 64; do
  NTTS_BENCH_MIN_ADMIT=$ma NTTS_BENCH_CODEC_ROWS=256 NTTS_ATTN_EXP=3 timeout 500 python bench.py --mode continuous --batch 1024 --gang 1 --requests 16384 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > gpurun_out/widec_$ma.json 2> gpurun_out/widec_$ma.err
  echo "rc=$?"; tail -2 gpurun_out/widec_$ma.err
done
