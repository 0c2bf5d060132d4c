#!/bin/bash
# scratch: one-off GPU experiment of the moment
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
echo "== continuous"; timeout 500 python bench.py --mode continuous --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_continuous.json 2> gpurun_out/bench_continuous.err; echo rc=$?; cut -c1-1500 gpurun_out/bench_continuous.json; tail -4 gpurun_out/bench_continuous.err
echo "== default"; timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_nocpu.json 2> gpurun_out/bench_nocpu.err; echo rc=$?; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_nocpu.json'))
print(d['value'], d['phase_ms'])
r=d['roofline']; print(r['kernel'], r['frac'], r['traffic'])
for s in r['rocprof']['symbols']: print(s['symbol'], s['share_pct'], s['avg_us'], round(s['live_avg_us'],2), s['agree'], round(s['frac_of_hbm_peak'],3))
PY
