cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_drv.json 2> gpurun_out/bench_drv.err; echo "rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_drv.json'))
print(round(d['value']), 'ms/step', round(d['ms_per_step'],1), d['config']['engine_slots'], d['step_wall_ms'])
print({k:(round(v,1) if isinstance(v,float) else v) for k,v in d['phase_ms'].items() if k in ('prefill','decode','codec')})
print(d['config']['workload'][:420])
r=d['roofline']; print({k:r[k] for k in ('kernel','frac','traffic','avg_launch_us','step_frac','step_ms')}); print({k:v for k,v in r['gang_step'].items() if k!='kernels_side_by_side' and k!='counters_note'})
c=d.get('continuous'); print({k:c[k] for k in c if k not in ('workload','phase_ms','steady_state_note')} if c else None)
print(d['cpu_baseline'])
PY
tail -5 gpurun_out/bench_drv.err
