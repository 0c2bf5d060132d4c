cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_neutts_class.py tests/test_gpu_encoder.py -q -x -m gpu 2>&1 | tail -2
for b in 32 128 512; do
python bench.py --config nano-fp8 --mode stream --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/stream_$b.json 2> gpurun_out/stream_$b.err
python - <<PY
import json
d=json.loads(open("gpurun_out/stream_$b.json").read().strip().splitlines()[-1]); ph=d.get("phase_ms",{})
print("stream $b", round(d["value"]), round(d["ms_per_step"],1), {k: v for k,v in d.items() if "ttfa" in k.lower() or "chunk" in k.lower()}, {k: v for k,v in ph.items() if "ttfa" in k.lower() or "chunk" in k.lower() or "first" in k.lower()})
PY
done
python bench.py --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_b1.json").read().strip().splitlines()[-1]); ph=d.get("phase_ms",{})
print("b1", round(d["value"]), round(d["ms_per_step"],1), {k: round(v,2) for k,v in ph.items() if k in ("prefill","decode","codec")})
PY
