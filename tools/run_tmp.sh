cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 300 python bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1 500/250', round(d['value'],1), d['step_wall_ms'])"
timeout 300 python bench.py --batch 1 --prefill 1200 --decode 600 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1 1200/600 adaptive', round(d['value'],1), d['step_wall_ms'], d['phase_ms']['decode'])"
NTTS_ATTN_SPLIT=0 timeout 300 python bench.py --batch 1 --prefill 1200 --decode 600 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b1 1200/600 nosplit ', round(d['value'],1), d['step_wall_ms'], d['phase_ms']['decode'])"
