export TMPDIR=/tmp
timeout 300 python tools/x_fused_timeline.py 2>&1 | grep -v amdgpu.ids
for cfg in "0 1" "1 1" "1 0" "0 1"; do set -- $cfg
  timeout 300 python tools/sweep_decode.py --knobs "[]" --base "{\"NTTS_FUSED_QKV_ATTN\":$1,\"NTTS_X_CUBUSY\":$2}" 2>/dev/null | grep step_ms | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('fused $1 cubusy $2:', r['step_ms'])"
done
