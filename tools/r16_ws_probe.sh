for ws in 64 32 48; do
  for l in 2 1; do
    FFP_ROWS16_WS=$ws timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 40 --lanes $l > gpurun_out/ws_${ws}_$l.json 2>/dev/null || exit 1
    python3 -c "
import json; d=json.load(open('gpurun_out/ws_${ws}_$l.json')); print('ws', $ws, 'lanes', $l, d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['sr_ms_last_call'], d['stage_ms_last_call']['total'])"
  done
done
