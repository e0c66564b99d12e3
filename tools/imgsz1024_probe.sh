# image_size=1024 (the reference wrapper's default): detection group size and lanes
for cfg in "2 2" "2 1" "3 2" "5 1" "5 2"; do
  set -- $cfg
  FFP_DET_PLAN_GIB=100 timeout -k 10 300 python bench.py --imgsz 1024 --det-batch-frames $1 --lanes $2 --no-secondary --no-cpu-baseline --steps 20 > gpurun_out/i1024_$1_$2.json 2>/dev/null || { echo "DB $1 lanes $2 failed"; continue; }
  python3 -c "
import json; d=json.load(open('gpurun_out/i1024_$1_$2.json')); print('DB', $1, 'lanes', $2, d['value'], d['stage_ms_last_call']['total'], d['sr_ms_last_call'], d['config']['hbm_bytes_peak']/1e9)"
done
