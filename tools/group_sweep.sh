# detection group / SR batch sizes with two lanes (the headline's loop), 60 steps each
for cfg in "5 10" "4 8" "6 12" "5 5" "3 9" "5 15" "10 10"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --det-batch-frames $1 --sr-batch-frames $2 --lanes 2 --no-secondary --no-cpu-baseline --steps 60 --warmup $1 > gpurun_out/gs_$1_$2.json 2>/dev/null || { echo "DB $1 SB $2 failed"; continue; }
  python3 -c "
import json; d=json.load(open('gpurun_out/gs_$1_$2.json')); print('DB', $1, 'SB', $2, d['value'], d['latency_ms_rank0'], d['config']['hbm_bytes_peak']/1e9)"
done
