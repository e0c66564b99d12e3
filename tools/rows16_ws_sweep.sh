for ws in 64 32 40 48; do
  FFP_ROWS16_WS=$ws timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ws', $ws, r['value'], r['ms_per_step'], r['stage_ms_last_call'], r['roofline']['frac'], r.get('per_rank',[{}])[0].get('stage_ms_per_step'))"
done
