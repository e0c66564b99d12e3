#!/bin/bash
# A/B a bench.py run with and without an environment switch inside ONE gpurun call (boxes differ by ~20 %).
# usage: tools/ab_env.sh VAR=VALUE [bench args...]
SW=$1; shift
for r in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_base$r.json 2>/dev/null
  env $SW timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_sw$r.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("base1", "sw1", "base2", "sw2"):
    d = json.load(open(f"gpurun_out/ab_{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["stage_ms_last_call"], [(p["kernel"], p["ms"]) for p in d["conv_profile_last_step"][:4]])
PY
