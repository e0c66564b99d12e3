#!/bin/bash
# A/B two bench.py argument sets inside ONE gpurun call (boxes differ by ~20 %): tools/ab_args.sh "<args A>" "<args B>"
for r in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $1 > gpurun_out/ab_A$r.json 2>/dev/null
  timeout -k 10 300 python bench.py --no-cpu-baseline $2 > gpurun_out/ab_B$r.json 2>/dev/null
done
python - <<'PY'
import json
for n in ("A1", "B1", "A2", "B2"):
    d = json.load(open(f"gpurun_out/ab_{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], [(p["kernel"], p["ms"]) for p in d["conv_profile_last_step"][:4]])
PY
