#!/bin/bash
# A/B of two builds of the library in ONE gpurun call (same box): A = the shipped libffp.so, B = $FFP_LIB_B (a second build, e.g.
# csrc/build/libffp_<variant>.so or a copy made before an edit). Each side is loaded through FFP_LIB — libffp.so is never overwritten.
# usage: FFP_LIB_B=path tools/ab_lib.sh [probe.py]
set -euo pipefail
: "${FFP_LIB_B:?set FFP_LIB_B to the second library}"
mkdir -p gpurun_out
for r in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab_A$r.json 2>/dev/null
  FFP_LIB=$FFP_LIB_B timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab_B$r.json 2>/dev/null
done
if [ -n "${1:-}" ]; then
  timeout -k 10 200 python "$1" > gpurun_out/ab_probe_A.txt 2>&1
  FFP_LIB=$FFP_LIB_B timeout -k 10 200 python "$1" > gpurun_out/ab_probe_B.txt 2>&1
fi
python - <<'PY'
import json
for n in ("A1", "B1", "A2", "B2"):
    d = json.load(open(f"gpurun_out/ab_{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], [(p["kernel"], p["ms"]) for p in d["conv_profile_last_step"][:5]])
PY
