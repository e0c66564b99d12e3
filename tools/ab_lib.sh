#!/bin/bash
# A/B two builds of libffp.so inside ONE gpurun call: <pkg>/libffp.so (A) vs <pkg>/libffp_B.so (B). Optional: a probe script run under both.
P=face-detection-with-yolov11-sahi-and-real-esrgan_amd
cp $P/libffp.so $P/libffp_A.so
for r in 1 2; do
  cp $P/libffp_A.so $P/libffp.so; timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab_A$r.json 2>/dev/null
  cp $P/libffp_B.so $P/libffp.so; timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab_B$r.json 2>/dev/null
done
if [ -n "$1" ]; then
  cp $P/libffp_A.so $P/libffp.so; timeout -k 10 200 python $1 > gpurun_out/ab_probe_A.txt 2>&1
  cp $P/libffp_B.so $P/libffp.so; timeout -k 10 200 python $1 > gpurun_out/ab_probe_B.txt 2>&1
fi
cp $P/libffp_A.so $P/libffp.so
python - <<'PY'
import json
for n in ("A1", "B1", "A2", "B2"):
    d = json.load(open(f"gpurun_out/ab_{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], [(p["kernel"], p["ms"]) for p in d["conv_profile_last_step"][:5]])
PY
