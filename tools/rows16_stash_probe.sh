#!/bin/bash
# Staging-placement experiment of conv_rows16_kernel: variants r16_stash1 / r16_stash2 (build.build_variant, built HERE before gpurun) against
# the shipped library, each loaded through FFP_LIB — libffp.so is never overwritten. See profiles/r03_rows16_stash_placement_probe.txt
set -euo pipefail
P=face-detection-with-yolov11-sahi-and-real-esrgan_amd
for v in 0 1 2; do
  if [ $v = 0 ]; then unset FFP_LIB; else export FFP_LIB=$PWD/$P/csrc/build/libffp_r16_stash$v.so; test -f "$FFP_LIB" || { echo "missing $FFP_LIB"; exit 1; }; fi
  echo "== stash variant $v"
  timeout -k 10 120 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import ffp_amd
from ffp_amd import _lib
for n, cin, cout in ((2048, 128, 32), (2048, 64, 32), (2048, 192, 64), (4700, 160, 32)):
    flops = 2.0 * cin * cout * 9 * 256 * n
    us = min(_lib.op_conv2d_time(n, 16, 16, cin, cout, 3, 1, False, _lib.PREC_F16, 40, 0, 9) for _ in range(3))
    print(f"tiles={n} {cin}->{cout}: {us:.1f} us  {flops / us / 1e6:.0f} TF/s", flush=True)
PY
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_us'])"
done
