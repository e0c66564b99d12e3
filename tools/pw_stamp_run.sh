#!/bin/bash
# Diagnostic: conv_pw.hip with -DFFP_PW_STAMP=1 (python -m ffp_amd.build --variant pw_stamp, built HERE before gpurun), loaded through
# FFP_LIB — the shipped libffp.so is never overwritten. See profiles/r03_pw_stage_stamps.txt
set -euo pipefail
P=face-detection-with-yolov11-sahi-and-real-esrgan_amd
export FFP_LIB=$PWD/$P/csrc/build/libffp_pw_stamp.so
test -f "$FFP_LIB" || { echo "build it first: python -c 'import ffp_amd.build as b; b.build_variant(\"pw_stamp\")'"; exit 1; }
timeout -k 10 120 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import ffp_amd
from ffp_amd import _lib
for n, hw, cin, cout in ((305, 16, 1024, 512), (305, 16, 512, 512)):
    us = _lib.op_conv2d_time(n, hw, hw, cin, cout, 1, 1, False, _lib.PREC_F32X3, 2, 0, 16)
    print(f"== n={n} {hw}x{hw} {cin}->{cout} pw1x4s: {us:.0f} us", flush=True)
PY
