# Diagnostic: build libffp_stamp.so first (conv_pw.hip with -DFFP_PW_STAMP=1, linked with the other objects) — see profiles/r03_pw_stage_stamps.txt
P=face-detection-with-yolov11-sahi-and-real-esrgan_amd
cp $P/libffp.so /tmp/libffp_keep.so; cp $P/libffp_stamp.so $P/libffp.so
timeout -k 10 120 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import ffp_amd
from ffp_amd import _lib
for n, hw, cin, cout in ((305, 16, 1024, 512), (305, 16, 512, 512)):
    us = _lib.op_conv2d_time(n, hw, hw, cin, cout, 1, 1, False, _lib.PREC_F32X3, 2, 0, 16)
    print(f"== n={n} {hw}x{hw} {cin}->{cout} pw1x4s: {us:.0f} us", flush=True)
PY
cp /tmp/libffp_keep.so $P/libffp.so
