"""conv_pw.hip against the generic split kernel on the detector's 1x1 layer shapes (2-frame batch: 122 items):
microseconds per launch for every shape that can run, and the implied HBM rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 122
LAYERS = [("model.2.cv1", 128, 64, 64), ("model.2.cv2", 128, 96, 128), ("model.4.cv1", 64, 128, 128), ("model.4.cv2", 64, 192, 256),
          ("model.16.cv2", 64, 192, 128), ("cv3.0.x.1", 64, 128, 128), ("cv2.0.2", 64, 64, 64), ("model.6.cv1", 32, 256, 256), ("model.6.cv2", 32, 384, 256),
          ("model.8.cv1", 16, 512, 512), ("model.9.cv2", 16, 1024, 512), ("model.22.cv1", 16, 768, 512), ("model.13.cv2", 32, 384, 256), ("model.10.cv1", 16, 512, 512)]
names = {0: "wide", 1: "wideH", 2: "narrow2", 3: "narrow2H", 4: "narrow1", 5: "narrow1H", 10: "pw1x4", 11: "pw2x2", 12: "pw2x1", 13: "pw1x4w", 14: "pw2x2w", 15: "pw2x1w", 16: "pw1x4s"}
PW = {10: (4, 40), 11: (2, 40), 12: (1, 40), 13: (4, 80), 14: (2, 80), 15: (1, 80), 16: (4, 0)}
for name, hw, cin, cout in LAYERS:
    mb = N * hw * hw * (cin + cout) * 4 / 1e6
    row = []
    for shape in (0, 2, 10, 11, 13, 14, 15, 16):
        if shape in PW:
            nt, fr = PW[shape]
            if (-(-cout // 32)) % nt or (cin % 128 if fr == 0 else nt * (cin // 16) > fr):
                continue
        if shape == 0 and cout < 96:
            continue
        try:
            us = _lib.op_conv2d_time(N, hw, hw, cin, cout, k=1, precision=_lib.PREC_F32X3, iters=20, shape=shape)
        except Exception as e:
            row.append(f"{names[shape]} ERR"); continue
        row.append(f"{names[shape]} {us:6.1f} ({mb / us:4.2f})")
    print(f"{name:12s} {hw:3d}^2 {cin:4d}->{cout:<4d} {mb:6.0f} MB | " + " | ".join(row) + "   us (TB/s)", flush=True)
