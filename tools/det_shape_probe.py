"""Timing of the detector's 3x3 fp32-split layers per workgroup shape (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib
SH = ["wide", "wideH", "narrow2", "narrow2H", "narrow1", "narrow1H"]
cases = [("model.1 k3s2 32->64 @256", 61, 256, 256, 32, 64, 3, 2), ("model.3 k3s2 128->128 @128", 61, 128, 128, 128, 128, 3, 2),
         ("model.5 k3s2 256->256 @64", 61, 64, 64, 256, 256, 3, 2), ("model.7 k3s2 256->512 @32", 61, 32, 32, 256, 512, 3, 2),
         ("head cv2.0.0 k3s1 128->64 @64", 61, 64, 64, 128, 64, 3, 1), ("m.8 bottleneck k3s1 64->64 @16", 61, 16, 16, 64, 64, 3, 1),
         ("m.6 k3s1 64->64 @32", 61, 32, 32, 64, 64, 3, 1), ("m.2 k3s1 16->32 @128", 61, 128, 128, 16, 32, 3, 1)]
for name, n, h, w, cin, cout, k, s in cases:
    row = []
    for shape in range(-1, 6):
        try:
            t = _lib.op_conv2d_time(n, h, w, cin, cout, k, s, False, _lib.PREC_F32X3, 20, 0, shape)
            row.append(f"{'auto' if shape < 0 else SH[shape]} {t:7.1f}")
        except Exception:
            pass
    print(f"{name:34s} " + "  ".join(row), flush=True)
