"""Phase attribution of the generic split kernel on the detector's 3x3 layers (2-frame batch): time with stores / MFMAs / chunk refetch /
LDS stash switched off (results wrong by construction), per workgroup shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib

N = 122
LAYERS = [("model.3", 128, 128, 128, 2), ("model.5", 64, 256, 256, 2), ("model.1", 256, 32, 64, 2), ("cv2.0.0", 64, 128, 64, 1), ("m.0.cv1", 128, 32, 16, 1), ("cv2.1.0", 32, 256, 64, 1)]
names = {0: "wide", 1: "wideH", 2: "narrow2", 3: "narrow2H", 4: "narrow1", 5: "narrow1H"}
for name, hw, cin, cout, s in LAYERS:
    for shape in (0, 1, 2, 3, 4):
        row = []
        for dbg, tag in ((0, "full"), (1, "-store"), (2, "-mfma"), (8, "-stash"), (4, "-refetch"), (3, "-store-mfma"), (11, "-store-mfma-stash")):
            try:
                us = _lib.op_conv2d_time(N, hw, hw, cin, cout, k=3, stride=s, precision=_lib.PREC_F32X3, iters=10, dbg=dbg, shape=shape)
            except Exception as e:
                row = None
                break
            row.append(f"{tag} {us:6.1f}")
        if row:
            print(f"{name:8s} {hw:3d}^2 {cin:3d}->{cout:<3d} s{s} {names[shape]:9s} " + " | ".join(row), flush=True)
