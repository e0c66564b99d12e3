"""Phase attribution of the generic split kernel on the detector's deep 1x1 layers (2-frame batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib

N = 122
LAYERS = [("model.4.cv2", 64, 192, 256), ("model.9.cv2", 16, 1024, 512), ("model.8.cv1", 16, 512, 512), ("model.22.cv1", 16, 768, 512), ("model.6.cv2", 32, 384, 256), ("model.13.cv1", 32, 768, 256)]
names = {0: "wide", 1: "wideH", 2: "narrow2", 16: "pw1x4s"}
for name, hw, cin, cout in LAYERS:
    for shape in (0, 1, 2, 16):
        row = []
        for dbg, tag in ((0, "full"), (1, "-store"), (2, "-mfma"), (8, "-stash"), (4, "-refetch"), (3, "-store-mfma"), (11, "-store-mfma-stash")):
            if shape == 16 and dbg:
                continue
            try:
                us = _lib.op_conv2d_time(N, hw, hw, cin, cout, k=1, stride=1, precision=_lib.PREC_F32X3, iters=20, dbg=dbg, shape=shape)
            except Exception:
                row = None
                break
            row.append(f"{tag} {us:6.1f}")
        if row:
            fl = N * hw * hw * cin * cout * 2 * 3 / 1e6
            print(f"{name:12s} {hw:3d}^2 {cin:4d}->{cout:<4d} {names[shape]:8s} " + " | ".join(row) + f"   ({fl / float(row[0].split()[1]):5.0f} TF/s fp16-MFMA)", flush=True)
