"""A/B of the stem-fused first conv: raw head outputs of YOLO11s (f32x3) on one synthetic frame, with the loader computing model.0
(default) or with the two-kernel path (FFP_NO_STEM_FUSE=1). Run once per setting with an output path, then `compare a.npz b.npz`."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    worst = 0.0
    for k in a.files:
        d = np.abs(a[k] - b[k])
        print(k, a[k].shape, "max |diff|", float(d.max()), "max |value|", float(np.abs(b[k]).max()))
        worst = max(worst, float(d.max()))
    print("worst", worst)
    sys.exit(0 if worst < 5e-3 else 1)

import ffp_amd  # noqa
from ffp_amd import _lib, synth
frame = synth.synthetic_frame(1080, 1920, seed=3)
tiles = [(0, 0, 512, 512), (700, 300, 1212, 812), (1408, 568, 1920, 1080), (0, 0, 1920, 1080), (100, 100, 400, 300), (5, 7, 261, 263)]
det = _lib.Detector(synth.yolo11_pose_weights("s"), arch="s", precision=_lib.PREC_F32X3)
out = {}
for order in (0, 1):
    for imgsz in (512, 256):
        o = det.forward_raw(frame, tiles, imgsz, chan_order=order)
        for i, x in enumerate(o):
            out[f"o{order}_s{imgsz}_t{i}"] = x
np.savez(sys.argv[1], **out)
print("saved", len(out))
