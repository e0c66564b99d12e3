"""How the SR body convs and the whole enhancer scale with the number of crops per ragged batch (tuning aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib, synth
SH = ["wide", "wideH", "narrow2", "narrow2H", "narrow1", "narrow1H", "rows2st", "rows3st"]
for n in (32, 64, 128):
    for cin, cout in ((64, 32), (128, 32), (192, 64)):
        row = []
        for shape in (-1, 2, 4, 6, 7):
            try:
                t = _lib.op_conv2d_time(n, 41, 42, cin, cout, 3, 1, False, _lib.PREC_F16, 40, 0, shape)
                row.append(f"{'auto' if shape < 0 else SH[shape]} {t:6.1f}")
            except Exception:
                row.append(f"{SH[shape]}   n/a")
        fl = 2.0 * cin * cout * 9 * n * 41 * 42
        print(f"n={n:3d} {cin:3d}->{cout:2d}  " + "  ".join(row), flush=True)
W = synth.rrdbnet_weights(4)
enh = _lib.Enhancer(W, scale=4, half=True)
rng = np.random.default_rng(0)
for n in (8, 16, 32, 64, 128):
    crops = [rng.integers(0, 255, (int(rng.integers(32, 52)), int(rng.integers(32, 52)), 3), dtype=np.uint8) for _ in range(n)]
    px = sum(c.shape[0] * c.shape[1] for c in crops)
    for _ in range(3):
        enh.enhance_batch(crops)
    ms = []
    for _ in range(5):
        enh.enhance_batch(crops)
        ms.append(enh.last_ms())
    print(f"enhance_batch n={n:3d} px={px:7d} device ms {min(ms):7.3f}  us/kpx {1e3 * min(ms) / (px / 1e3):6.2f}", flush=True)
