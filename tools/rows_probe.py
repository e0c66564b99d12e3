"""Phase attribution of the row-reuse conv kernel (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib
for n in (32,):
    for cin, cout in ((64, 32), (128, 32), (160, 32), (192, 64)):
        for shape, nm in ((4, "narrow1"), (2, "narrow2"), (6, "rows2st"), (8, "rows64")):
            parts = []
            for m, lab in ((0, "full"), (1, "-stores"), (2, "-mfma"), (3, "-stores-mfma"), (7, "floor")):
                try:
                    parts.append(f"{lab} {_lib.op_conv2d_time(n, 41, 42, cin, cout, 3, 1, False, _lib.PREC_F16, 40, m, shape):6.1f}")
                except Exception as e:
                    parts.append(f"{lab} n/a")
            print(f"n={n} {cin:3d}->{cout:2d} {nm:8s} " + "  ".join(parts), flush=True)
