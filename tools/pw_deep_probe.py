"""Deep 1x1 layers (16^2 / 32^2 levels, K >= 384) per workgroup shape: microseconds per launch for a 2-frame and a 5-frame group."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib
NAMES = {0: "g0", 1: "g1", 2: "g2", 3: "g3", 4: "g4", 5: "g5", 12: "pw2x1", 15: "pw2x1w", 14: "pw2x2w", 16: "pw1x4s", 22: "pw2x2s", 25: "pw1x8s"}
for n in (122, 305):
    for hw, cin, cout in ((16, 1024, 512), (16, 512, 512), (16, 768, 512), (32, 768, 256), (32, 384, 256), (16, 256, 512)):
        flops = 2.0 * cin * cout * hw * hw * n * 3
        row = []
        for shape in (0, 1, 12, 14, 15, 16, 22, 25):
            try:
                us = min(_lib.op_conv2d_time(n, hw, hw, cin, cout, 1, 1, False, _lib.PREC_F32X3, 20, 0, shape) for _ in range(2))
                row.append(f"{NAMES[shape]}:{us:.0f}")
            except Exception:
                pass
        best = min(float(r.split(":")[1]) for r in row)
        print(f"n={n} {hw}x{hw} {cin}->{cout}: " + " ".join(row) + f"   best {best:.0f} us = {flops / (best * 1e-6) / 2.5e15 * 100:.0f} % of 2.5 PF", flush=True)
