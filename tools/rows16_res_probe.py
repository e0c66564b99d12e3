"""conv_rows16 streaming (shape 9) vs weight-resident (shape 23) over the dense-block shapes and launch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib
def t(n, cin, cout, shape, hw=(16, 16)):
    return min(_lib.op_conv2d_time(n, hw[0], hw[1], cin, cout, 3, 1, False, _lib.PREC_F16, 40, 0, shape) for _ in range(2))
for n in (256, 1024, 2570, 4096):
    for cin, cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64)):
        a, b, c = t(n, cin, cout, 9), t(n, cin, cout, 23), t(n, cin, cout, 24)
        fl = 2.0 * cin * cout * 9 * 256 * n
        print(f"tiles={n:5d} {cin:3d}->{cout:2d}  rows16 {a:7.1f} us ({fl/a/1e6:5.0f} TF/s)  resident {b:7.1f} us x{a/b:.2f}  producer/consumer {c:7.1f} us ({fl/c/1e6:5.0f} TF/s) x{a/c:.2f}", flush=True)
