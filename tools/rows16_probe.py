"""A/B of the Real-ESRGAN body conv kernels on one device, one process: first-generation row-reuse kernel (shape 6) vs
rows16 (shape 9) over the dense-block shapes and batch sizes, plus rows16's switches (1 no epilogue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib

def t(n, cin, cout, shape, dbg=0, it=60, hw=(41, 42)):
    try:
        return min(_lib.op_conv2d_time(n, hw[0], hw[1], cin, cout, 3, 1, False, _lib.PREC_F16, it, dbg, shape) for _ in range(2))
    except Exception as e:
        return float("nan")

for n in (32, 64, 128):
    for cin, cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64)):
        a, b = t(n, cin, cout, 6), t(n, cin, cout, 9)
        fl = 2.0 * cin * cout * 9 * n * 41 * 42
        print(f"n={n:3d} tiles={n*9:4d} {cin:3d}->{cout:2d}  rows {a:6.1f}  rows16 {b:6.1f} us  no-epi {t(n,cin,cout,9,1):6.1f}   rows16 {fl/b/1e6:6.0f} TF/s", flush=True)
for n in (64, 128, 256, 384, 512, 768, 1024, 2048, 4096):
    a, b = t(n, 128, 32, 6, hw=(16, 16)), t(n, 128, 32, 9, hw=(16, 16))
    print(f"16x16 tiles={n:4d} 128->32 rows {a:6.1f} rows16 {b:6.1f}  {2.0*128*32*9*256*n/b/1e6:6.0f} TF/s", flush=True)
