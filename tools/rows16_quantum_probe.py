"""Is conv_rows16's launch time a staircase in the number of 16 x 16 tiles (512 workgroup slots: 2 per CU)? Single layers on n one-tile images."""
import os, sys
sys.path.insert(0, os.getcwd())
import ffp_amd  # noqa: F401
from ffp_amd import _lib
for cin, cout in ((128, 32), (192, 64)):
    row = []
    for n in (1536, 2040, 2048, 2056, 2304, 2552, 2560, 2568, 2690, 2816, 3064, 3072, 3080):
        us = min(_lib.op_conv2d_time(n, 16, 16, cin, cout, 3, 1, False, _lib.PREC_F16, 30, 0, 9) for _ in range(3))
        row.append(f"{n}:{us:.1f}")
    print(f"{cin}->{cout}  " + "  ".join(row), flush=True)
