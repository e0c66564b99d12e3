"""In-kernel stamps of conv_trunk_kernel (diagnostic build trunk_dbg; FFP_TRUNK_DUMP=1 prints the sums of workgroup 0 after every launch)."""
import os, sys
sys.path.insert(0, os.getcwd())
import ffp_amd  # noqa: F401
from ffp_amd import _lib
for n, hw, cin, cout in ((1024, 32, 128, 32), (1024, 32, 192, 64), (64, 32, 128, 32)):
    print(f"== images={n} {hw}x{hw} {cin}->{cout}", flush=True)
    sys.stderr.flush()
    _lib.op_conv2d_time(n, hw, hw, cin, cout, 3, 1, False, _lib.PREC_F16, 1, 0, 25)
