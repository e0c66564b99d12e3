"""One conv layer, one workgroup shape, a few launches: the process rocprofv3 wraps for SQ / TCC counter passes of conv_k3d.hip.
    python3 tools/k3d_one.py <n> <h> <w> <cin> <cout> <stride> <shape> [iters]"""
import sys
sys.path.insert(0, '.')
import ffp_amd  # noqa: F401
from ffp_amd import _lib
n, h, w, ci, co, s, shape = (int(x) for x in sys.argv[1:8])
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
print(_lib.op_conv2d_time(n, h, w, ci, co, 3, s, False, _lib.PREC_F32X3, iters, 0, shape))
