"""One conv shape launched repeatedly (for rocprofv3 --pmc passes): python tools/_pmc_one.py n cin cout shape [h w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib
n, cin, cout, shape = (int(v) for v in sys.argv[1:5])
h, w = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (41, 42)
print(_lib.op_conv2d_time(n, h, w, cin, cout, 3, 1, False, _lib.PREC_F16, 20, 0, shape))
