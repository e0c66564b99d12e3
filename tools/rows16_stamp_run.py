import sys; sys.path.insert(0,'.')
import ffp_amd
from ffp_amd import _lib
for cin,cout in ((128,32),(192,64)):
    print(cin, cout, _lib.op_conv2d_time(2570, 16, 16, cin, cout, 3, 1, False, _lib.PREC_F16, 1, 0, 9), flush=True)
