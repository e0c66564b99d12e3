import sys
sys.path.insert(0, '.')
import ffp_amd
from ffp_amd import _lib
P = _lib.PREC_F32X3
# dbg bits of conv_k3d_kernel: 1 no epilogue, 2 no MFMA, 4 no piece requests, 8 no split + LDS writes, 16 no A requests, 32 no B reads, 64 no barriers
cases = [("model.3", 122, 128, 128, 128, 128, 2, 17), ("model.5", 122, 64, 64, 256, 256, 2, 17), ("cv2.0.0", 122, 64, 64, 128, 64, 1, 21), ("m2.m.cv1", 122, 128, 128, 32, 16, 1, 19)]
masks = [0, 1, 2, 4, 8, 16, 32, 64, 4 | 8, 2 | 16 | 32, 1 | 2 | 16 | 32, 1 | 4 | 8, 1 | 4 | 8 | 16, 1 | 4 | 8 | 32, 1 | 4 | 8 | 16 | 32, 1 | 2 | 4 | 8 | 16 | 32, 127]
for name, n, h, w, ci, co, s, shape in cases:
    out = []
    for m in masks:
        out.append(f"{m}:{_lib.op_conv2d_time(n, h, w, ci, co, 3, s, False, P, 20, m, shape):.0f}")
    print(name, "shape", shape, " ".join(out), flush=True)
