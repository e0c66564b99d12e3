import sys, json
sys.path.insert(0, '.')
import ffp_amd
from ffp_amd import _lib
P = _lib.PREC_F32X3
layers = [  # (name, n, h, w, cin, cout, stride)
 ("model.3", 122, 128, 128, 128, 128, 2), ("model.5", 122, 64, 64, 256, 256, 2), ("model.7", 122, 32, 32, 256, 512, 2),
 ("model.17", 122, 64, 64, 128, 128, 2), ("model.20", 122, 32, 32, 256, 256, 2),
 ("cv2.0.0", 122, 64, 64, 128, 64, 1), ("cv2.0.1", 122, 64, 64, 64, 64, 1), ("cv4.0.0", 122, 64, 64, 128, 32, 1),
 ("m2.m.cv1", 122, 128, 128, 32, 16, 1), ("m2.m.cv2", 122, 128, 128, 16, 32, 1), ("m4.m.cv1", 122, 64, 64, 64, 32, 1), ("m4.m.cv2", 122, 64, 64, 32, 64, 1),
 ("m6.c3k", 122, 32, 32, 64, 64, 1), ("m13.m.cv1", 122, 32, 32, 128, 64, 1), ("m13.m.cv2", 122, 32, 32, 64, 128, 1), ("m8.c3k", 122, 16, 16, 128, 128, 1),
 ("cv2.1.0", 122, 32, 32, 256, 64, 1), ("cv2.2.0", 122, 16, 16, 512, 64, 1),
]
for name, n, h, w, ci, co, s in layers:
    row = {}
    for shape in (-1, 2, 4, 17, 18, 19, 20, 21):
        try:
            row[shape] = round(_lib.op_conv2d_time(n, h, w, ci, co, 3, s, False, P, 20, 0, shape), 1)
        except Exception as e:
            row[shape] = None
    flops = 2.0 * n * ((h - 1)//s + 1) * ((w - 1)//s + 1) * ci * co * 9
    best = min(v for v in row.values() if v)
    print(f"{name:10s} {ci:4d}->{co:4d} s{s} @{h:3d}: " + " ".join(f"{k}:{v}" for k, v in row.items()) + f"  best {flops/best/1e6:.0f} TF/s (of 833)", flush=True)
