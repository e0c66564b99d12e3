"""Attribute conv kernel time to phases by skipping them (tuning aid; results of skipped runs are wrong by design)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib
SH = ["wide", "wideH", "narrow2", "narrow2H", "narrow1", "narrow1H", "rows2st", "rows3st"]
cases = [
    # name, precision, n,h,w,cin,cout,k,s
    ("sr conv1 64->32 (32 crops 48^2-ish)", _lib.PREC_F16, 32, 41, 42, 64, 32, 3, 1),
    ("sr conv4 160->32", _lib.PREC_F16, 32, 41, 42, 160, 32, 3, 1),
    ("sr conv5 192->64", _lib.PREC_F16, 32, 41, 42, 192, 64, 3, 1),
    ("sr conv_hr 64->64 @16x", _lib.PREC_F16, 32, 164, 168, 64, 64, 3, 1),
    ("det model.3 k3s2 64->128 @128^2 x61", _lib.PREC_F32X3, 61, 128, 128, 64, 128, 3, 2),
    ("det model.2.cv2 k1 96->128 @128^2 x61", _lib.PREC_F32X3, 61, 128, 128, 96, 128, 1, 1),
    ("det model.16.cv1 k1 512->128 @64^2 x61", _lib.PREC_F32X3, 61, 64, 64, 512, 128, 1, 1),
]
for name, prec, n, h, w, cin, cout, k, s in cases:
    print("==", name)
    for shape in range(-1, 8):
        try:
            t = _lib.op_conv2d_time(n, h, w, cin, cout, k, s, False, prec, 30, 0, shape)
        except Exception as e:
            continue
        extra = ""
        if shape == -1 or True:
            parts = []
            for m, lab in ((1, "-stores"), (2, "-mfma"), (3, "-stores-mfma"), (15, "fetch0+barriers only")):
                try:
                    parts.append(f"{lab} {_lib.op_conv2d_time(n, h, w, cin, cout, k, s, False, prec, 30, m, shape):7.1f}")
                except Exception:
                    pass
            extra = "  ".join(parts)
        flops = 2.0 * cin * cout * k * k * n * ((h + s - 1) // s) * ((w + s - 1) // s)
        print(f"  shape {('auto' if shape < 0 else SH[shape]):9s} {t:8.1f} us {flops / t / 1e6:7.1f} TF/s   {extra}")
