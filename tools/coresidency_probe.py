"""First experiment for DESIGN "Next 0 (d)": one detector kernel (conv_k3d, stride-2 128 -> 128 at 128^2, 122 items: model.3) and one enhancer kernel (conv_rows16, 128 -> 32 on 2,570 tiles)
launched back to back on two streams by two host threads — each alone, then both at once; at the shipped occupancy (two workgroups of ~80 KiB per CU each: they can only time-share a CU)
and with FFP_K3D_WS=32 FFP_ROWS16_WS=32 (one workgroup per CU each: a CU can hold one of each). us per launch."""
import os, sys, threading
sys.path.insert(0, os.getcwd())
import ffp_amd  # noqa: F401
from ffp_amd import _lib
IT = 300
PW = os.environ.get("PARTNER") == "pw"            # the detector kernel: conv_k3d (matrix-bound) or the HBM-bound 1x1 of model.2.cv2 (conv_pw, 96 -> 128 at 128^2)
def k3d(out):
    if PW:
        out["k3d"] = _lib.op_conv2d_time(122, 128, 128, 96, 128, 1, 1, False, _lib.PREC_F32X3, IT, 0, 10)
    else:
        out["k3d"] = _lib.op_conv2d_time(122, 128, 128, 128, 128, 3, 2, False, _lib.PREC_F32X3, IT, 0, 17)
def r16(out):
    out["r16"] = _lib.op_conv2d_time(2570, 16, 16, 128, 32, 3, 1, False, _lib.PREC_F16, IT * 8, 0, 9)
a, b, c = {}, {}, {}
k3d(a); r16(a)                      # warm-up
k3d(a); r16(b)
th = [threading.Thread(target=k3d, args=(c,)), threading.Thread(target=r16, args=(c,))]
import time
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
wall = time.perf_counter() - t0
alone = a["k3d"] * IT + b["r16"] * IT * 8
print(f"partner {'conv_pw' if PW else 'conv_k3d'} FFP_K3D_WS={os.environ.get('FFP_K3D_WS', '-')} FFP_ROWS16_WS={os.environ.get('FFP_ROWS16_WS', '-')}: alone k3d {a['k3d']:.1f} us, rows16 {b['r16']:.1f} us per launch; together k3d {c['k3d']:.1f}, rows16 {c['r16']:.1f}; "
      f"work alone (sum) {alone / 1e3:.1f} ms, together wall {wall * 1e3:.1f} ms", flush=True)
