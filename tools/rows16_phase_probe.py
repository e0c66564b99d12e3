"""Where conv_rows16_kernel's time goes: microseconds per launch with phase-skip bits (build with FFP_EXTRA_FLAGS=-DFFP_R16_DBG=1).
bits: 1 no epilogue, 2 no MFMA, 4 no staging requests, 8 no staging LDS writes, 16 no fragment reads, 32 no barriers"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib
masks = [0, 1, 2, 12, 16, 32, 13, 29, 61, 19, 31, 63]
for n, cin, cout in ((2048, 128, 32), (2048, 64, 32), (2048, 192, 64)):
    flops = 2.0 * cin * cout * 9 * 256 * n
    row = []
    for m in masks:
        us = min(_lib.op_conv2d_time(n, 16, 16, cin, cout, 3, 1, False, _lib.PREC_F16, 40, m, 9) for _ in range(2))
        row.append(f"{m}:{us:.1f}")
    base = float(row[0].split(":")[1])
    print(f"tiles={n} {cin}->{cout}: " + " ".join(row) + f"   full = {flops / base / 1e6:.0f} TF/s", flush=True)
