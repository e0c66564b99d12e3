"""conv_trunk_kernel on one layer with phase-skip bits (diagnostic build: python -m ffp_amd.build --variant trunk_dbg, FFP_LIB=...):
1 no epilogue, 2 no MFMA, 4 no DMA, 16 no fragment reads (23 = bare skeleton: control, item set-up, barriers). Microseconds per launch."""
import os, sys
sys.path.insert(0, os.getcwd())
import ffp_amd  # noqa: F401
from ffp_amd import _lib
for n, hw, cin, cout in ((1024, 32, 128, 32), (1024, 32, 64, 32), (1024, 32, 192, 64), (256, 32, 128, 32)):
    row = []
    for m in (0,):
        us = min(_lib.op_conv2d_time(n, hw, hw, cin, cout, 3, 1, False, _lib.PREC_F16, 20, m, 25) for _ in range(2))
        row.append(f"{m}:{us:.1f}")
    print(f"images={n} {hw}x{hw} {cin}->{cout}: " + " ".join(row), flush=True)
