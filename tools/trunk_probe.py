"""conv_trunk_kernel (force_shape 25, one layer) against conv_rows16_kernel (9): microseconds per launch and algorithmic TFLOP/s on single-tile
images (the round-3 probe geometry) and on 32 x 32 images (what a 32 x 16 tile likes). Then the whole enhancer, fused body vs per-layer launches."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import ffp_amd  # noqa: F401
from ffp_amd import _lib, synth, pipeline

def t(n, hw, cin, cout, shape):
    return min(_lib.op_conv2d_time(n, hw, hw, cin, cout, 3, 1, False, _lib.PREC_F16, 30, 0, shape) for _ in range(3))

for hw, ns in ((32, (64, 256, 643, 1024)), (16, (1024, 2570))):
    for n in ns:
        for cin, cout in ((64, 32), (96, 32), (128, 32), (160, 32), (192, 64)):
            fl = 2.0 * cin * cout * 9 * hw * hw * n
            a, b = t(n, hw, cin, cout, 9), t(n, hw, cin, cout, 25)
            print(f"images={n:5d} {hw}x{hw} {cin:3d}->{cout:2d}  rows16 {a:7.1f} us ({fl / a / 1e6:6.0f} TF/s)  trunk {b:7.1f} us ({fl / b / 1e6:6.0f} TF/s)  x{a / b:.2f}", flush=True)

e = _lib.Enhancer(synth.rrdbnet_weights(4, 23), 4, 23, half=True)
rng = np.random.default_rng(0)
for n_crops in (32, 320):
    sizes = pipeline.sr_crop_sizes(n_crops, seed=7)
    imgs = [rng.integers(0, 256, (int(s), int(s), 3), dtype=np.uint8) for s in sizes]
    px = int(sum(int(s) ** 2 for s in sizes))
    for fused in (False, True, False, True):
        e.set_fused_body(fused)
        for _ in range(3):
            e.enhance_batch(imgs)
        t0 = time.perf_counter()
        for _ in range(5):
            e.enhance_batch(imgs)
        dt = (time.perf_counter() - t0) / 5
        print(f"crops={n_crops} ({px} px) fused={fused}: host wall {dt * 1e3:.2f} ms, device {e.last_ms():.2f} ms, body-equivalent {35.8e6 * px / (e.last_ms() * 1e-3) / 1e12:.0f} TF/s", flush=True)
