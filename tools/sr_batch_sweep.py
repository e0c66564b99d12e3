"""Device time of the whole enhancer against the number of crops per call (same size mix): is the 320-crop batch (dense-block buffer 261 MB,
past the 256 MiB Infinity Cache) slower per pixel than batches whose working set stays cached? Per-layer launches and the fused body."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import ffp_amd  # noqa: F401
from ffp_amd import _lib, synth, pipeline
e = _lib.Enhancer(synth.rrdbnet_weights(4, 23), 4, 23, half=True)
rng = np.random.default_rng(0)
sizes_all = pipeline.sr_crop_sizes(320, seed=7)
for n_crops in (32, 64, 96, 128, 160, 240, 320):
    sizes = sizes_all[:n_crops]
    imgs = [rng.integers(0, 256, (int(s), int(s), 3), dtype=np.uint8) for s in sizes]
    px = int(sum(int(s) ** 2 for s in sizes))
    row = []
    for fused in (False, True):
        e.set_fused_body(fused)
        ms = []
        for _ in range(6):
            e.enhance_batch(imgs)
            ms.append(e.last_ms())
        row.append(min(ms[2:]))
    print(f"crops={n_crops:4d} px={px:7d}  per-layer {row[0]:7.2f} ms ({row[0] * 1e6 / px:6.1f} ns/px, {35.8e6 * px / row[0] / 1e9:5.0f} TF/s)   fused {row[1]:7.2f} ms ({row[1] * 1e6 / px:6.1f} ns/px, {35.8e6 * px / row[1] / 1e9:5.0f} TF/s)", flush=True)
