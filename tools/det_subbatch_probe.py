"""Does a detection sub-batch small enough for the 256 MiB Infinity Cache run its big early layers faster per item?
Per-layer time per item for batches of 8 / 15 / 30 / 61 / 122 items (eager, HIP events)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib, pipeline, synth
import torch

H, W = 2160, 3840
Wd = synth.yolo11_pose_weights("s")
det = _lib.Detector(Wd, arch="s", precision=_lib.PREC_F32X3)
frame = np.concatenate([synth.synthetic_frame(H, W, seed=i) for i in range(2)], 0)
cfg = pipeline.PipeConfig()
items = pipeline.frame_items(H, W, cfg, 2)
slices = [tuple(t) for t in items if (t[2] - t[0]) == 512][:120]
names = ["model.1.conv", "model.2.cv1.conv", "model.2.m.0.cv1.conv", "model.2.m.0.cv2.conv", "model.2.cv2.conv", "model.3.conv", "model.4.cv1.conv", "model.4.cv2.conv",
         "model.5.conv", "model.16.cv1.conv", "model.23.cv2.0.0.conv"]
res = {}
for n in (8, 15, 30, 60, 120):
    t = slices[:n]
    for it in range(4):
        det.set_profile(it == 3)
        det.infer_tiles(frame, t, 512, 0.5)
    d = {x["name"].split(" ", 1)[1]: x["ms"] for x in det.profile_detail()}
    res[n] = d
    tot = sum(d.values())
    print(f"n={n:4d} total conv {tot:8.3f} ms  per item {tot/n*1e3:8.1f} us  stage {det.last_ms()}", flush=True)
print(f"{'layer':28s}" + "".join(f"{n:>10d}" for n in res))
for nm in names:
    print(f"{nm:28s}" + "".join(f"{res[n][nm]/n*1e3:10.2f}" for n in res) + "   us per item")
