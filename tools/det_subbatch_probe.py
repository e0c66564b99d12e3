"""Would the detector's high-resolution layers run faster on SUB-BATCHES whose tensors stay in the 256 MiB Infinity Cache? Per-item time of every conv layer
for n slices of 512 x 512 per call (n = 4 .. 122), from the per-launch events of a profiled (eager) call. A layer whose per-item time falls at small n is paying
HBM for its neighbours' tensors at the batch sizes the pipeline uses."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import ffp_amd  # noqa: F401
from ffp_amd import _lib, synth
W = synth.yolo11_pose_weights("s")
det = _lib.Detector(W, arch="s", nc=int(W["model.23.cv3.0.2.weight"].shape[0]), nkpt=int(W["model.23.cv4.0.2.weight"].shape[0]) // 3, device=0, precision=_lib.PREC_F32X3)
frame = synth.synthetic_frame(2160, 3840, seed=0)
tiles_all = [[x, y, x + 512, y + 512] for y in range(0, 2160 - 511, 410) for x in range(0, 3840 - 511, 410)]
tiles_all = (tiles_all * 4)[:122]
res = {}
for n in (4, 8, 16, 32, 61, 122):
    t = tiles_all[:n]
    for it in range(4):
        det.set_profile(it == 3)
        det.infer_tiles(frame, t, 512, 0.5, 0.7, 300)
    det.set_profile(False)
    for x in det.profile_detail():
        variant, name = x["name"].split(" ", 1)
        res.setdefault(name, {})[n] = (x["ms"] * 1e3 / n, variant)
    tot = sum(x["ms"] for x in det.profile_detail())
    print(f"n={n:4d}: conv total {tot:.3f} ms = {tot / n * 1e3:.1f} us per item", flush=True)
names = sorted(res, key=lambda k: -res[k].get(122, (0,))[0])[:40]
print(f"{'layer':34s} " + " ".join(f"{'n=' + str(n):>9s}" for n in (4, 8, 16, 32, 61, 122)) + "   us per item; variant at n=122")
for k in names:
    print(f"{k:34s} " + " ".join(f"{res[k].get(n, (float('nan'),))[0]:9.2f}" for n in (4, 8, 16, 32, 61, 122)) + f"   {res[k][122][1]}")
