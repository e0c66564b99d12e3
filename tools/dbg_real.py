import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import ffp_amd
from ffp_amd import _lib, synth
from PIL import Image
from util import match_by_iou, iou_xyxy
G = "tests/golden"
z = np.load(G + "/real_expected.npz")
det = _lib.Detector(synth.yolo11_pose_weights("n"), arch="n", precision=_lib.PREC_F32)
for k, c in enumerate(z["cases"]):
    name, sub, imgsz = str(c).split("|")
    img = np.asarray(Image.open(f"{G}/real/{name}.png").convert("RGB"))
    h, w = img.shape[:2]
    tile = tuple(int(v) for v in sub.split(",")) if sub else (0, 0, w, h)
    d = det.infer_tiles(img, [tile], int(imgsz), 0.25, 0.7, 300)[0]
    exy, econf = z[f"case{k}_xyxy"], z[f"case{k}_conf"]
    print(c, d.shape[0], exy.shape[0])
    m = match_by_iou(exy, d[:, :4])
    for i, j, u in m:
        if abs(d[j, 4] - econf[i]) > 1e-3 or u < 0.999:
            print("  exp", i, exy[i], econf[i], " gpu", j, d[j, :5], "iou", u)
            # other gpu boxes near
            ious = iou_xyxy(np.repeat(exy[i:i+1], d.shape[0], 0), d[:, :4])
            for jj in np.argsort(-ious)[:3]: print("      gpu cand", jj, d[jj, :5], ious[jj])
            ious = iou_xyxy(np.repeat(d[j:j+1, :4], exy.shape[0], 0), exy)
            for ii in np.argsort(-ious)[:3]: print("      exp cand", ii, exy[ii], econf[ii], ious[ii])
