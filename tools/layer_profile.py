"""Per-layer conv timing of one 4K frame (detector + SR) — tuning aid. Usage on the GPU box: python tools/layer_profile.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib, pipeline, synth
import torch

H, W = 2160, 3840
cfg = pipeline.PipeConfig()
pipe = pipeline.FramePipeline(synth.yolo11_pose_weights("s"), synth.rrdbnet_weights(4, 23), cfg, arch="s",
                              det_precision=_lib.PREC_F16 if "--f16" in sys.argv else _lib.PREC_F32X3 if "--x3" in sys.argv else _lib.PREC_F32)
frame = torch.from_numpy(synth.synthetic_frame(H, W, seed=0)).cuda()
sizes = pipeline.sr_crop_sizes(32, 0)
boxes = pipeline.crop_boxes_for_sr(np.zeros((0, 21), np.float32), H, W, 32, sizes, 0)
for it in range(3):
    prof = it == 2
    pipe.det.set_profile(prof); pipe.sr.set_profile(prof)
    d, c, _, _ = pipe.detect(frame, H, W, 1)
    det_detail = pipe.det.profile_detail()
    pipe.enhance_crops(frame, H, W, boxes)
    sr_detail = pipe.sr.profile_detail()
print("== detector: total conv ms", sum(x["ms"] for x in det_detail), "stage", pipe.det.last_ms())
for x in det_detail:
    print(f'{x["name"]:70s} {x["ms"]*1e3:9.1f} us {x["flops"]/1e9:8.2f} GF {x["flops"]/max(x["ms"],1e-9)/1e9:8.1f} TF/s')
print("== sr: total conv ms", sum(x["ms"] for x in sr_detail), "call ms", pipe.sr.last_ms())
agg = {}
for x in sr_detail:
    k = x["name"].split(" ")[0] + " " + (x["name"].split(".")[-1] if "body" in x["name"] else x["name"].split(" ")[1])
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += x["ms"]; a[2] += x["flops"]
for k, a in agg.items():
    print(f'{k:40s} n={a[0]:4d} avg {a[1]/a[0]*1e3:8.1f} us  {a[2]/max(a[1],1e-9)/1e9:8.1f} TF/s  total {a[1]:.3f} ms')
