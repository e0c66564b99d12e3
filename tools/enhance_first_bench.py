"""Secondary measurement, SURVEY §8 (f1): the reference's enhance-first ordering (pipeline_v4_yolo/app_yolo_full.py:87-123)
on a resident 1920x1080 picture: Real-ESRGAN x2plus (tile 400, pad 10, fp16) -> 3840x2160 -> SAHI 512/0.2 + YOLO11s-pose
(f32x3) + GREEDYNMM merge. Prints one JSON line (not the contract bench; see bench.py)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ffp_amd  # noqa
from ffp_amd import _lib, synth, pipeline

H, W = 1080, 1920
torch.cuda.init()
cfg = pipeline.PipeConfig(slice_h=512, slice_w=512, overlap=0.2, imgsz=512, conf=0.5, sr_crops=0)
pipe = pipeline.FramePipeline(synth.yolo11_pose_weights("s"), None, cfg, arch="s", device=0, det_precision=_lib.PREC_F32X3)
enh2 = _lib.Enhancer(synth.rrdbnet_weights(2, 23), 2, 23, half=True)
frame = torch.from_numpy(synth.synthetic_frame(H, W, seed=0)).cuda()
for _ in range(3):
    pipe.enhance_first(frame, H, W, enhancer=enh2)
torch.cuda.synchronize()
t_sr = t_all = 0.0
N = 8
for _ in range(N):
    t0 = time.perf_counter()
    e = pipe.enhance_frame(frame, H, W, enh2)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dets, counts, L, gathered = pipe.detect(e, 2 * H, 2 * W, 1)
    rows, n = pipe.merge_frame_of(dets, counts, L, 0, gathered)
    k = int(n.item())
    t2 = time.perf_counter()
    t_sr += t1 - t0
    t_all += t2 - t0
print(json.dumps({"metric": "enhance-first 1080p->4K frames/s (ESRGAN x2 tile 400 + SAHI + YOLO11s)", "value": round(N / t_all, 3), "unit": "frames/s",
                  "ms_per_frame": round(t_all / N * 1e3, 2), "sr_ms": round(t_sr / N * 1e3, 2), "detect_merge_ms": round((t_all - t_sr) / N * 1e3, 2),
                  "sr_tiles": 15, "detections": k, "data": "synthetic"}))
