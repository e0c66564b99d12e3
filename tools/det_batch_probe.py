"""Detector time per frame when 1, 2 or 3 frames' items form one ragged batch (tuning aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ffp_amd  # noqa
from ffp_amd import _lib, synth, pipeline
H, W = 2160, 3840
cfg = pipeline.PipeConfig(slice_h=512, slice_w=512, overlap=0.2, imgsz=512, conf=0.5, pp_type="GREEDYNMM", sr_crops=0)
pipe = pipeline.FramePipeline(synth.yolo11_pose_weights("s"), None, cfg, arch="s", device=0, det_precision=_lib.PREC_F32X3, sr_half=True, rank=0, world=1)
for B in (1, 2, 3, 1, 2):
    sf = torch.from_numpy(np.concatenate([synth.synthetic_frame(H, W, seed=i) for i in range(B)], 0)).cuda()
    for _ in range(3):
        pipe.detect(sf, H, W, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        pipe.detect(sf, H, W, B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    print(f"B={B}: {dt * 1e3:.3f} ms per call, {dt * 1e3 / B:.3f} ms per frame, stage {pipe.det.last_ms()}", flush=True)
