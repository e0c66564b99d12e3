import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ffp_amd
from ffp_amd import _lib, synth, pipeline
torch.cuda.init()
H, W = 2160, 3840
cfg = pipeline.PipeConfig(slice_h=512, slice_w=512, overlap=0.2, imgsz=512, conf=0.5, sr_crops=0)
t0 = time.perf_counter()
pipe = pipeline.FramePipeline(synth.yolo11_pose_weights("s"), None, cfg, arch="s", device=0, det_precision=_lib.PREC_F32X3)
t1 = time.perf_counter()
sf = torch.from_numpy(synth.synthetic_frame(H, W, seed=0)).cuda()
torch.cuda.synchronize()
t2 = time.perf_counter()
pipe.detect(sf, H, W, 1); torch.cuda.synchronize(); t3 = time.perf_counter()
pipe.detect(sf, H, W, 1); torch.cuda.synchronize(); t4 = time.perf_counter()
pipe.detect(sf, H, W, 1); torch.cuda.synchronize(); t5 = time.perf_counter()
print(f"create {t1-t0:.2f}s  first call (plan + tune) {t3-t2:.2f}s  second (capture) {t4-t3:.3f}s  third {t5-t4:.4f}s")
