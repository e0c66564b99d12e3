"""Steady-state detector loop for `rocprofv3 --kernel-trace --stats`: 3 untimed calls (tuning, capture), then N graph replays of a 2-frame group."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib, pipeline, synth
import torch

H, W, N = 2160, 3840, int(sys.argv[1]) if len(sys.argv) > 1 else 40
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = pipeline.PipeConfig(sr_crops=0)
pipe = pipeline.FramePipeline(synth.yolo11_pose_weights("s"), None, cfg, arch="s", det_precision=_lib.PREC_F32X3)
frame = torch.from_numpy(np.concatenate([synth.synthetic_frame(H, W, seed=i) for i in range(NF)], 0)).cuda()
for _ in range(3 + N):
    pipe.detect(frame, H, W, NF)
torch.cuda.synchronize()
print("done", pipe.det.last_ms())
