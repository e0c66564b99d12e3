"""Do the detector's and the enhancer's launch graphs really run side by side? One 4K frame (61 items) detected N times, 32 crops enhanced N times: each alone,
then both at once from two host threads on their own streams. Wall time of the pair against the sum and the maximum of the two alone."""
import os, sys, threading, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import ffp_amd  # noqa: F401
from ffp_amd import _lib, synth, pipeline
H, W, N = 2160, 3840, 40
NF = int(os.environ.get("NF", "1")); NC = int(os.environ.get("NC", "32"))
cfg = pipeline.PipeConfig(sr_crops=NC)
dev = torch.device("cuda", 0)
pipe = pipeline.FramePipeline(synth.yolo11_pose_weights("s"), synth.rrdbnet_weights(4, 23), cfg, arch="s", det_precision=_lib.PREC_F32X3, sr_half=True)
frame = torch.from_numpy(np.concatenate([synth.synthetic_frame(H, W, seed=i) for i in range(NF)], 0)).to(dev)
boxes = pipeline.crop_boxes_for_sr(np.zeros((0, 5), np.float32), H, W, NC, pipeline.sr_crop_sizes(NC, seed=1000), seed=0)
def det(n):
    for _ in range(n):
        d, c, L, g = pipe.detect(frame, H, W, NF)
        torch.cuda.current_stream().synchronize()
        pipe.merged_count(pipe.merge_frame_of(d, c, L, 0, g)[1])
def sr(n):
    for _ in range(n):
        pipe.enhance_crops(frame[:H], H, W, boxes, wait=True)
det(3); sr(3)
def wall(fns):
    th = [threading.Thread(target=f, args=(N,)) for f in fns]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3
only_both = os.environ.get("ONLY_BOTH") == "1"        # for a kernel trace: nothing but the concurrent phase after the warm-up
a, b = (0.0, 0.0) if only_both else (wall([det]), wall([sr]))
c = wall([det, sr])
print(f"frames per detect call {NF}, crops per SR call {NC}: detect alone {a:.2f} ms, SR alone {b:.2f} ms (device {pipe.sr.last_ms():.2f}), both at once {c:.2f} ms per pair  (sum {a + b:.2f}, max {max(a, b):.2f})", flush=True)
