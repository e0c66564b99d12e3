"""Does a second detector engine on its own stream (independent frame groups in flight together) raise detector throughput?
One engine vs two engines driven from two host threads, same total number of 2-frame groups."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib, pipeline, synth
import torch

H, W, NF, N = 2160, 3840, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = pipeline.PipeConfig(sr_crops=0)
Wd = synth.yolo11_pose_weights("s")
frame = torch.from_numpy(np.concatenate([synth.synthetic_frame(H, W, seed=i) for i in range(NF)], 0)).cuda()
pipes = [pipeline.FramePipeline(Wd, None, cfg, arch="s", det_precision=_lib.PREC_F32X3) for _ in range(2)]
for p in pipes:
    for _ in range(3):
        p.detect(frame, H, W, NF)
torch.cuda.synchronize()

def run(p, n):
    for _ in range(n):
        p.detect(frame, H, W, NF)

t0 = time.perf_counter(); run(pipes[0], N); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"one engine : {(t1 - t0) / N * 1e3:.3f} ms per 2-frame group", flush=True)
th = [threading.Thread(target=run, args=(p, N // 2)) for p in pipes]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"two engines: {(t1 - t0) / (N // 2 * 2) * 1e3:.3f} ms per 2-frame group", flush=True)
