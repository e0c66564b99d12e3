"""One SR batch (the crops of 10 frames) as ONE ragged launch sequence on one enhancer, or as S independent parts on S enhancers (S streams)
whose launches interleave on the card: does the batch finish sooner? (two lanes do: profiles/r03_two_lanes_probe.txt)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib, pipeline, synth
import torch

H, W, NF = 2160, 3840, int(os.environ.get("NF", 10))
Ws = synth.rrdbnet_weights(4, 23)
cfg = pipeline.PipeConfig()
frames = [torch.from_numpy(synth.synthetic_frame(H, W, seed=i % 2)).cuda() for i in range(NF)]
rng = np.random.default_rng(0)
boxes_pf = []
for f in range(NF):
    sizes = pipeline.sr_crop_sizes(32, seed=1000 + f)
    boxes_pf.append(pipeline.crop_boxes_for_sr(np.zeros((0, 21), np.float32), H, W, 32, sizes, seed=f))
boxes = np.ascontiguousarray(np.concatenate(boxes_pf, 0), np.int32)
fidx = np.ascontiguousarray(np.concatenate([np.full(len(b), i, np.int32) for i, b in enumerate(boxes_pf)]))
area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
ptrs = [t.data_ptr() for t in frames]
for S in (1, 2, 3, 4):
    enh = [_lib.Enhancer(Ws, 4, 23, device=0, half=True) for _ in range(S)]
    part = pipeline.lpt_assign(area.astype(np.int64), S)
    sub = []
    for k in range(S):
        b, fi = np.ascontiguousarray(boxes[part == k]), np.ascontiguousarray(fidx[part == k])
        tot = pipeline.FramePipeline.sr_out_bytes(b, H, W, 4) if hasattr(pipeline.FramePipeline, "sr_out_bytes") else int((((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).sum()) * 16 * 3)
        out = torch.empty((int(tot),), dtype=torch.uint8, device="cuda")
        sub.append((b, fi, out, int(tot)))

    def run():
        for e, (b, fi, out, tot) in zip(enh, sub):
            e.enhance_crops_dev(ptrs, H, W, b, out.data_ptr(), tot, fi, cfg.sr_tile, cfg.sr_tile_pad, wait=False)
        for e in enh:
            _lib._check(_lib.lib().ffp_sr_wait(e.handle))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f"{NF} frames' crops ({len(boxes)} crops, {int(area.sum())} px) in {S} part(s): {dt:.2f} ms per batch = {dt / NF:.3f} ms per frame", flush=True)
    del enh
