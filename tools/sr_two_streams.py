"""Does splitting a frame's crops over two enhancer handles (two streams, two graphs) shorten the SR stage? (tuning aid)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ffp_amd  # noqa
from ffp_amd import _lib, synth, pipeline
H, W = 2160, 3840
frame = torch.from_numpy(synth.synthetic_frame(H, W, seed=1)).cuda()
rng = np.random.default_rng(0)
sizes = pipeline.sr_crop_sizes(32)
boxes = []
for s_ in sizes:
    h = w = int(s_)
    x = int(rng.integers(0, W - w)); y = int(rng.integers(0, H - h))
    boxes.append((x, y, x + w, y + h))
boxes = np.array(boxes, np.int32)
Wt = synth.rrdbnet_weights(4)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
enh = [_lib.Enhancer(Wt, scale=4, half=True) for _ in range(max(K, 4))]
lib = _lib.lib()
def run(parts):
    outs = []
    for e, b in zip(enh, parts):
        b = np.ascontiguousarray(b, np.int32)
        tot = int(sum(((int(q[3] - q[1]) * 4) * (int(q[2] - q[0]) * 4) * 3 + 15) // 16 * 16 for q in b))
        out = torch.empty(tot, dtype=torch.uint8, device="cuda")
        offs = np.zeros(len(b) + 1, np.int64)
        _lib._check(lib.ffp_sr_enhance_crops_dev_async(e.handle, frame.data_ptr(), H, W, _lib._ip(b), len(b), 400, 10, out.data_ptr(), tot,
                                                       offs.ctypes.data_as(C.POINTER(C.c_int64))))
        outs.append(out)
    for e, _ in zip(enh, parts):
        _lib._check(lib.ffp_sr_wait(e.handle))
    return outs
for k in (1, 2, 3, 4):
    order = np.argsort([-(b[3] - b[1]) * (b[2] - b[0]) for b in boxes])
    parts = [boxes[order[i::k]] for i in range(k)]
    for _ in range(4):
        run(parts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run(parts)
    torch.cuda.synchronize()
    print(f"{k} handle(s): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per 32 crops", flush=True)
