"""4K JPEG decode, device path only, for rocprofv3 --kernel-trace --stats: which of the decode kernels cost what."""
import io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib, synth
from PIL import Image
import torch

frame = synth.synthetic_frame(2160, 3840, seed=0)
b = io.BytesIO(); Image.fromarray(frame).save(b, "JPEG", quality=95); data = b.getvalue()
d = torch.zeros((2160, 3840, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    _lib.jpeg_decode_dev(data, d.data_ptr(), 3840 * 3, d.numel(), bgr=True)
n = int(os.environ.get("N", 20))
t0 = time.perf_counter()
for _ in range(n):
    _lib.jpeg_decode_dev(data, d.data_ptr(), 3840 * 3, d.numel(), bgr=True)
dt = (time.perf_counter() - t0) / n * 1e3
print(f"4K decode {dt:.3f} ms per frame; stats {_lib.jpeg_decode_stats()}")
