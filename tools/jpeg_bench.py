"""JPEG at the file boundaries, measured separately from the frames/s metric: the device codec against libjpeg-turbo
(Pillow) on the host, for the two boundaries of the path — reading a 4K frame, writing a frame's enhanced crops."""
import io, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib, synth, pipeline
from PIL import Image
import torch


def t(fn, n):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


frame = synth.synthetic_frame(2160, 3840, seed=0)
b = io.BytesIO(); Image.fromarray(frame).save(b, "JPEG", quality=95); data = b.getvalue()
d = torch.zeros((2160, 3840, 3), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
print(f"4K frame, quality 95, 4:2:0: {len(data) / 1e6:.2f} MB file")
print(f"  decode  libjpeg-turbo (Pillow, 1 core) -> host array          {t(lambda: np.asarray(Image.open(io.BytesIO(data)).convert('RGB')), 5):7.1f} ms")
print(f"  decode  device codec -> frame in device memory                {t(lambda: _lib.jpeg_decode_dev(data, d.data_ptr(), 3840 * 3, d.numel(), bgr=True), 5):7.1f} ms  (" + ("host Huffman decoding into pinned coefficient planes, 37 MB of int16 coefficients over PCIe" if os.environ.get("FFP_JPEG_HOST_HUFFMAN") == "1" else "the file over PCIe, Huffman decoding on the device") + ", IDCT / upsampling / colour on the device)")
print("  decode stats (device, host fallbacks, extra sync rounds):", _lib.jpeg_decode_stats())
dfr = torch.from_numpy(frame).cuda(); torch.cuda.synchronize()
print(f"  encode  libjpeg-turbo (Pillow, 1 core) from host array        {t(lambda: Image.fromarray(frame).save(io.BytesIO(), 'JPEG', quality=95), 5):7.1f} ms")
print(f"  encode  device codec from device memory -> file bytes on host {t(lambda: _lib.jpeg_encode_dev(dfr.data_ptr(), 2160, 3840, 3840 * 3, 95, bgr=False), 5):7.1f} ms")
sizes = pipeline.sr_crop_sizes(32, 0)
crops = [synth.synthetic_frame(int(s) * 4, int(s) * 4, seed=i) for i, s in enumerate(sizes)]
dcr = [torch.from_numpy(c).cuda() for c in crops]; torch.cuda.synchronize()
px = sum(c.shape[0] * c.shape[1] for c in crops)
print(f"32 enhanced crops of one frame ({px / 1e6:.2f} Mpx, sizes x4 of {sorted(set(int(s) for s in sizes))}):")
print(f"  encode  libjpeg-turbo (Pillow, 1 core), crops already on host {t(lambda: [Image.fromarray(c).save(io.BytesIO(), 'JPEG', quality=95) for c in crops], 5):7.1f} ms  (+ {px * 3 / 1e6:.1f} MB of D2H first)")
print(f"  encode  device codec from the SR output buffer                {t(lambda: [_lib.jpeg_encode_dev(x.data_ptr(), x.shape[0], x.shape[1], x.shape[1] * 3, 95, bgr=True) for x in dcr], 5):7.1f} ms  (one call per crop)")
flat = torch.cat([x.reshape(-1) for x in dcr]); torch.cuda.synchronize()
offs = np.cumsum([0] + [c.size for c in crops])[:-1]
hs, ws = [c.shape[0] for c in crops], [c.shape[1] for c in crops]
print(f"  encode  device codec, the 32 crops as one batch               {t(lambda: _lib.jpeg_encode_batch_dev(flat.data_ptr(), offs, hs, ws, 95, bgr=True), 10):7.1f} ms  (ffp_jpeg_encode_batch_dev)")
