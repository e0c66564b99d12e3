"""GPU parity of the JPEG encoder (csrc/jpeg.hip through ffp_jpeg_encode / ffp_jpeg_encode_dev) — SURVEY.md §8 f2. The bar is the
FILE: byte-identical to oracle/jpeg_ref.py (itself byte-identical to libjpeg-turbo via Pillow, tests/test_jpeg_oracle.py) and, checked
directly here too, to what Pillow writes — i.e. what the reference's cv2.imwrite(path, crop) leaves on disk for the same pixels."""
import io
import os

import numpy as np
import pytest
from PIL import Image

from oracle import jpeg_ref as J

pytestmark = pytest.mark.gpu
REAL = os.path.join(os.path.dirname(__file__), "golden", "real")


def pil_jpeg(img, q):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=q)
    return b.getvalue()


def test_files_identical_to_oracle_and_pillow_over_sizes_and_qualities(gpu_lib):
    rng = np.random.default_rng(1)
    from ffp_amd import synth
    big = synth.synthetic_frame(300, 400, seed=2)
    for k in range(50):
        h, w = int(rng.integers(1, 120)), int(rng.integers(1, 150))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if k % 3 == 0 else np.ascontiguousarray(big[k:k + h, 2 * k:2 * k + w])
        q = (95, 75, 100, 10, 50)[k % 5]
        got = gpu_lib.jpeg_encode(img, q)
        assert got == J.encode(img, q), (img.shape, q)
        assert got == pil_jpeg(img, q)


def test_bgr_order_and_saturated_images(gpu_lib):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (45, 61, 3), dtype=np.uint8)
    assert gpu_lib.jpeg_encode(np.ascontiguousarray(img[..., ::-1]), 95, bgr=True) == pil_jpeg(img, 95)
    for v in (0, 255):
        flat = np.full((33, 20, 3), v, np.uint8)
        assert gpu_lib.jpeg_encode(flat, 95) == pil_jpeg(flat, 95)
    noise = (rng.integers(0, 2, (64, 64, 3)) * 255).astype(np.uint8)          # maximal coefficients at quality 100: long codes, many 0xFF bytes
    assert gpu_lib.jpeg_encode(noise, 100) == pil_jpeg(noise, 100)


def test_full_frame_and_real_photographs(gpu_lib):
    from ffp_amd import synth
    frame = synth.synthetic_frame(1080, 1920, seed=9)
    got = gpu_lib.jpeg_encode(frame, 95)
    assert got == pil_jpeg(frame, 95)
    back = np.asarray(Image.open(io.BytesIO(got)).convert("RGB"))
    assert back.shape == frame.shape
    for name in sorted(f for f in os.listdir(REAL) if f.endswith(".png")):
        img = np.asarray(Image.open(os.path.join(REAL, name)).convert("RGB"))
        assert gpu_lib.jpeg_encode(img, 95) == pil_jpeg(img, 95), name


def test_enhanced_crops_encode_from_the_device_buffer(gpu_lib):
    """The boundary the reference has after SR (enhancer.py:273-278 cv2.imwrite of each enhanced crop): the crops stay in the device
    buffer ffp_sr_enhance_crops_dev filled and only JPEG bytes come back; same files as encoding the downloaded crop with Pillow."""
    import torch
    from ffp_amd import pipeline, synth
    H, W = 270, 480
    frame = synth.synthetic_frame(H, W, seed=4)
    bgr = np.ascontiguousarray(frame[..., ::-1])
    enh = gpu_lib.Enhancer(synth.rrdbnet_weights(4, 2), 4, 2, half=True)
    boxes = np.asarray([[10, 20, 42, 60], [100, 50, 147, 81], [300, 200, 324, 224]], np.int32)
    d_frame = torch.from_numpy(bgr).cuda()
    tot = pipeline.FramePipeline.sr_out_bytes(boxes, H, W)
    out = torch.zeros((tot,), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    offs = enh.enhance_crops_dev([d_frame.data_ptr()], H, W, boxes, out.data_ptr(), tot)
    host = out.cpu().numpy()
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        h4, w4 = (y2 - y1) * 4, (x2 - x1) * 4
        crop = host[offs[i]:offs[i] + h4 * w4 * 3].reshape(h4, w4, 3)
        got = gpu_lib.jpeg_encode_dev(out.data_ptr() + int(offs[i]), h4, w4, w4 * 3, 95, bgr=True)
        assert got == pil_jpeg(np.ascontiguousarray(crop[..., ::-1]), 95)
    # the whole frame's crops in one pass
    files = gpu_lib.jpeg_encode_batch_dev(out.data_ptr(), offs[:len(boxes)], (boxes[:, 3] - boxes[:, 1]) * 4, (boxes[:, 2] - boxes[:, 0]) * 4, 95, bgr=True)
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        h4, w4 = (y2 - y1) * 4, (x2 - x1) * 4
        crop = host[offs[i]:offs[i] + h4 * w4 * 3].reshape(h4, w4, 3)
        assert files[i] == pil_jpeg(np.ascontiguousarray(crop[..., ::-1]), 95)


def test_batch_of_many_sizes_matches_single_image_files(gpu_lib):
    import torch
    rng = np.random.default_rng(12)
    imgs = [rng.integers(0, 256, (int(rng.integers(1, 140)), int(rng.integers(1, 140)), 3), dtype=np.uint8) for _ in range(40)]
    imgs[3] = np.full((17, 33, 3), 255, np.uint8)
    imgs[7] = (rng.integers(0, 2, (64, 48, 3)) * 255).astype(np.uint8)
    flat = np.concatenate([im.reshape(-1) for im in imgs])
    offs = np.cumsum([0] + [im.size for im in imgs])[:-1]
    d = torch.from_numpy(flat).cuda()
    torch.cuda.synchronize()
    for q in (95, 100, 40):
        files = gpu_lib.jpeg_encode_batch_dev(d.data_ptr(), offs, [im.shape[0] for im in imgs], [im.shape[1] for im in imgs], q, bgr=False)
        for im, f in zip(imgs, files):
            assert f == pil_jpeg(im, q), (im.shape, q)
    assert gpu_lib.jpeg_encode_batch_dev(d.data_ptr(), [], [], [], 95) == []


def pil_decode(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def test_decoder_pixel_identical_to_oracle_and_pillow(gpu_lib):
    rng = np.random.default_rng(5)
    from ffp_amd import synth
    big = synth.synthetic_frame(300, 400, seed=6)
    sizes = [(17, 2), (1, 1), (2, 1), (1, 2), (3, 3), (16, 1), (9, 4), (9, 5), (20, 6)] + [(int(rng.integers(1, 120)), int(rng.integers(1, 150))) for _ in range(30)]
    for k, (h, w) in enumerate(sizes):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if k % 3 == 0 else np.ascontiguousarray(big[k:k + h, 2 * k:2 * k + w])
        for kw in ({}, {"subsampling": 0}, {"subsampling": 1}):
            b = io.BytesIO()
            Image.fromarray(img).save(b, "JPEG", quality=(95, 60, 100, 30)[k % 4], **kw)
            data = b.getvalue()
            got = gpu_lib.jpeg_decode(data)
            assert gpu_lib.jpeg_info(data) == (h, w, 3)
            assert np.array_equal(got, pil_decode(data)), (h, w, kw)
            if k < 12:
                assert np.array_equal(got, J.decode(data))
    b = io.BytesIO()
    Image.fromarray(big).convert("L").save(b, "JPEG", quality=90)
    assert gpu_lib.jpeg_info(b.getvalue())[2] == 1
    assert np.array_equal(gpu_lib.jpeg_decode(b.getvalue()), pil_decode(b.getvalue()))
    assert np.array_equal(gpu_lib.jpeg_decode(b.getvalue(), bgr=True), pil_decode(b.getvalue())[..., ::-1])


def test_decode_full_frame_into_device_memory_and_roundtrip(gpu_lib):
    """A 4K-shaped frame decoded straight into device memory feeds the detector without an upload of the pixels; decode(encode(x))
    equals Pillow's round trip; progressive files are refused loudly."""
    import torch
    from ffp_amd import synth
    frame = synth.synthetic_frame(1080, 1920, seed=10)
    data = gpu_lib.jpeg_encode(frame, 95)
    ref = pil_decode(data)
    d = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert gpu_lib.jpeg_decode_dev(data, d.data_ptr(), 1920 * 3, d.numel(), bgr=False) == (1080, 1920)
    assert np.array_equal(d.cpu().numpy(), ref)
    for name in sorted(f for f in os.listdir(REAL) if f.endswith(".png")):
        img = np.asarray(Image.open(os.path.join(REAL, name)).convert("RGB"))
        data = pil_jpeg(img, 88)
        assert np.array_equal(gpu_lib.jpeg_decode(data), pil_decode(data)), name
    b = io.BytesIO()
    Image.fromarray(frame[:64, :64]).save(b, "JPEG", progressive=True)
    with pytest.raises(RuntimeError):
        gpu_lib.jpeg_decode(b.getvalue())


def test_restart_intervals_and_cv2_shim_files(gpu_lib, tmp_path):
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (70, 90, 3), dtype=np.uint8)
    for kw in ({"restart_marker_blocks": 3}, {"restart_marker_rows": 1}):
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=90, **kw)
        assert b.getvalue().count(b"\xff\xdd") == 1
        assert np.array_equal(gpu_lib.jpeg_decode(b.getvalue()), pil_decode(b.getvalue())), kw
    # the reference's file boundary through the shim's cv2.imwrite / cv2.imread: the file on disk is what Pillow / OpenCV write
    import importlib
    import sys
    import ffp_amd.compat
    ffp_amd.compat.install()
    sys.modules.pop("cv2", None)
    cv2 = importlib.import_module("cv2")
    if getattr(cv2, "__version__", "") != "0.0-ffp-shim":
        pytest.skip("a real OpenCV is installed")
    p = str(tmp_path / "crop.jpg")
    assert cv2.imwrite(p, np.ascontiguousarray(img[..., ::-1]), [cv2.IMWRITE_JPEG_QUALITY, 95])
    with open(p, "rb") as fh:
        assert fh.read() == pil_jpeg(img, 95)
    back = cv2.imread(p)
    assert np.array_equal(back[..., ::-1], pil_decode(pil_jpeg(img, 95)))
