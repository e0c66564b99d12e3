"""oracle/jpeg_ref.py against Pillow's libjpeg-turbo (the codec family OpenCV bundles for the reference's cv2.imwrite / imread):
the encoder must produce the identical FILE, byte for byte, for any size and quality."""
import io
import os

import numpy as np
import pytest
from PIL import Image

from oracle import jpeg_ref as J

REAL = os.path.join(os.path.dirname(__file__), "golden", "real")


def pil_jpeg(img, q):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=q)
    return b.getvalue()


def test_encoder_is_byte_identical_to_libjpeg_turbo_on_seeded_images():
    rng = np.random.default_rng(0)
    smooth = np.clip(np.cumsum(np.cumsum(rng.normal(0, 2.0, (200, 260, 3)), 0), 1) * 0.02 + 128, 0, 255).astype(np.uint8)
    n = 0
    for k in range(60):
        h, w = int(rng.integers(1, 80)), int(rng.integers(1, 100))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if k % 2 else smooth[k:k + h, 2 * k:2 * k + w]
        for q in (95, 75, 100, 20)[: 2 + k % 3]:
            assert J.encode(img, q) == pil_jpeg(img, q), (img.shape, q)
            n += 1
    assert n > 100


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(REAL) if f.endswith(".png")) if os.path.isdir(REAL) else [])
def test_encoder_is_byte_identical_on_real_photographs(name):
    img = np.asarray(Image.open(os.path.join(REAL, name)).convert("RGB"))
    assert J.encode(img, 95) == pil_jpeg(img, 95)


def test_header_and_tables():
    hd = J.header(37, 53, 95)
    assert hd[:4] == b"\xff\xd8\xff\xe0" and hd[6:11] == b"JFIF\x00"
    ql, qc = J.quant_tables(95)
    assert ql[0] == 2 and ql.max() <= 255 and qc.min() >= 1 and list(J.quant_tables(100)[0][:4]) == [1, 1, 1, 1]
    assert J.huff_codes(*J.DC_LUMA)[0] == (0, 2) and J.huff_codes(*J.AC_LUMA)[0x00] == (0b1010, 4) and J.huff_codes(*J.AC_LUMA)[0xF0] == (0b11111111001, 11)


def pil_decode(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def test_decoder_is_pixel_identical_to_libjpeg_turbo():
    """4:2:0 / 4:4:4 / 4:2:2 / grayscale, tiny and odd sizes (replicated instead of filtered chroma below 5 pixels of width)."""
    rng = np.random.default_rng(3)
    smooth = np.clip(np.cumsum(np.cumsum(rng.normal(0, 2.0, (200, 260, 3)), 0), 1) * 0.02 + 128, 0, 255).astype(np.uint8)
    sizes = [(17, 2), (1, 1), (2, 1), (1, 2), (3, 3), (16, 1), (33, 2), (9, 4), (9, 5), (20, 6)] + [(int(rng.integers(1, 70)), int(rng.integers(1, 90))) for _ in range(25)]
    for k, (h, w) in enumerate(sizes):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if k % 2 else smooth[k:k + h, 2 * k:2 * k + w]
        for kw in ({}, {"subsampling": 0}, {"subsampling": 1}):
            b = io.BytesIO()
            Image.fromarray(img).save(b, "JPEG", quality=(95, 60, 100)[k % 3], **kw)
            assert np.array_equal(J.decode(b.getvalue()), pil_decode(b.getvalue())), (h, w, kw)
    b = io.BytesIO()
    Image.fromarray(smooth).convert("L").save(b, "JPEG", quality=90)
    assert np.array_equal(J.decode(b.getvalue()), pil_decode(b.getvalue()))


def test_decoder_restart_intervals():
    img = np.random.default_rng(8).integers(0, 256, (70, 90, 3), dtype=np.uint8)
    for kw in ({"restart_marker_blocks": 3}, {"restart_marker_rows": 1}):
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=90, **kw)
        assert np.array_equal(J.decode(b.getvalue()), pil_decode(b.getvalue()))


def test_decoder_rejects_progressive_files():
    b = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(b, "JPEG", progressive=True)
    with pytest.raises(ValueError):
        J.decode(b.getvalue())
