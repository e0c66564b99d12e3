"""The device-side Huffman decoder (csrc/jpeg_huff.hip: self-synchronising parallel decoding) — SURVEY.md §8 f2, the read side of the
reference's file boundary (cv2.imread, pipeline_v4_yolo/1_Inference.py:328-330). The bar is unchanged: pixels identical to
libjpeg-turbo (Pillow). On top of it: valid files really are decoded on the device (ffp_jpeg_decode_stats), whatever their sampling,
restart interval or Huffman tables, and damaged streams go to the host decoder instead of producing garbage silently."""
import io
import os

import numpy as np
import pytest
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 25          # Pillow's optimize=True encoder needs the whole file in one buffer

pytestmark = pytest.mark.gpu
REAL = os.path.join(os.path.dirname(__file__), "golden", "real")


def pil_decode(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def save(img, **kw):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def decode_on_device(gpu_lib, data):
    """decode + assert that the device decoder did it"""
    d0, f0, _ = gpu_lib.jpeg_decode_stats()
    got = gpu_lib.jpeg_decode(data)
    d1, f1, _ = gpu_lib.jpeg_decode_stats()
    assert (d1 - d0, f1 - f0) == (1, 0), "the file went to the host decoder"
    return got


def test_samplings_tables_and_restart_intervals_decode_on_the_device(gpu_lib):
    from ffp_amd import synth
    rng = np.random.default_rng(21)
    photo = synth.synthetic_frame(360, 500, seed=3)
    noise = rng.integers(0, 256, (200, 264, 3), dtype=np.uint8)
    mixed = photo.copy()
    mixed[40:240, 100:364] = noise
    n = 0
    for img in (photo, noise, mixed, photo[:9, :23], photo[:1, :1], photo[:8, :8], photo[:16, :16], photo[:17, :33]):
        for sub in (0, 1, 2):
            for q, extra in ((95, {}), (100, {}), (35, {}), (90, {"optimize": True}), (90, {"restart_marker_blocks": 1}),
                             (75, {"restart_marker_blocks": 7}), (92, {"restart_marker_rows": 1}), (60, {"restart_marker_rows": 2, "optimize": True})):
                data = save(img, quality=q, subsampling=sub, **extra)
                assert np.array_equal(decode_on_device(gpu_lib, data), pil_decode(data)), (img.shape, sub, q, extra)
                n += 1
    gray = np.asarray(Image.fromarray(mixed).convert("L"))
    for extra in ({}, {"optimize": True}, {"restart_marker_blocks": 1}, {"restart_marker_rows": 3}):
        b = io.BytesIO()
        Image.fromarray(gray).save(b, "JPEG", quality=85, **extra)
        assert np.array_equal(decode_on_device(gpu_lib, b.getvalue()), pil_decode(b.getvalue())), extra
    assert n == 8 * 3 * 8


def test_large_streams_many_workgroups(gpu_lib):
    """Streams of several MB: thousands of subsequences, hundreds of workgroups, the scan over tiles and subsequences at length; a
    high-entropy picture at quality 100 has long codes (the slow path of the table) and thousands of stuffed 0xFF bytes."""
    from ffp_amd import synth
    rng = np.random.default_rng(4)
    frame = synth.synthetic_frame(2160, 3840, seed=11)
    noisy = np.clip(frame.astype(np.int16) + rng.integers(-40, 41, frame.shape), 0, 255).astype(np.uint8)
    for img, kw in ((frame, dict(quality=95)), (noisy, dict(quality=100, subsampling=0)), (noisy[:1000, :1500], dict(quality=98, restart_marker_rows=4)),
                    (noisy[:1081, :1923], dict(quality=90, subsampling=1, optimize=True))):
        data = save(img, **kw)
        assert data.count(b"\xff\x00") > 10
        assert np.array_equal(decode_on_device(gpu_lib, data), pil_decode(data)), kw
    # the library's own files (what the shim's cv2.imwrite leaves on disk)
    data = gpu_lib.jpeg_encode(frame, 95)
    assert np.array_equal(decode_on_device(gpu_lib, data), pil_decode(data))
    for name in sorted(f for f in os.listdir(REAL) if f.endswith(".png")):
        img = np.asarray(Image.open(os.path.join(REAL, name)).convert("RGB"))
        data = save(img, quality=88)
        assert np.array_equal(decode_on_device(gpu_lib, data), pil_decode(data)), name


def test_damaged_streams_go_to_the_host_decoder(gpu_lib):
    """A stream the device decoder cannot vouch for (cut short, bytes flipped, a restart marker removed) is decoded again by the host
    decoder — same pixels as with FFP_JPEG_HOST_HUFFMAN=1, or its error — and the next valid file is unaffected."""
    from ffp_amd import synth
    img = synth.synthetic_frame(240, 320, seed=9)
    good = save(img, quality=90)
    sos = good.index(b"\xff\xda")
    cut = good[:sos + 14 + (len(good) - sos) // 2] + b"\xff\xd9"
    flipped = bytearray(good)
    for k in range(sos + 200, sos + 260):
        flipped[k] ^= 0x5A
    rst = save(img, quality=90, restart_marker_blocks=4)
    k = rst.index(b"\xff\xd3", rst.index(b"\xff\xda"))
    no_marker = rst[:k] + rst[k + 2:]
    for name, data in (("cut", cut), ("flipped", bytes(flipped)), ("marker removed", no_marker)):
        d0, f0, _ = gpu_lib.jpeg_decode_stats()
        try:
            out = gpu_lib.jpeg_decode(data)
            assert out.shape == (240, 320, 3), name
        except RuntimeError:
            pass
        d1, f1, _ = gpu_lib.jpeg_decode_stats()
        assert f1 - f0 == 1 and d1 == d0, name
    assert np.array_equal(decode_on_device(gpu_lib, good), pil_decode(good))


def test_decoder_states_need_no_extra_rounds_on_photographs(gpu_lib):
    """One blind round of synchronisation across workgroups is what the decode queues; extra rounds are legal but would mean a host
    round trip each — on ordinary pictures there are none."""
    from ffp_amd import synth
    frame = synth.synthetic_frame(1080, 1920, seed=2)
    _, _, r0 = gpu_lib.jpeg_decode_stats()
    for q in (95, 75, 50):
        data = save(frame, quality=q)
        assert np.array_equal(decode_on_device(gpu_lib, data), pil_decode(data))
    _, _, r1 = gpu_lib.jpeg_decode_stats()
    assert r1 == r0


def test_stream_that_never_self_synchronises_takes_extra_rounds(gpu_lib):
    """Flat areas coded with optimised tables are one-bit codes: a decoder that starts out of phase stays out of phase, nothing
    synchronises by itself and the exact states have to travel from workgroup to workgroup — the path behind jpeg_huff_finish's
    return code 2 (one host round trip per round). Still pixel-identical, still on the device."""
    img = np.zeros((1600, 2048, 3), np.uint8)
    img[:, :1024] = (200, 30, 90)
    img[:, 1024:] = (20, 180, 240)
    data = save(img, quality=90, subsampling=0, optimize=True)
    _, _, r0 = gpu_lib.jpeg_decode_stats()
    assert np.array_equal(decode_on_device(gpu_lib, data), pil_decode(data))
    _, _, r1 = gpu_lib.jpeg_decode_stats()
    assert r1 > r0


def test_corrupted_streams_never_hang_or_crash(gpu_lib):
    """120 damaged variants of three files (bit flips in the entropy-coded data, truncations, stray markers): every call returns — pixels
    of the right shape, from the device decoder when the stream is still a consistent baseline scan or from the host decoder otherwise,
    or the host decoder's error — and the library decodes the next valid file correctly. Streams that still pass on the device equal the
    host decoder's result by construction (same symbols, every special case flagged); spot-checked against Pillow where Pillow agrees
    to decode."""
    from ffp_amd import synth
    rng = np.random.default_rng(77)
    img = synth.synthetic_frame(200, 264, seed=12)
    files = [save(img, quality=90), save(img, quality=60, subsampling=0, optimize=True), save(img, quality=85, restart_marker_blocks=5)]
    outcomes = {"device": 0, "host": 0, "error": 0}
    for k in range(120):
        data = bytearray(files[k % 3])
        sos = data.index(b"\xff\xda") + 14
        kind = k % 4
        if kind == 0:                                                # a few flipped bits
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(sos, len(data) - 2))
                data[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:                                              # cut short (with and without an EOI behind the cut)
            cut = int(rng.integers(sos + 1, len(data) - 2))
            data = data[:cut] + (bytearray(b"\xff\xd9") if k % 8 == 1 else bytearray())
        elif kind == 2:                                              # a marker where data should be
            pos = int(rng.integers(sos, len(data) - 4))
            data[pos:pos + 2] = bytes([0xFF, int(rng.choice([0xD0, 0xD5, 0xD9, 0xC4, 0x01]))])
        else:                                                        # a run of random bytes
            pos = int(rng.integers(sos, len(data) - 40))
            data[pos:pos + 32] = rng.integers(0, 256, 32, dtype=np.uint8).tobytes()
        d0, f0, _ = gpu_lib.jpeg_decode_stats()
        try:
            out = gpu_lib.jpeg_decode(bytes(data))
            assert out.shape == (200, 264, 3) and out.dtype == np.uint8
            d1, f1, _ = gpu_lib.jpeg_decode_stats()
            outcomes["device" if d1 > d0 else "host"] += 1
            if d1 > d0:
                try:
                    ref = pil_decode(bytes(data))
                except Exception:
                    ref = None
                if ref is not None and ref.shape == out.shape:
                    assert np.array_equal(out, ref), k
        except RuntimeError:
            outcomes["error"] += 1
    assert sum(outcomes.values()) == 120 and outcomes["host"] + outcomes["error"] > 0
    assert np.array_equal(decode_on_device(gpu_lib, files[0]), pil_decode(files[0]))
