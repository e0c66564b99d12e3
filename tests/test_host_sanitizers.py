"""Host-side sanitizer run: the parsers of untrusted bytes — FFPW containers (`csrc/weights.cpp`) and JPEG markers /
Huffman tables / entropy-coded data (`csrc/jpeg_dec.cpp`, behind `ffp_jpeg_info`, `ffp_jpeg_decode*` and the cv2 shim's `imread`) —
are built with the HOST compiler under AddressSanitizer + UBSan (`build.py --asan`, no device code) and fed valid files, every header
truncation, seeded mutations and hand-made malformed streams (over-subscribed DHT, all-ones codes, fill bytes at the end of the
stream, RGB-coded files). Any out-of-bounds access aborts the driver. CPU only."""
import io
import os
import subprocess

import numpy as np
import pytest

import ffp_amd  # noqa: F401
from ffp_amd import build as ffp_build
from ffp_amd import weights_io

PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def driver():
    try:
        return ffp_build.build_asan(verbose=False)
    except Exception as e:                                      # no host clang with sanitizer runtimes on this box
        pytest.skip(f"sanitizer build unavailable: {e}")


def _jpegs(tmp_path):
    rng = np.random.default_rng(5)
    img = (rng.random((67, 91, 3)) * 255).astype(np.uint8)
    img[10:40, 20:70] = np.linspace(0, 255, 50, dtype=np.uint8)[None, :, None]
    out = []
    for name, arr, kw in [("a420", img, dict(quality=90, subsampling=2)), ("a444", img, dict(quality=75, subsampling=0)),
                          ("a422", img, dict(quality=95, subsampling=1)), ("opt", img, dict(quality=85, optimize=True)),
                          ("rst", img, dict(quality=80, subsampling=2, restart_marker_blocks=3)), ("gray", img[..., 0], dict(quality=80)),
                          ("tiny", img[:1, :1], dict(quality=95))]:
        p = tmp_path / f"{name}.jpg"
        try:
            PIL.fromarray(arr).save(p, **kw)
        except TypeError:
            kw.pop("restart_marker_blocks", None)
            PIL.fromarray(arr).save(p, **kw)
        out.append(str(p))
    return out


def test_parsers_survive_malformed_input_under_asan_ubsan(driver, tmp_path):
    rng = np.random.default_rng(0)
    tensors = {f"model.{i}.conv.weight": rng.standard_normal((4, 3, 3, 3)).astype(np.float32) for i in range(5)}
    tensors["model.0.conv.bias"] = np.zeros((7,), np.float32)
    w = tmp_path / "w.ffpw"
    w.write_bytes(weights_io.pack(tensors))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([driver, "jpeg", *_jpegs(tmp_path), "ffpw", str(w)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "no sanitizer report" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def _with_dht_before_sos(good: bytes, payload: bytes) -> bytes:
    i = 2
    while good[i] == 0xFF and good[i + 1] != 0xDA:
        i += 2 + ((good[i + 2] << 8) | good[i + 3])
    seg = bytes([0xFF, 0xC4]) + (len(payload) + 2).to_bytes(2, "big") + payload
    return good[:i] + seg + good[i:]


def test_c_abi_rejects_oversubscribed_huffman_tables(tmp_path):
    """The shipped library (not the sanitizer build): `ffp_jpeg_info` — header parsing needs no device — returns FFP_ERR_ARG for the
    DHT that overflowed `HuffTab::build` in round 2, for codes that use the all-ones pattern, and for fill bytes
    that run into the end of the stream."""
    from ffp_amd import _lib
    good = open(_jpegs(tmp_path)[0], "rb").read()
    assert _lib.jpeg_info(good) == (67, 91, 3)
    for tc in (0, 1):
        over = bytes([tc << 4, 200] + [0] * 15) + bytes(range(200))
        three = bytes([tc << 4, 3] + [0] * 15) + bytes([1, 2, 3])
        ones = bytes([tc << 4, 2] + [0] * 15) + bytes([0, 1])
        for payload in (over, three, ones):
            with pytest.raises(_lib.FfpError, match="Huffman"):
                _lib.jpeg_info(_with_dht_before_sos(good, payload))
    with pytest.raises(_lib.FfpError):
        _lib.jpeg_info(good[:2] + b"\xff" * 9)
    with pytest.raises(_lib.FfpError):
        _lib.jpeg_info(good[:2] + b"\xff" * 9 + b"\xdb")
    adobe = bytes([0xFF, 0xEE, 0, 14]) + b"Adobe" + bytes([0, 100, 0, 0, 0, 0, 0])
    with pytest.raises(_lib.FfpError, match="RGB"):
        _lib.jpeg_info(good[:2] + adobe + good[2:])
