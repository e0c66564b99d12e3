"""Minimal `cv2` stand-in for the reference's scripts when OpenCV is not installed (it is not a dependency of this
build): the image-file calls they make — `cv2.imread` for the image size (pipeline_v4_yolo/app_yolo_sahi.py:37-42),
`cv2.imwrite` with `[cv2.IMWRITE_JPEG_QUALITY, q]` (utils/enhancer.py:273-278), `cv2.cvtColor` BGR<->RGB, `cv2.resize` —
on Pillow + numpy, BGR arrays like OpenCV's. A real OpenCV further down sys.path always wins: this module then
re-exports it untouched. Drawing primitives are not provided (utils.visualization of this build draws with Pillow).

`.jpg` files go through this build's JPEG codec on the GPU (csrc/jpeg.hip: ffp_jpeg_decode / ffp_jpeg_encode) whenever a device is
visible; its output is byte-identical (files) and pixel-identical (decoded arrays) to the libjpeg-turbo inside OpenCV and Pillow
(tests/test_gpu_jpeg.py) for YCbCr / grayscale baseline files, so the choice changes the time, not one bit of the result. Files the
device codec does not read (progressive, RGB-coded with Adobe transform 0 or R/G/B component ids) and machines without a GPU use
Pillow, and say so once on stderr. Like `cv2.imread`, `imread` applies the EXIF orientation tag (not with IMREAD_UNCHANGED).
"""
import importlib.machinery
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _real_cv2():
    for p in sys.path:
        if not p or os.path.abspath(p) == _here:
            continue
        spec = importlib.machinery.PathFinder.find_spec("cv2", [p])
        if spec is not None and spec.origin and os.path.dirname(os.path.dirname(os.path.abspath(spec.origin))) != _here:
            return spec
    return None


_spec = _real_cv2()
if _spec is not None:                                   # OpenCV is installed: be it
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[__name__] = _mod
    _spec.loader.exec_module(_mod)
else:
    import numpy as np
    from PIL import Image

    __version__ = "0.0-ffp-shim"
    IMREAD_COLOR, IMREAD_UNCHANGED = 1, -1
    IMWRITE_JPEG_QUALITY, IMWRITE_PNG_COMPRESSION = 1, 16
    COLOR_BGR2RGB, COLOR_RGB2BGR = 4, 4
    INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4

    _state = {"gpu": None, "told": False}

    def _gpu_codec():
        if _state["gpu"] is None:
            try:
                from ffp_amd import _lib
                _state["gpu"] = _lib if _lib.device_count() > 0 else False
            except Exception:
                _state["gpu"] = False
        return _state["gpu"]

    def _tell(why):
        if not _state["told"]:
            _state["told"] = True
            print(f"cv2 shim: JPEG through Pillow ({why}); same bytes, host speed", file=sys.stderr)

    def _is_jpeg(path):
        return os.path.splitext(str(path))[1].lower() in (".jpg", ".jpeg")

    def _exif_orientation(path):
        """EXIF orientation tag (1..8; 1 = as stored). Reads the APP1 segment only, no pixel decoding."""
        try:
            with Image.open(path) as im:
                o = int(im.getexif().get(0x0112, 1))
            return o if 1 <= o <= 8 else 1
        except Exception:
            return 1

    def _oriented(a, o):
        """What OpenCV's imread does with the EXIF orientation unless IMREAD_IGNORE_ORIENTATION / IMREAD_UNCHANGED is set."""
        if o == 2:
            a = a[:, ::-1]
        elif o == 3:
            a = a[::-1, ::-1]
        elif o == 4:
            a = a[::-1]
        elif o == 5:
            a = a.transpose(1, 0, 2)
        elif o == 6:
            a = a.transpose(1, 0, 2)[:, ::-1]
        elif o == 7:
            a = a.transpose(1, 0, 2)[::-1, ::-1]
        elif o == 8:
            a = a.transpose(1, 0, 2)[::-1]
        return np.ascontiguousarray(a)

    def imread(path, flags=IMREAD_COLOR):
        """HxWx3 uint8 BGR with the EXIF orientation applied (OpenCV's default), or None when the file cannot be read (OpenCV's
        convention: no exception)."""
        try:
            o = _exif_orientation(path) if flags != IMREAD_UNCHANGED else 1
            if _is_jpeg(path):
                codec = _gpu_codec()
                if codec:
                    with open(path, "rb") as fh:
                        data = fh.read()
                    try:
                        return _oriented(codec.jpeg_decode(data, bgr=True), o)
                    except codec.FfpError as e:               # e.g. a progressive or an RGB-coded (Adobe transform 0) file
                        _tell(str(e))
                else:
                    _tell("no GPU visible")
            return _oriented(np.asarray(Image.open(path).convert("RGB"))[..., ::-1], o)
        except Exception:
            return None

    def imwrite(path, img, params=None):
        """JPEG quality from [IMWRITE_JPEG_QUALITY, q] (OpenCV default 95). Returns success like OpenCV."""
        try:
            q = 95
            if params:
                for k, v in zip(params[0::2], params[1::2]):
                    if k == IMWRITE_JPEG_QUALITY:
                        q = int(v)
            a = np.ascontiguousarray(img)
            if _is_jpeg(path) and a.ndim == 3 and a.shape[2] == 3 and a.dtype == np.uint8:
                codec = _gpu_codec()
                if codec:
                    data = codec.jpeg_encode(a, q, bgr=True)
                    with open(path, "wb") as fh:
                        fh.write(data)
                    return True
                _tell("no GPU visible")
            pil = Image.fromarray(a[..., ::-1] if a.ndim == 3 and a.shape[2] == 3 else a)
            if _is_jpeg(path):
                pil.save(path, quality=q)
            else:
                pil.save(path)
            return True
        except Exception:
            return False

    def cvtColor(img, code):
        if code != COLOR_BGR2RGB:
            raise NotImplementedError("cv2 shim: only BGR<->RGB")
        return np.ascontiguousarray(img[..., ::-1])

    def resize(img, dsize, fx=0, fy=0, interpolation=INTER_LINEAR):
        w, h = dsize if dsize and dsize[0] > 0 else (int(round(img.shape[1] * fx)), int(round(img.shape[0] * fy)))
        mode = {INTER_NEAREST: Image.NEAREST, INTER_LINEAR: Image.BILINEAR, INTER_CUBIC: Image.BICUBIC, INTER_AREA: Image.BOX,
                INTER_LANCZOS4: Image.LANCZOS}[interpolation]
        return np.asarray(Image.fromarray(np.ascontiguousarray(img)).resize((w, h), mode))
