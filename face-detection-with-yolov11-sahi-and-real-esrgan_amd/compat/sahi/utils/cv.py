"""Image reading helper used by sahi.predict / PredictionResult (PIL; EXIF transposed, RGB)."""
import numpy as np
from PIL import Image, ImageOps


def read_image_as_pil(image, exif_fix: bool = True) -> Image.Image:
    Image.MAX_IMAGE_PIXELS = None
    if isinstance(image, Image.Image):
        pil = image
    elif isinstance(image, str):
        pil = Image.open(image)
        if exif_fix:
            pil = ImageOps.exif_transpose(pil)
        pil = pil.convert("RGB")
    elif isinstance(image, np.ndarray):
        if image.ndim == 3 and image.shape[0] < 5 and image.shape[2] > 4:   # CHW -> HWC
            image = image.transpose(1, 2, 0)
        pil = Image.fromarray(image)
    else:
        raise TypeError("read image with 'pillow' using 'Image.open()'")
    return pil


def read_image_as_array(image) -> np.ndarray:
    """HxWx3 uint8 RGB ndarray of what read_image_as_pil would return, without the PIL round trip where none is needed: an ndarray is
    used as it is (SAHI's own np.asarray(read_image_as_pil(ndarray)) is the same bytes), a baseline .jpg without an EXIF rotation is
    decoded by this build's codec (Huffman on the host, the rest on the GPU; same pixels as libjpeg — tests/test_gpu_jpeg.py)."""
    if isinstance(image, np.ndarray) and image.ndim == 3 and image.shape[2] == 3 and image.dtype == np.uint8:
        return image
    if isinstance(image, str) and image.lower().endswith((".jpg", ".jpeg")):
        try:
            from ffp_amd import _lib
            if _lib.device_count() > 0:
                with Image.open(image) as im:
                    plain = im.getexif().get(0x0112, 1) in (0, 1) and im.mode == "RGB"
                if plain:
                    with open(image, "rb") as fh:
                        return _lib.jpeg_decode(fh.read(), bgr=False)
        except Exception:        # noqa: BLE001 — progressive / CMYK / truncated files: PIL decides
            pass
    return np.asarray(read_image_as_pil(image))
