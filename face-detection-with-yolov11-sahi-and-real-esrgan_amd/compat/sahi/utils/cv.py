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
