"""sahi.slicing subset: grid computation in libffp.so (ffp_slice_bboxes), slices as ndarray views."""
from __future__ import annotations

from typing import List, Optional

import numpy as np

import ffp_amd  # noqa: F401
from ffp_amd import _lib
from sahi.utils.cv import read_image_as_array


def get_slice_bboxes(image_height: int, image_width: int, slice_height: Optional[int] = None, slice_width: Optional[int] = None,
                     auto_slice_resolution: bool = True, overlap_height_ratio: float = 0.2, overlap_width_ratio: float = 0.2) -> List[List[int]]:
    if not (slice_height and slice_width):
        raise ValueError("slice_height and slice_width must be given (auto slice resolution is not used by the reference's callers)")
    return _lib.slice_bboxes(image_height, image_width, slice_height, slice_width, overlap_height_ratio, overlap_width_ratio).tolist()


class SliceImageResult:
    def __init__(self, original_image_size, image_dir=None):
        self.original_image_height, self.original_image_width = original_image_size
        self.image_dir = image_dir
        self.images: List[np.ndarray] = []
        self.starting_pixels: List[List[int]] = []

    def __len__(self):
        return len(self.images)


def slice_image(image, output_file_name=None, output_dir=None, slice_height=None, slice_width=None, overlap_height_ratio=0.2,
                overlap_width_ratio=0.2, auto_slice_resolution=True, **_ignored) -> SliceImageResult:
    arr = read_image_as_array(image)
    h, w = arr.shape[:2]
    res = SliceImageResult([h, w], output_dir)
    for x0, y0, x1, y1 in get_slice_bboxes(h, w, slice_height, slice_width, auto_slice_resolution, overlap_height_ratio, overlap_width_ratio):
        res.images.append(arr[y0:y1, x0:x1])
        res.starting_pixels.append([x0, y0])
    res.full_image = arr
    return res
