"""save_face_crops — the crop extraction that feeds SR in pipeline_v1 (utils/visualization.py:185-223), Pillow I/O.
Drawing/summary helpers are presentation code and outside the hot path."""
import os

import numpy as np
from PIL import Image


def crop_boxes(result, width, height):
    """int box, clamped to the image; empty crops dropped. Returns [(index, x1, y1, x2, y2, score)]."""
    out = []
    for i, det in enumerate(result.object_prediction_list):
        x1, y1, x2, y2 = [int(c) for c in det.bbox.to_xyxy()]
        x1, y1, x2, y2 = max(0, x1), max(0, y1), min(width, x2), min(height, y2)
        if x2 > x1 and y2 > y1:
            out.append((i, x1, y1, x2, y2, det.score.value))
    return out


def save_face_crops(image_path, result, output_dir, prefix="face_crop"):
    try:
        img = Image.open(image_path).convert("RGB")
    except Exception:
        print(f"Error: Gagal membaca gambar dari {image_path}")
        return []
    os.makedirs(output_dir, exist_ok=True)
    paths = []
    for i, x1, y1, x2, y2, score in crop_boxes(result, img.width, img.height):
        p = os.path.join(output_dir, f"{prefix}_{i + 1}_conf_{score:.2f}.jpg")
        img.crop((x1, y1, x2, y2)).save(p, quality=95)
        paths.append(p)
    return paths
