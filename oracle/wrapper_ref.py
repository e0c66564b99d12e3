"""ORACLE (test infrastructure): the SAHI plugin adapter's result conversion and keypoint side channel, plain Python.

Follows /root/reference/utils/yolo_wrapper.py:
  * `convert` — `_create_object_prediction_list_from_original_predictions` (:84-166): per box `float(conf)`, `xyxy.astype(int)`
    (truncation towards zero), ObjectPrediction in slice coordinates carrying the shift, NO second confidence filter, keypoints
    copied, x / y moved by the shift and stored in the cache under "x1_y1_x2_y2" of the SHIFTED int box;
  * `attach` — `attach_keypoints_to_predictions` (:168-200): exact key first, else the cache entry of highest IoU if that is > 0.5
    (strict), scanning the cache in insertion order (the first entry wins ties because only a strictly larger IoU replaces it);
  * `iou` — `_calculate_iou` (:202-217).
The reference pulls box / keypoint arrays off torch tensors; here they are numpy arrays. `iou` and `attach` are PINNED: tests/test_wrapper_pinned.py
checks them against the outputs of the reference's own methods on seeded inputs (tests/golden/make_wrapper_fixtures.py); `convert` needs sahi's
ObjectPrediction, which is not in the tree, and stays unpinned.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def iou(b1: Sequence[float], b2: Sequence[float]) -> float:
    xa, ya, xb, yb = max(b1[0], b2[0]), max(b1[1], b2[1]), min(b1[2], b2[2]), min(b1[3], b2[3])
    if xb < xa or yb < ya:
        return 0.0
    inter = (xb - xa) * (yb - ya)
    union = (b1[2] - b1[0]) * (b1[3] - b1[1]) + (b2[2] - b2[0]) * (b2[3] - b2[1]) - inter
    return inter / union if union > 0 else 0.0


def convert(xyxy: np.ndarray, conf: np.ndarray, kpts: Optional[np.ndarray], shift: Sequence[int], full_shape: Optional[Sequence[int]],
            cache: Dict[str, np.ndarray]) -> List[Tuple[List[int], float, List[int]]]:
    """-> [(bbox in slice coords after sahi's clipping, score, shift)], and fills `cache` like the wrapper does."""
    out = []
    for i in range(len(xyxy)):
        x1, y1, x2, y2 = (int(v) for v in np.asarray(xyxy[i]).astype(int))
        # sahi ObjectAnnotation: clip to >= 0 and to full_shape (docs sahi/prediction.py via sahi.annotation)
        bx = [max(x1, 0), max(y1, 0), min(x2, full_shape[1]) if full_shape else x2, min(y2, full_shape[0]) if full_shape else y2]
        out.append((bx, float(conf[i]), list(shift)))
        if kpts is not None and i < len(kpts):
            k = np.array(kpts[i], dtype=np.float32, copy=True)
            k[:, 0] += shift[0]
            k[:, 1] += shift[1]
            cache[f"{x1 + shift[0]}_{y1 + shift[1]}_{x2 + shift[0]}_{y2 + shift[1]}"] = k
    return out


def attach(boxes: Sequence[Sequence[int]], cache: Dict[str, np.ndarray]) -> List[Optional[np.ndarray]]:
    """For every merged box (voc / xyxy ints): the keypoints the wrapper would attach, or None."""
    res = []
    for b in boxes:
        key = f"{b[0]}_{b[1]}_{b[2]}_{b[3]}"
        if key in cache:
            res.append(cache[key])
            continue
        best, best_k = 0.0, None
        for k, v in cache.items():
            c = [int(float(x)) for x in k.split("_")]
            u = iou(b, c)
            if u > best:
                best, best_k = u, v
        res.append(best_k if (best > 0.5 and best_k is not None) else None)
    return res
