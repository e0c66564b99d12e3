"""ORACLE (test infrastructure): SAHI slicing, shifting and post-processing arithmetic + orchestration.

Orchestration follows the vendored text /root/reference/docs sahi/predict.py:63-139 (get_prediction) and
:142-345 (get_sliced_prediction). The arithmetic files (`sahi/slicing.py`, `sahi/postprocess/combine.py`,
`sahi/postprocess/utils.py`, `sahi/annotation.py`) are NOT under /root/reference — `sahi==0.11.34`
(/root/reference/requirements.txt:140) is not installed; their published algorithms are restated per SURVEY.md
Appendix C. Parity unpinned (oracle/__init__.py).

Determinism note (stated in DESIGN.md): on exact score ties the upstream sort order is implementation defined;
here ties are broken by ascending original index (the earlier box is processed first).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------------
# C.1 slicing
# ------------------------------------------------------------------------------------------------
def get_slice_bboxes(image_height: int, image_width: int, slice_height: int, slice_width: int,
                     overlap_height_ratio: float = 0.2, overlap_width_ratio: float = 0.2) -> List[List[int]]:
    """sahi.slicing.get_slice_bboxes with explicit slice size (used by docs sahi/predict.py:229-238)."""
    slice_bboxes = []
    y_max = y_min = 0
    y_overlap = int(overlap_height_ratio * slice_height)
    x_overlap = int(overlap_width_ratio * slice_width)
    while y_max < image_height:
        x_min = x_max = 0
        y_max = y_min + slice_height
        while x_max < image_width:
            x_max = x_min + slice_width
            if y_max > image_height or x_max > image_width:
                xmax = min(image_width, x_max)
                ymax = min(image_height, y_max)
                xmin = max(0, xmax - slice_width)
                ymin = max(0, ymax - slice_height)
                slice_bboxes.append([xmin, ymin, xmax, ymax])
            else:
                slice_bboxes.append([x_min, y_min, x_max, y_max])
            x_min = x_max - x_overlap
        y_min = y_max - y_overlap
    return slice_bboxes


# ------------------------------------------------------------------------------------------------
# C.2 value types (just enough of sahi.annotation / docs sahi/prediction.py:13-163)
# ------------------------------------------------------------------------------------------------
class Det:
    """One ObjectPrediction: bbox xyxy (clipped like sahi.annotation.BoundingBox), score, category id,
    shift_amount, full_shape, + index-carried keypoints (a build deviation)."""
    __slots__ = ("bbox", "score", "cat", "shift", "full_shape", "kpts", "src")

    def __init__(self, bbox, score, cat=0, shift=(0, 0), full_shape=None, kpts=None, src=-1):
        x1, y1, x2, y2 = bbox
        if x1 < 0 or y1 < 0 or x2 < 0 or y2 < 0:
            raise ValueError("Box coords must be positive")        # sahi BoundingBox contract
        if full_shape is not None:                                    # ObjectAnnotation clips to full_shape
            x2 = min(x2, full_shape[1]); y2 = min(y2, full_shape[0])
        self.bbox = [max(x1, 0), max(y1, 0), x2, y2]
        self.score, self.cat = float(score), int(cat)
        self.shift, self.full_shape = list(shift), (list(full_shape) if full_shape is not None else None)
        self.kpts, self.src = kpts, src

    def shifted(self) -> "Det":
        """docs sahi/prediction.py:94-120 get_shifted_object_prediction: bbox + shift, shift reset, full_shape None."""
        sx, sy = self.shift
        b = [self.bbox[0] + sx, self.bbox[1] + sy, self.bbox[2] + sx, self.bbox[3] + sy]
        return Det(b, self.score, self.cat, (0, 0), None, self.kpts, self.src)

    def row(self) -> List[float]:
        return [*self.bbox, self.score, self.cat]


# ------------------------------------------------------------------------------------------------
# C.3 matching (float32 tensors upstream)
# ------------------------------------------------------------------------------------------------
def _metric_vs_rest(b: np.ndarray, areas: np.ndarray, i: int, rest: np.ndarray, metric: str) -> np.ndarray:
    xx1 = np.maximum(b[rest, 0], b[i, 0]); yy1 = np.maximum(b[rest, 1], b[i, 1])
    xx2 = np.minimum(b[rest, 2], b[i, 2]); yy2 = np.minimum(b[rest, 3], b[i, 3])
    w = np.maximum(xx2 - xx1, np.float32(0)); h = np.maximum(yy2 - yy1, np.float32(0))
    inter = w * h
    with np.errstate(divide="ignore", invalid="ignore"):
        if metric == "IOU":
            return inter / ((areas[rest] - inter) + areas[i])
        if metric == "IOS":
            return inter / np.minimum(areas[rest], areas[i])
    raise ValueError(metric)


def _order_desc(scores: np.ndarray) -> np.ndarray:
    return np.argsort(-scores, kind="stable")


def nms(rows: np.ndarray, match_metric: str = "IOU", match_threshold: float = 0.5) -> List[int]:
    """sahi.postprocess.combine.nms: greedy by score desc; a kept box removes every remaining box whose metric is
    NOT < threshold (i.e. >= thr, NaN counts as a match). Returns keep indices in processing order."""
    b = rows[:, :4].astype(np.float32)
    scores = rows[:, 4].astype(np.float32)
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    order = _order_desc(scores)
    thr = np.float32(match_threshold)
    keep = []
    while order.size:
        i = order[0]
        keep.append(int(i))
        order = order[1:]
        if not order.size:
            break
        m = _metric_vs_rest(b, areas, i, order, match_metric)
        order = order[m < thr]
    return keep


def greedy_nmm(rows: np.ndarray, match_metric: str = "IOU", match_threshold: float = 0.5) -> Dict[int, List[int]]:
    """sahi.postprocess.combine.greedy_nmm: as nms, but remembers for each keeper the boxes it absorbed
    (matched against the ORIGINAL keeper box), in descending score order. Insertion order = processing order."""
    b = rows[:, :4].astype(np.float32)
    scores = rows[:, 4].astype(np.float32)
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    order = _order_desc(scores)
    thr = np.float32(match_threshold)
    keep_to_merge: Dict[int, List[int]] = {}
    while order.size:
        i = order[0]
        order = order[1:]
        if not order.size:
            keep_to_merge[int(i)] = []
            break
        m = _metric_vs_rest(b, areas, i, order, match_metric)
        unmatched = m < thr
        keep_to_merge[int(i)] = [int(j) for j in order[~unmatched]]
        order = order[unmatched]
    return keep_to_merge


def batched(fn, rows: np.ndarray, match_metric: str, match_threshold: float):
    """batched_nms / batched_greedy_nmm: run per category id (ascending), indices mapped back."""
    cats = rows[:, 5]
    if fn is nms:
        keep: List[int] = []
        for c in np.unique(cats):
            idx = np.nonzero(cats == c)[0]
            keep += [int(idx[k]) for k in nms(rows[idx], match_metric, match_threshold)]
        keep.sort(key=lambda k: (-np.float32(rows[k, 4]), k))
        return keep
    out: Dict[int, List[int]] = {}
    for c in np.unique(cats):
        idx = np.nonzero(cats == c)[0]
        for k, v in greedy_nmm(rows[idx], match_metric, match_threshold).items():
            out[int(idx[k])] = [int(idx[j]) for j in v]
    return out


# ------------------------------------------------------------------------------------------------
# C.4 merge (float64 numpy upstream: sahi.postprocess.utils)
# ------------------------------------------------------------------------------------------------
def _area(b):
    return float((b[2] - b[0]) * (b[3] - b[1]))


def _inter(b1, b2):
    w = min(b1[2], b2[2]) - max(b1[0], b2[0])
    h = min(b1[3], b2[3]) - max(b1[1], b2[1])
    return float(max(w, 0) * max(h, 0))


def has_match(d1: Det, d2: Det, match_type: str, match_threshold: float) -> bool:
    a1, a2, it = _area(d1.bbox), _area(d2.bbox), _inter(d1.bbox, d2.bbox)
    with np.errstate(divide="ignore", invalid="ignore"):
        if match_type == "IOU":
            v = np.float64(it) / np.float64(a1 + a2 - it)
        elif match_type == "IOS":
            v = np.float64(it) / np.float64(min(a1, a2))
        else:
            raise ValueError(match_type)
    return bool(v > match_threshold)


def merge_pair(d1: Det, d2: Det) -> Det:
    """merge_object_prediction_pair: union box, max score, category (and here: keypoints) of the higher score,
    shift/full_shape of the first."""
    b = [min(d1.bbox[0], d2.bbox[0]), min(d1.bbox[1], d2.bbox[1]), max(d1.bbox[2], d2.bbox[2]), max(d1.bbox[3], d2.bbox[3])]
    hi = d1 if d1.score > d2.score else d2
    return Det(b, max(d1.score, d2.score), hi.cat, d1.shift, d1.full_shape, hi.kpts, hi.src)


def postprocess(dets: List[Det], ptype: str = "GREEDYNMM", metric: str = "IOS", thr: float = 0.5,
                class_agnostic: bool = False) -> List[Det]:
    """NMSPostprocess / GreedyNMMPostprocess.__call__ (selected at docs sahi/predict.py:254-259)."""
    if not dets:
        return []
    rows = np.asarray([d.row() for d in dets], np.float32)
    if ptype == "NMS":
        keep = nms(rows, metric, thr) if class_agnostic else batched(nms, rows, metric, thr)
        return [dets[k] for k in keep]
    if ptype == "GREEDYNMM":
        k2m = greedy_nmm(rows, metric, thr) if class_agnostic else batched(greedy_nmm, rows, metric, thr)
        out = []
        work = list(dets)
        for k, ms in k2m.items():
            for m in ms:
                if has_match(work[k], work[m], metric, thr):
                    work[k] = merge_pair(work[k], work[m])
            out.append(work[k])
        return out
    raise ValueError(f"postprocess_type {ptype} not restated")


# ------------------------------------------------------------------------------------------------
# orchestration
# ------------------------------------------------------------------------------------------------
def get_prediction(image: np.ndarray, predict_fn, shift_amount=(0, 0), full_shape=None) -> List[Det]:
    """docs sahi/predict.py:63-139 with the wrapper's conversion (utils/yolo_wrapper.py:84-166) inlined:
    boxes int-truncated (astype(int)), keypoints float + shift, no second confidence filter."""
    res = predict_fn(np.ascontiguousarray(image))
    if full_shape is None:
        full_shape = [image.shape[0], image.shape[1]]
    out = []
    for i in range(len(res)):
        x1, y1, x2, y2 = res.xyxy[i].astype(int)
        k = res.kpts[i].copy()
        k[:, 0] += shift_amount[0]
        k[:, 1] += shift_amount[1]
        out.append(Det([int(x1), int(y1), int(x2), int(y2)], float(res.conf[i]), 0, shift_amount, full_shape, k))
    return out


def get_sliced_prediction(image: np.ndarray, predict_fn, slice_height: int, slice_width: int,
                          overlap_height_ratio: float = 0.2, overlap_width_ratio: float = 0.2,
                          perform_standard_pred: bool = True, postprocess_type: str = "GREEDYNMM",
                          postprocess_match_metric: str = "IOS", postprocess_match_threshold: float = 0.5,
                          postprocess_class_agnostic: bool = False) -> List[Det]:
    """docs sahi/predict.py:142-345 for an in-memory RGB ndarray (merge_buffer_length=None)."""
    H, W = image.shape[:2]
    boxes = get_slice_bboxes(H, W, slice_height, slice_width, overlap_height_ratio, overlap_width_ratio)
    dets: List[Det] = []
    for (x0, y0, x1, y1) in boxes:
        for d in get_prediction(image[y0:y1, x0:x1], predict_fn, [x0, y0], [H, W]):
            dets.append(d.shifted())
    if len(boxes) > 1 and perform_standard_pred:
        dets.extend(get_prediction(image, predict_fn, [0, 0], [H, W]))
    if len(dets) > 1:
        dets = postprocess(dets, postprocess_type, postprocess_match_metric, postprocess_match_threshold,
                           postprocess_class_agnostic)
    return dets
