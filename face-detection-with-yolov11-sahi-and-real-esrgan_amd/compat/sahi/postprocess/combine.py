"""PostprocessPredictions family (selected at docs sahi/predict.py:44-49): the matching/merging runs in libffp.so
(ffp_merge, csrc/merge.hip); this file only converts between ObjectPrediction lists and rows."""
from __future__ import annotations

from typing import List

import numpy as np

import ffp_amd  # noqa: F401
from ffp_amd import _lib
from sahi.prediction import ObjectPrediction


def _rows(preds: List[ObjectPrediction]) -> np.ndarray:
    r = np.zeros((len(preds), 6), np.float32)
    for i, p in enumerate(preds):
        r[i, :4] = p.bbox.to_xyxy()
        r[i, 4] = p.score.value
        r[i, 5] = p.category.id
    return r


class PostprocessPredictions:
    _type = None

    def __init__(self, match_threshold: float = 0.5, match_metric: str = "IOU", class_agnostic: bool = True):
        self.match_threshold, self.class_agnostic, self.match_metric = match_threshold, class_agnostic, match_metric
        if match_metric not in ("IOU", "IOS"):
            raise ValueError(f"'match_metric' should be one of ['IOU', 'IOS'] but given as {match_metric}")

    def __call__(self, object_predictions: List[ObjectPrediction]) -> List[ObjectPrediction]:
        if self._type is None:
            raise NotImplementedError()
        if not object_predictions:
            return []
        rows = _rows(object_predictions)
        out, src = _lib.merge(rows, self._type, self.match_metric, self.match_threshold, self.class_agnostic)
        res = []
        for r, s in zip(out, src):
            p0 = object_predictions[int(s)]
            if self._type == "NMS":
                res.append(p0)
                continue
            first = p0          # merged box: union, max score, category (and keypoints) of the higher-scored source
            m = ObjectPrediction(bbox=[int(r[0]), int(r[1]), int(r[2]), int(r[3])] if float(r[0]).is_integer() else r[:4].tolist(),
                                 score=float(r[4]), category_id=first.category.id, category_name=first.category.name,
                                 shift_amount=first.bbox.shift_amount, full_shape=first.full_shape)
            if hasattr(first, "keypoints"):
                m.keypoints = first.keypoints
            res.append(m)
        return res


class NMSPostprocess(PostprocessPredictions):
    _type = "NMS"


class GreedyNMMPostprocess(PostprocessPredictions):
    _type = "GREEDYNMM"


class NMMPostprocess(PostprocessPredictions):
    def __call__(self, object_predictions):
        raise NotImplementedError("NMM (non-greedy) is not on the reference's path (apps use GREEDYNMM, eval uses NMS)")


class LSNMSPostprocess(PostprocessPredictions):
    def __call__(self, object_predictions):
        raise NotImplementedError("LSNMS needs the external lsnms package upstream; not on the reference's path")
