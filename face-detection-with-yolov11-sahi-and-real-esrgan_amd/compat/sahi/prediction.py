"""ObjectPrediction / PredictionScore / PredictionResult with the reference's signatures (docs sahi/prediction.py:13-243)."""
from __future__ import annotations

import copy
from typing import Any, Dict, List, Optional, Union

import numpy as np

from sahi.annotation import ObjectAnnotation
from sahi.utils.cv import read_image_as_pil


class PredictionScore:
    def __init__(self, value):
        if type(value).__module__ == "numpy":
            value = copy.deepcopy(value).tolist()
        self.value = value

    def is_greater_than_threshold(self, threshold):
        return self.value > threshold

    def __eq__(self, threshold):
        return self.value == threshold

    def __gt__(self, threshold):
        return self.value > threshold

    def __lt__(self, threshold):
        return self.value < threshold

    def __repr__(self):
        return f"PredictionScore: <value: {self.value}>"


class ObjectPrediction(ObjectAnnotation):
    def __init__(self, bbox: Optional[List[int]] = None, category_id: Optional[int] = None, category_name: Optional[str] = None,
                 segmentation=None, score: float = 0.0, shift_amount: Optional[List[int]] = [0, 0], full_shape: Optional[List[int]] = None):
        self.score = PredictionScore(score)
        super().__init__(bbox=bbox, category_id=category_id, segmentation=segmentation, category_name=category_name,
                         shift_amount=shift_amount, full_shape=full_shape)

    def get_shifted_object_prediction(self):
        """Box moved by its shift_amount into full-frame coordinates; shift reset, full_shape dropped (docs :94-120)."""
        p = ObjectPrediction(bbox=self.bbox.get_shifted_box().to_xyxy(), category_id=self.category.id, score=self.score.value,
                             segmentation=None, category_name=self.category.name, shift_amount=[0, 0], full_shape=None)
        if hasattr(self, "keypoints"):
            p.keypoints = self.keypoints
        return p

    def to_coco_prediction(self, image_id=None):
        raise NotImplementedError("COCO export is outside the hot path")

    def __repr__(self):
        return f"ObjectPrediction<\n    bbox: {self.bbox},\n    mask: {self.mask},\n    score: {self.score},\n    category: {self.category}>"


class PredictionResult:
    def __init__(self, object_prediction_list: List[ObjectPrediction], image, durations_in_seconds: Dict[str, Any] = dict()):
        # the PIL copy of the picture (docs sahi/prediction.py:160-165) is made when somebody asks for it: the detection loop itself never does
        self._image_src, self._image = image, None
        self._size = (int(image.shape[1]), int(image.shape[0])) if isinstance(image, np.ndarray) and image.ndim == 3 and image.shape[2] <= 4 else None
        self.object_prediction_list = object_prediction_list
        self.durations_in_seconds = durations_in_seconds

    @property
    def image(self):
        if self._image is None:
            self._image = read_image_as_pil(self._image_src)
        return self._image

    @image.setter
    def image(self, value):
        self._image, self._size = value, None

    @property
    def image_width(self):
        return self._size[0] if self._size else self.image.size[0]

    @property
    def image_height(self):
        return self._size[1] if self._size else self.image.size[1]

    def to_coco_annotations(self):
        return [{"bbox": p.bbox.to_xywh(), "score": p.score.value, "category_id": p.category.id, "category_name": p.category.name}
                for p in self.object_prediction_list]
