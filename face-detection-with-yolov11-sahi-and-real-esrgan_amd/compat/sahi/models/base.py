"""DetectionModel plugin contract (docs sahi/base.py:12-197): constructor arguments, hooks and properties kept."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np

from sahi.annotation import Category
from sahi.prediction import ObjectPrediction


class DetectionModel:
    required_packages: List[str] = []

    def __init__(self, model_path: Optional[str] = None, model: Optional[Any] = None, config_path: Optional[str] = None,
                 device: Optional[str] = None, mask_threshold: float = 0.5, confidence_threshold: float = 0.3,
                 category_mapping: Optional[Dict] = None, category_remapping: Optional[Dict] = None, load_at_init: bool = True,
                 image_size: Optional[int] = None):
        self.model_path, self.config_path, self.model = model_path, config_path, None
        self.mask_threshold, self.confidence_threshold = mask_threshold, confidence_threshold
        self.category_mapping, self.category_remapping, self.image_size = category_mapping, category_remapping, image_size
        self._original_predictions = None
        self._object_prediction_list_per_image = None
        self.set_device(device)
        self.check_dependencies()
        if load_at_init:
            if model:
                self.set_model(model)
            else:
                self.load_model()

    def check_dependencies(self, packages: Optional[List[str]] = None) -> None:
        return None

    def load_model(self):
        raise NotImplementedError()

    def set_model(self, model: Any, **kwargs):
        raise NotImplementedError()

    def set_device(self, device: Optional[str] = None):
        self.device = device if device is not None else "cuda:0"

    def unload_model(self):
        self.model = None

    def perform_inference(self, image: np.ndarray):
        raise NotImplementedError()

    def _create_object_prediction_list_from_original_predictions(self, shift_amount_list=[[0, 0]], full_shape_list=None):
        raise NotImplementedError()

    def _apply_category_remapping(self):
        if self.category_remapping is None:
            raise ValueError("self.category_remapping cannot be None")
        for lst in self._object_prediction_list_per_image or []:
            for p in lst:
                p.category = Category(id=self.category_remapping[str(p.category.id)], name=p.category.name)

    def convert_original_predictions(self, shift_amount: Optional[List[List[int]]] = [[0, 0]], full_shape: Optional[List[List[int]]] = None):
        self._create_object_prediction_list_from_original_predictions(shift_amount_list=shift_amount, full_shape_list=full_shape)
        if self.category_remapping:
            self._apply_category_remapping()

    @property
    def object_prediction_list(self) -> List[ObjectPrediction]:
        if not self._object_prediction_list_per_image:
            return []
        return self._object_prediction_list_per_image[0]

    @property
    def object_prediction_list_per_image(self):
        return self._object_prediction_list_per_image or []

    @property
    def original_predictions(self):
        return self._original_predictions
