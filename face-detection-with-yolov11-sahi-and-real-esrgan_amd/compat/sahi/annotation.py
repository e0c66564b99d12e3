"""Value types behind ObjectPrediction (mirror of sahi.annotation as used by docs sahi/prediction.py:44-120)."""
from __future__ import annotations

from typing import List, Optional


class Category:
    def __init__(self, id: Optional[int] = None, name: Optional[str] = None):
        if id is not None and not isinstance(id, int):
            raise TypeError("id should be integer")
        if name is not None and not isinstance(name, str):
            raise TypeError("name should be string")
        self.id, self.name = id, name

    def __repr__(self):
        return f"Category: <id: {self.id}, name: {self.name}>"


class BoundingBox:
    """xyxy box with the shift that maps it from slice to full-frame coordinates."""

    def __init__(self, box: List[float], shift_amount: List[int] = [0, 0]):
        if box[0] < 0 or box[1] < 0 or box[2] < 0 or box[3] < 0:
            raise Exception("Box coords [minx, miny, maxx, maxy] cannot be negative")
        self.minx, self.miny, self.maxx, self.maxy = box[0], box[1], box[2], box[3]
        self.shift_x, self.shift_y = shift_amount[0], shift_amount[1]

    @property
    def shift_amount(self):
        return [self.shift_x, self.shift_y]

    @property
    def area(self):
        return (self.maxx - self.minx) * (self.maxy - self.miny)

    def to_xyxy(self):
        return [self.minx, self.miny, self.maxx, self.maxy]

    to_voc_bbox = to_xyxy

    def to_xywh(self):
        return [self.minx, self.miny, self.maxx - self.minx, self.maxy - self.miny]

    to_coco_bbox = to_xywh

    def get_shifted_box(self):
        return BoundingBox([self.minx + self.shift_x, self.miny + self.shift_y, self.maxx + self.shift_x, self.maxy + self.shift_y],
                           shift_amount=[0, 0])

    def __repr__(self):
        return f"BoundingBox: <{(self.minx, self.miny, self.maxx, self.maxy)}, w: {self.maxx - self.minx}, h: {self.maxy - self.miny}>"


class ObjectAnnotation:
    """bbox (clipped to >= 0 and, when full_shape is given, to <= full_shape) + category; masks are not produced on this path."""

    def __init__(self, bbox=None, segmentation=None, category_id=None, category_name=None, shift_amount=[0, 0], full_shape=None):
        if not isinstance(category_id, int):
            raise ValueError("category_id must be an integer")
        if bbox is None and segmentation is None:
            raise ValueError("you must provide a bbox or segmentation")
        if segmentation is not None:
            raise NotImplementedError("segmentation masks are outside the face-detection hot path")
        if type(bbox).__module__ == "numpy":
            bbox = bbox.tolist()
        xmin, ymin = max(bbox[0], 0), max(bbox[1], 0)
        xmax, ymax = (min(bbox[2], full_shape[1]), min(bbox[3], full_shape[0])) if full_shape else (bbox[2], bbox[3])
        self.mask = None
        self.bbox = BoundingBox([xmin, ymin, xmax, ymax], shift_amount=shift_amount)
        self.category = Category(id=category_id, name=category_name if category_name else str(category_id))
        self.merged = None
        self.full_shape = full_shape

    def __repr__(self):
        return f"ObjectAnnotation<\n    bbox: {self.bbox},\n    mask: {self.mask},\n    category: {self.category}>"
