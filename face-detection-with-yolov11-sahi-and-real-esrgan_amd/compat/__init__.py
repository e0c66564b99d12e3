"""Import shims that keep the reference's module names.

Put this directory on sys.path (``ffp_amd.compat.install()``) and the reference's scripts resolve
``from sahi.predict import get_sliced_prediction``, ``from sahi.prediction import ObjectPrediction``,
``from sahi.models.base import DetectionModel``, ``from utils.yolo_wrapper import YOLOv11PoseDetectionModel`` and
``from utils.enhancer import FaceEnhancer`` (imports at /root/reference/pipeline_v4_yolo/app_yolo_sahi.py:9-17) to
this build: same names, arguments and error behaviour, with libffp.so underneath.
"""
import os
import sys


def install() -> str:
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    return here
