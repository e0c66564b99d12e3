"""get_prediction / get_sliced_prediction with the reference's signatures (docs sahi/predict.py:63-139, 142-345).

Two execution paths with identical results (tests/test_gpu_compat.py):
  * generic — any DetectionModel plugin: the reference's sequential loop (one perform_inference per slice);
  * batched — models exposing `perform_inference_batch` (this build's YOLOv11PoseDetectionModel): all slices and the
    full-frame pass go to the GPU as ONE ragged batch (ffp_det_infer_tiles); conversion and merge are unchanged.
"""
from __future__ import annotations

import time
from typing import List, Optional

import numpy as np

from sahi.postprocess.combine import GreedyNMMPostprocess, LSNMSPostprocess, NMMPostprocess, NMSPostprocess, PostprocessPredictions
from sahi.prediction import ObjectPrediction, PredictionResult
from sahi.slicing import slice_image
from sahi.utils.cv import read_image_as_array

POSTPROCESS_NAME_TO_CLASS = {"GREEDYNMM": GreedyNMMPostprocess, "NMM": NMMPostprocess, "NMS": NMSPostprocess, "LSNMS": LSNMSPostprocess}


def filter_predictions(object_prediction_list, exclude_classes_by_name, exclude_classes_by_id):
    return [p for p in object_prediction_list
            if p.category.name not in (exclude_classes_by_name or []) and p.category.id not in (exclude_classes_by_id or [])]


def get_prediction(image, detection_model, shift_amount: list = [0, 0], full_shape=None, postprocess: Optional[PostprocessPredictions] = None,
                   verbose: int = 0, exclude_classes_by_name: Optional[List[str]] = None, exclude_classes_by_id: Optional[List[int]] = None) -> PredictionResult:
    durations = dict()
    arr = read_image_as_array(image)
    t0 = time.time()
    detection_model.perform_inference(np.ascontiguousarray(arr))
    durations["prediction"] = time.time() - t0
    if full_shape is None:
        full_shape = [arr.shape[0], arr.shape[1]]
    t0 = time.time()
    detection_model.convert_original_predictions(shift_amount=shift_amount, full_shape=full_shape)
    preds = filter_predictions(detection_model.object_prediction_list, exclude_classes_by_name, exclude_classes_by_id)
    if postprocess is not None:
        preds = postprocess(preds)
    durations["postprocess"] = time.time() - t0
    if verbose == 1:
        print("Prediction performed in", durations["prediction"], "seconds.")
    return PredictionResult(image=image, object_prediction_list=preds, durations_in_seconds=durations)


def get_sliced_prediction(image, detection_model=None, slice_height: Optional[int] = None, slice_width: Optional[int] = None,
                          overlap_height_ratio: float = 0.2, overlap_width_ratio: float = 0.2, perform_standard_pred: bool = True,
                          postprocess_type: str = "GREEDYNMM", postprocess_match_metric: str = "IOS", postprocess_match_threshold: float = 0.5,
                          postprocess_class_agnostic: bool = False, verbose: int = 1, merge_buffer_length: Optional[int] = None,
                          auto_slice_resolution: bool = True, slice_export_prefix: Optional[str] = None, slice_dir: Optional[str] = None,
                          exclude_classes_by_name: Optional[List[str]] = None, exclude_classes_by_id: Optional[List[int]] = None) -> PredictionResult:
    durations = dict()
    t0 = time.time()
    sl = slice_image(image=image, output_file_name=slice_export_prefix, output_dir=slice_dir, slice_height=slice_height, slice_width=slice_width,
                     overlap_height_ratio=overlap_height_ratio, overlap_width_ratio=overlap_width_ratio, auto_slice_resolution=auto_slice_resolution)
    num_slices = len(sl)
    durations["slice"] = time.time() - t0
    if postprocess_type not in POSTPROCESS_NAME_TO_CLASS:
        raise ValueError(f"postprocess_type should be one of {list(POSTPROCESS_NAME_TO_CLASS.keys())} but given as {postprocess_type}")
    postprocess = POSTPROCESS_NAME_TO_CLASS[postprocess_type](match_threshold=postprocess_match_threshold, match_metric=postprocess_match_metric,
                                                              class_agnostic=postprocess_class_agnostic)
    postprocess_time = 0.0
    t0 = time.time()
    if verbose in (1, 2):
        print(f"Performing prediction on {num_slices} slices.")
    full_shape = [sl.original_image_height, sl.original_image_width]
    preds: List[ObjectPrediction] = []
    do_standard = num_slices > 1 and perform_standard_pred
    if hasattr(detection_model, "perform_inference_batch") and merge_buffer_length is None:
        tiles = [[x, y, x + im.shape[1], y + im.shape[0]] for (x, y), im in zip(sl.starting_pixels, sl.images)]
        if do_standard:
            tiles.append([0, 0, full_shape[1], full_shape[0]])
        per_tile = detection_model.perform_inference_batch(sl.full_image, tiles)
        for k, res in enumerate(per_tile):
            detection_model._original_predictions = res
            standard = do_standard and k == len(tiles) - 1
            detection_model.convert_original_predictions(shift_amount=[0, 0] if standard else sl.starting_pixels[k], full_shape=full_shape)
            lst = filter_predictions(detection_model.object_prediction_list, exclude_classes_by_name, exclude_classes_by_id)
            preds.extend(lst if standard else [p.get_shifted_object_prediction() for p in lst if p])
    else:
        for k in range(num_slices):
            r = get_prediction(image=sl.images[k], detection_model=detection_model, shift_amount=sl.starting_pixels[k], full_shape=full_shape,
                               exclude_classes_by_name=exclude_classes_by_name, exclude_classes_by_id=exclude_classes_by_id)
            preds.extend(p.get_shifted_object_prediction() for p in r.object_prediction_list if p)
            if merge_buffer_length is not None and len(preds) > merge_buffer_length:
                t1 = time.time()
                preds = postprocess(preds)
                postprocess_time += time.time() - t1
        if do_standard:
            r = get_prediction(image=image, detection_model=detection_model, shift_amount=[0, 0], full_shape=full_shape, postprocess=None,
                               exclude_classes_by_name=exclude_classes_by_name, exclude_classes_by_id=exclude_classes_by_id)
            preds.extend(r.object_prediction_list)
    if len(preds) > 1:
        t1 = time.time()
        preds = postprocess(preds)
        postprocess_time += time.time() - t1
    total = time.time() - t0
    durations["prediction"] = total - postprocess_time
    durations["postprocess"] = postprocess_time
    if verbose == 2:
        print("Slicing performed in", durations["slice"], "seconds.")
        print("Prediction performed in", durations["prediction"], "seconds.")
        print("Postprocessing performed in", durations["postprocess"], "seconds.")
    return PredictionResult(image=image, object_prediction_list=preds, durations_in_seconds=durations)
