"""YOLOv11PoseDetectionModel — the SAHI plugin of the reference (utils/yolo_wrapper.py:7-229) over libffp.so.

Same constructor arguments, hooks, properties and side channel (keypoints_cache keyed by the shifted int box,
attach_keypoints_to_predictions with the exact-key / best-IoU>0.5 lookup). `self.model` is a `YOLO` object with the
two call forms the reference uses: `.predict(source=, conf=, device=, imgsz=, verbose=)` (:74-80) and
`model(img, conf=, verbose=False)` (eval/eval_dual.py:153,245), returning Results-like objects whose
`.boxes.xyxy/.conf/.cls` and `.keypoints.data` are torch CPU tensors (the reference calls `.cpu().numpy()` on them).
Additions: `perform_inference_batch` (all slices of a frame as one ragged GPU batch) and `get_keypoints_for_bbox`
(called by pipeline_v4_yolo/app_yolo_sahi.py:80-84 but missing in the reference).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

import ffp_amd  # noqa: F401
from ffp_amd import _lib, synth, weights_io
from sahi.models.base import DetectionModel
from sahi.prediction import ObjectPrediction


class _Boxes:
    def __init__(self, rows):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(rows[:, :6]))
        self.xyxy, self.conf, self.cls = t[:, :4], t[:, 4], t[:, 5]
        self.data = t

    def __len__(self):
        return int(self.data.shape[0])


class _Keypoints:
    def __init__(self, rows, nkpt):
        import torch
        self.data = torch.from_numpy(np.ascontiguousarray(rows[:, 6:].reshape(-1, nkpt, 3)))
        self.xy, self.conf = self.data[..., :2], self.data[..., 2]


class Results:
    def __init__(self, rows: np.ndarray, nkpt: int, orig_shape, names):
        self.boxes = _Boxes(rows)
        self.keypoints = _Keypoints(rows, nkpt) if nkpt > 0 else None
        self.orig_shape, self.names = tuple(orig_shape), names

    def __len__(self):
        return len(self.boxes)


def _device_index(device) -> int:
    if isinstance(device, int):
        return device
    s = str(device)
    if s in ("cpu", "mps"):
        raise RuntimeError(f"device='{s}': this build runs on MI355X only (libffp.so has no CPU path)")
    return int(s.split(":")[1]) if ":" in s else 0


class YOLO:
    """Loader + predictor standing in for `ultralytics.YOLO(model_path)` (utils/yolo_wrapper.py:55)."""

    def __init__(self, model_path: str, device="cuda:0", precision: str = "f32x3"):
        """precision: "f32x3" (fp32-grade split-fp16 MFMA, default: same parity bar as exact fp32, 1.5x faster), "f32" (exact fp32 MFMA), "f16"."""
        self.names = {0: "face"}
        self.model_path = model_path
        if isinstance(model_path, dict):
            W = model_path
        elif str(model_path).startswith("synthetic:"):       # e.g. synthetic:yolo11s-pose (random-init, for benchmarks/tests)
            W = synth.yolo11_pose_weights(str(model_path).split("yolo11")[1][0])
        elif str(model_path).endswith(".ffpw"):
            W = weights_io.load(model_path)
        else:
            raise ValueError(f"{model_path}: Ultralytics .pt checkpoints are pickled module graphs; convert them once with "
                             "weights_io.from_ultralytics_state_dict where `ultralytics` is installed and pass the .ffpw file")
        c0 = W["model.0.conv.weight"].shape[0]
        self.arch = {16: "n", 32: "s"}[c0]
        self.nc = int(W["model.23.cv3.0.2.weight"].shape[0])
        self.nkpt = int(W["model.23.cv4.0.2.weight"].shape[0]) // 3
        self._det = _lib.Detector(W, arch=self.arch, nc=self.nc, nkpt=self.nkpt, device=_device_index(device),
                                  precision={"f16": _lib.PREC_F16, "f32": _lib.PREC_F32, "f32x3": _lib.PREC_F32X3}[precision])

    def predict_tiles(self, frame: np.ndarray, tiles, conf: float, imgsz: int, iou: float = 0.7, max_det: int = 300) -> List[List[Results]]:
        rows = self._det.infer_tiles(frame, tiles, imgsz, conf, iou, max_det, chan_order=_lib.CHAN_AS_BGR)
        return [[Results(r, self.nkpt, (t[3] - t[1], t[2] - t[0]), self.names)] for r, t in zip(rows, tiles)]

    def predict(self, source=None, conf: float = 0.25, device=None, imgsz: int = 640, verbose: bool = False, iou: float = 0.7,
                max_det: int = 300, **_ignored) -> List[Results]:
        img = source
        if not isinstance(img, np.ndarray):
            img = np.asarray(img.convert("RGB"))[..., ::-1]      # PIL -> BGR like Ultralytics' loader
        h, w = img.shape[:2]
        return self.predict_tiles(np.ascontiguousarray(img), [[0, 0, w, h]], conf, imgsz, iou, max_det)[0]

    def __call__(self, source=None, **kw):
        return self.predict(source, **kw)


class YOLOv11PoseDetectionModel(DetectionModel):
    def __init__(self, model_path: str = None, confidence_threshold: float = 0.3, device: str = "cuda:0", image_size: int = 1024, **kwargs):
        self._model_path, self._device, self._image_size, self._confidence_threshold = model_path, device, image_size, confidence_threshold
        self.keypoints_cache = {}
        super().__init__(model_path=model_path, confidence_threshold=confidence_threshold, device=device, **kwargs)
        self.model_path, self.device, self.image_size = self._model_path, self._device, self._image_size
        self.confidence_threshold = self._confidence_threshold

    def set_device(self, device=None):
        self.device = device

    def load_model(self):
        if not self.model_path:
            raise ValueError("model_path harus ditentukan")
        print(f"Loading model from: {self.model_path}")
        print(f"Device: {self._device}, Image size: {self._image_size}")
        self.model = YOLO(self.model_path, device=self._device)
        self.category_mapping = {"0": "face"}

    def unload_model(self):
        self.model = None
        self.keypoints_cache = {}

    def perform_inference(self, image: np.ndarray):
        if image.dtype != np.uint8:
            image = (image * 255).astype(np.uint8)
        self._original_predictions = self.model.predict(source=image, conf=self.confidence_threshold, device=self.device,
                                                        imgsz=self.image_size, verbose=False)

    def perform_inference_batch(self, frame: np.ndarray, tiles) -> List[List[Results]]:
        """One GPU batch for all tiles (slices + full frame) of a frame; each entry is what perform_inference would have stored."""
        if frame.dtype != np.uint8:
            frame = (frame * 255).astype(np.uint8)
        return self.model.predict_tiles(np.ascontiguousarray(frame), tiles, self.confidence_threshold, self.image_size)

    def _create_object_prediction_list_from_original_predictions(self, shift_amount_list: Optional[List[List[int]]] = [[0, 0]],
                                                                 full_shape_list: Optional[List[List[int]]] = None):
        original = self._original_predictions
        if not original or len(original[0].boxes) == 0:
            self._object_prediction_list_per_image = [[]]
            return
        # SAHI passes flat [x, y] / [h, w]; nested lists are accepted too (utils/yolo_wrapper.py:99-118)
        if shift_amount_list is None:
            shift = [0, 0]
        elif isinstance(shift_amount_list, list):
            if len(shift_amount_list) > 0 and isinstance(shift_amount_list[0], list):
                shift = shift_amount_list[0]
            else:
                shift = shift_amount_list if len(shift_amount_list) == 2 else [0, 0]
        else:
            shift = [0, 0]
        if full_shape_list is None:
            full_shape = None
        elif isinstance(full_shape_list, list):
            full_shape = full_shape_list[0] if (len(full_shape_list) > 0 and isinstance(full_shape_list[0], list)) else full_shape_list
        else:
            full_shape = None
        result = original[0]
        boxes = result.boxes
        kdata = result.keypoints.data.cpu().numpy() if getattr(result, "keypoints", None) is not None else None
        out = []
        for i in range(len(boxes)):
            score = float(boxes.conf[i])                      # no second confidence filter (:134-135)
            x1, y1, x2, y2 = boxes.xyxy[i].cpu().numpy().astype(int)
            pred = ObjectPrediction(bbox=[int(x1), int(y1), int(x2), int(y2)], category_id=0, category_name="face", score=score,
                                    shift_amount=shift, full_shape=full_shape)
            if kdata is not None and i < len(kdata):
                k = kdata[i].copy()
                k[:, 0] += shift[0]
                k[:, 1] += shift[1]
                self.keypoints_cache[f"{int(x1) + shift[0]}_{int(y1) + shift[1]}_{int(x2) + shift[0]}_{int(y2) + shift[1]}"] = k
            out.append(pred)
        self._object_prediction_list_per_image = [out]

    def attach_keypoints_to_predictions(self, object_prediction_list):
        for pred in object_prediction_list:
            k = self.get_keypoints_for_bbox(pred.bbox.to_voc_bbox())
            if k is not None:
                pred.keypoints = k
        return object_prediction_list

    def get_keypoints_for_bbox(self, bbox):
        key = f"{bbox[0]}_{bbox[1]}_{bbox[2]}_{bbox[3]}"
        if key in self.keypoints_cache:
            return self.keypoints_cache[key]
        best_iou, best = 0.0, None
        for k, kp in self.keypoints_cache.items():
            iou = self._calculate_iou(bbox, [int(float(v)) for v in k.split("_")])
            if iou > best_iou:
                best_iou, best = iou, kp
        return best if best_iou > 0.5 else None

    def _calculate_iou(self, box1, box2):
        xa, ya, xb, yb = max(box1[0], box2[0]), max(box1[1], box2[1]), min(box1[2], box2[2]), min(box1[3], box2[3])
        if xb < xa or yb < ya:
            return 0.0
        inter = (xb - xa) * (yb - ya)
        union = (box1[2] - box1[0]) * (box1[3] - box1[1]) + (box2[2] - box2[0]) * (box2[3] - box2[1]) - inter
        return inter / union if union > 0 else 0.0

    @property
    def num_categories(self):
        return len(self.category_names)

    @property
    def has_mask(self):
        return False

    @property
    def category_names(self):
        return ["face"]
