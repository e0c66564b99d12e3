"""eval.eval_official_widerface with the reference's surface (/root/reference/eval/eval_official_widerface.py:44-494): the official
WIDER FACE protocol (easy / medium / hard AP from the .mat ground truth). The per-image matching and the per-threshold PR counts —
`_image_eval` :302-347 + `_img_pr_info` :349-375, the O(images x predictions x faces) part, including the `bbox_overlaps` the reference
imports from a Cython extension (:24-33) — run in ONE launch on the GPU (ffp_eval_wider_pr); `_dataset_pr_info` and `_voc_ap` are the
reference's few numpy lines on the resulting integer counts. Inference (`_run_single_inference` :166-255, all four pipelines: baseline,
SAHI, enhance -> detect, enhance -> SAHI, with full or bounded enhancement) goes through this build's `utils.yolo_wrapper` /
`utils.enhancer` / `sahi.predict` shims; predictions can also be handed in (`run(all_predictions=...)`), which is how the tests
drive it (no WIDER FACE data or trained weights exist on the build machine). Plots (matplotlib) are left to the caller.
"""
from collections import defaultdict
from pathlib import Path

import numpy as np

from ffp_amd import _lib


class OfficialWiderFaceEvaluator:
    def __init__(self, gt_path="data/dataset/widerface/wider_face_split", images_path="data/dataset/widerface/WIDER_val/images",
                 model_path="models/yolo11s-pose-default/yolo11s_pose/weights/best.pt", device="cuda:0", use_sahi=True, slicing_strategy="uniform",
                 use_enhancer=False, bounded_enhancement=False, face_size_threshold=50, load_model=True):
        self.gt_path, self.images_path = Path(gt_path), Path(images_path)
        self.settings = ["easy", "medium", "hard"]
        self.iou_threshold = 0.5
        self.thresh_num = 1000
        self.use_sahi, self.slicing_strategy = use_sahi, slicing_strategy
        self.use_enhancer, self.bounded_enhancement, self.face_size_threshold = use_enhancer, bounded_enhancement, face_size_threshold
        self.inference_confidence = 0.01
        self.sahi_config = {"slice_height": 640, "slice_width": 640, "overlap_ratio": 0.2} if slicing_strategy == "uniform" else {"overlap_ratio": 0.2}
        self.enhancement_stats = defaultdict(lambda: {"enhanced": 0, "skipped": 0, "total": 0})
        self.detection_model = None
        self.face_enhancer = None
        if load_model:
            from utils.yolo_wrapper import YOLOv11PoseDetectionModel
            self.detection_model = YOLOv11PoseDetectionModel(model_path=model_path, confidence_threshold=self.inference_confidence, device=device, load_at_init=True)
            if self.use_enhancer:                                      # :90-98: a failing enhancer switches the mode off, it does not abort
                try:
                    from utils.enhancer import FaceEnhancer
                    self.face_enhancer = FaceEnhancer(model_name="RealESRGAN_x2plus")
                    print(f"   ✓ Model Enhancer dimuat! (Skala: {self.face_enhancer.scale}x)")
                except Exception as e:
                    print(f"   ❌ Gagal memuat model Enhancer: {e}. Enhancement dinonaktifkan.")
                    self.use_enhancer = False
        self._build_mode_string()
        self._load_official_ground_truth()

    def _build_mode_string(self):
        """:103-116 (the official evaluator joins with ' -> ')"""
        mode_parts = []
        if self.use_enhancer:
            mode_parts.append(f"BOUNDED-ENHANCE (<{self.face_size_threshold}px)" if self.bounded_enhancement else "FULL-ENHANCE")
        mode_parts.append(f"SAHI ({self.slicing_strategy})" if self.use_sahi else "BASELINE")
        self.mode_string = " -> ".join(mode_parts)

    def _quick_face_analysis(self, img):
        """Bounded enhancement's decision (:147-158): a plain predict at conf 0.05; enhance when nothing is found, when more than half of
        the faces are smaller than face_size_threshold (longer side) or when their mean size is."""
        if img is None:
            return False, "Image load failed", {}
        results = self.detection_model.model(img, conf=0.05, verbose=False)
        if len(results) == 0 or results[0].boxes is None or len(results[0].boxes) == 0:
            return True, "No faces detected", {}
        face_sizes = [max(box[2] - box[0], box[3] - box[1]) for box in results[0].boxes.xyxy.cpu().numpy()]
        small_face_ratio = sum(1 for sz in face_sizes if sz < self.face_size_threshold) / len(face_sizes)
        if small_face_ratio > 0.5 or np.mean(face_sizes) < self.face_size_threshold:
            return True, f"Small faces detected (ratio: {small_face_ratio:.2f})", {}
        return False, "Faces are large enough", {}

    def _load_official_ground_truth(self):
        from scipy.io import loadmat
        gt_mat = loadmat(self.gt_path / "wider_face_val.mat")
        self.facebox_list, self.event_list, self.file_list = gt_mat["face_bbx_list"], gt_mat["event_list"], gt_mat["file_list"]
        self.setting_gts = {s: loadmat(self.gt_path / f"wider_{s}_val.mat")["gt_list"] for s in self.settings}

    def _get_slice_size_adaptive(self, w, h):
        max_dim = max(w, h)
        if max_dim > 2500:
            return 512
        if max_dim > 1500:
            return 416
        return 320

    # ---- inference (:166-255), SAHI or plain predict; boxes as x, y, w, h, score -------------------------------------------------
    def _run_single_inference(self, img_path):
        import cv2                                                     # a real OpenCV if present, else this build's shim: BGR either way (:168)
        img = cv2.imread(str(img_path))
        if img is None:
            return np.array([])
        # 1. enhancement phase (:174-186): the whole picture through Real-ESRGAN x2plus, always or when the quick analysis says so
        inference_img, was_enhanced = img, False
        if getattr(self, "use_enhancer", False) and getattr(self, "face_enhancer", None):
            enhance_decision = self._quick_face_analysis(img)[0] if self.bounded_enhancement else True
            if enhance_decision:
                enhanced_image, success = self.face_enhancer.enhance_image(img)
                if success:
                    inference_img, was_enhanced = enhanced_image, True
        img = inference_img
        pred = self._detect(img)
        # 3. boxes of an upscaled picture go back to the original's coordinates (:246-251)
        if was_enhanced and self.face_enhancer.scale > 1 and len(pred) > 0:
            pred[:, :4] /= self.face_enhancer.scale
        return pred.astype("float")

    def _detect(self, img):
        """2. detection phase (:188-243) on the picture to infer on"""
        if self.use_sahi:
            from sahi.predict import get_sliced_prediction
            cfg = dict(self.sahi_config)
            if "slice_height" in cfg:
                slice_h, slice_w = cfg["slice_height"], cfg["slice_width"]
            else:                                                      # 'adaptive' (:196-198)
                slice_h = slice_w = self._get_slice_size_adaptive(img.shape[1], img.shape[0])
            res = get_sliced_prediction(img, self.detection_model, slice_height=slice_h, slice_width=slice_w, overlap_height_ratio=cfg["overlap_ratio"],
                                        overlap_width_ratio=cfg["overlap_ratio"], postprocess_type="NMS", postprocess_match_threshold=0.5,
                                        postprocess_class_agnostic=True, verbose=0)
            rows = [[*p.bbox.to_xywh(), p.score.value] for p in res.object_prediction_list]
            return np.asarray(rows, np.float64).reshape(-1, 5) if rows else np.array([])
        # baseline (:219-243): the YOLO object itself at its default imgsz (640), float xyxy -> top-left x, y, w, h — not the wrapper's
        # image_size, not the int-truncated boxes of convert_original_predictions
        results = self.detection_model.model(img, conf=self.inference_confidence, verbose=False)
        boxes = results[0].boxes
        if len(boxes) == 0:
            return np.array([])
        xyxy = boxes.xyxy.cpu().numpy()
        x1, y1 = xyxy[:, 0], xyxy[:, 1]
        xywh = np.column_stack((x1, y1, xyxy[:, 2] - x1, xyxy[:, 3] - y1))
        return np.column_stack((xywh, boxes.conf.cpu().numpy())).astype("float")

    def _run_inference_on_all_images(self):
        predictions = defaultdict(dict)
        for i, event in enumerate(self.event_list):
            event_name = event[0][0]
            for img_file in self.file_list[i][0]:
                img_name = img_file[0][0]
                p = self.images_path / event_name / f"{img_name}.jpg"
                if p.exists():
                    predictions[event_name][img_name] = self._run_single_inference(str(p))
        return predictions

    # ---- evaluation --------------------------------------------------------------------------------------------------------------
    def _voc_ap(self, rec, prec):
        """Area under the monotone precision envelope (:282-300), vectorised: running maximum from the right, summed over recall steps."""
        r = np.concatenate(([0.0], np.asarray(rec, np.float64), [1.0]))
        p = np.concatenate(([0.0], np.asarray(prec, np.float64), [0.0]))
        env = np.maximum.accumulate(p[::-1])[::-1]
        step = np.flatnonzero(r[1:] != r[:-1])
        return np.sum((r[step + 1] - r[step]) * env[step + 1])

    def _dataset_pr_info(self, pr_curve, count_face):
        out = np.zeros((self.thresh_num, 2))
        nz = pr_curve[:, 0] != 0
        out[nz, 0] = pr_curve[nz, 1] / pr_curve[nz, 0]
        out[:, 1] = pr_curve[:, 1] / count_face
        return out

    def _flatten(self, setting, all_predictions):
        """The reference's nested loops (:404-431) as three per-image lists + the face count of the setting."""
        gt_list = self.setting_gts[setting]
        preds, gts, evaluate, count_face = [], [], [], 0
        for i in range(len(self.event_list)):
            event_name = self.event_list[i][0][0]
            img_list = self.file_list[i][0]
            pred_list_event = all_predictions.get(event_name, {})
            for j in range(len(img_list)):
                pred_info = np.asarray(pred_list_event.get(img_list[j][0][0], np.array([])), np.float64)
                gt_boxes = self.facebox_list[i][0][j][0].astype("float")
                keep_index = np.asarray(gt_list[i][0][j][0]).reshape(-1).astype(np.int64)
                count_face += len(keep_index)
                if len(gt_boxes) == 0 or len(pred_info) == 0:
                    continue
                flag = np.zeros(gt_boxes.shape[0], np.uint8)
                if len(keep_index):
                    flag[keep_index - 1] = 1
                preds.append(pred_info.reshape(-1, 5)); gts.append(gt_boxes.reshape(-1, 4)); evaluate.append(flag)
        return preds, gts, evaluate, count_face

    def _evaluate_setting(self, setting, all_predictions):
        preds, gts, evaluate, count_face = self._flatten(setting, all_predictions)
        counts = _lib.eval_wider_pr(preds, gts, evaluate, self.iou_threshold, self.thresh_num).astype(np.float64)
        pr_curve = self._dataset_pr_info(counts, count_face)
        propose, recall = pr_curve[:, 0], pr_curve[:, 1]
        return self._voc_ap(recall, propose), recall, propose

    def run(self, all_predictions=None):
        if all_predictions is None:
            all_predictions = self._run_inference_on_all_images()
        results = {}
        for setting in self.settings:
            ap, _, _ = self._evaluate_setting(setting, all_predictions)
            results[setting] = ap
        print("\n" + "=" * 50 + f"\n📊 HASIL EVALUASI - {self.mode_string}\n" + "=" * 50)
        for setting, ap in results.items():
            print(f"  - {setting.capitalize():<7} AP: {ap:.4f}")
        print("=" * 50)
        return results
