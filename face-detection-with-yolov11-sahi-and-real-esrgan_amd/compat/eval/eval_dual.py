"""eval.eval_dual's evaluation core with the reference's names (/root/reference/eval/eval_dual.py:272-433): sub-category and difficulty
metrics from `subcategory_gt.json`. The matching of every image (IoU of each prediction against the valid and the ignored faces,
first-best assignment, :369-399) is one GPU launch (ffp_eval_dual_match); the 11-point AP and precision / recall / F1 are the reference's
numpy expressions on the resulting flags. Predictions are handed in per image path ({path: [{'bbox': [x, y, w, h], 'confidence': c}]}) or produced by `run_inference`
(:185-270: the four pipelines — baseline, SAHI, enhance -> detect, enhance -> SAHI — with full or bounded enhancement, cached per path)
through this build's `utils.yolo_wrapper` / `utils.enhancer` / `sahi.predict` shims when `load_model=True`.
"""
import json

import numpy as np

from ffp_amd import _lib


class DualWiderFaceEvaluator:
    subcategories = ["large_clear", "large_degraded", "medium_clear", "medium_degraded", "small_clear", "small_degraded"]

    def __init__(self, subcategory_file=None, subcategory_gt=None, predictions=None, iou_threshold=0.5, global_confidence=0.25,
                 base_path="data/dataset/widerface", model_path="models/yolo11s-pose-default/yolo11s_pose/weights/best.pt", device="cuda:0",
                 use_sahi=False, use_enhancer=False, bounded_enhancement=False, face_size_threshold=50, slicing_strategy="uniform",
                 sahi_match_thresholds=(0.5,), sahi_match_metric="IOS", load_model=False):
        if subcategory_gt is None:
            with open(subcategory_file, "r") as f:
                subcategory_gt = json.load(f)
        self.subcategory_gt = subcategory_gt
        self.predictions = predictions or {}
        self.iou_threshold, self.global_confidence = iou_threshold, global_confidence
        # pipeline configuration (:66-112)
        self.base_path = base_path
        self.inference_confidence = 0.01 if use_sahi else 0.5
        self.use_sahi, self.use_enhancer, self.bounded_enhancement = use_sahi, use_enhancer, bounded_enhancement
        self.face_size_threshold, self.slicing_strategy = face_size_threshold, slicing_strategy
        self.enhancement_stats = {"total_images": 0, "enhanced_images": 0, "skipped_images": 0}
        self.prediction_cache = {}
        self.sahi_config = {"confidence_threshold": 0.01, "postprocess_match_thresholds": list(sahi_match_thresholds), "postprocess_match_metric": sahi_match_metric,
                            "postprocess_class_threshold": 0.25, "postprocess_type": "NMS"}
        self.sahi_config.update({"slice_height": 640, "slice_width": 640, "overlap_ratio": 0.25} if slicing_strategy == "uniform" else {"overlap_ratio": 0.2})
        self.detection_model = self.face_enhancer = None
        if load_model:
            from utils.yolo_wrapper import YOLOv11PoseDetectionModel
            self.detection_model = YOLOv11PoseDetectionModel(model_path=model_path, confidence_threshold=self.inference_confidence, device=device, load_at_init=True)
            if self.use_enhancer:
                try:
                    from utils.enhancer import FaceEnhancer
                    self.face_enhancer = FaceEnhancer(model_name="RealESRGAN_x2plus")
                    print(f"   ✓ Enhancer loaded! (Scale: {self.face_enhancer.scale}x)")
                except Exception as e:
                    print(f"   ❌ Failed to load enhancer: {e}")
                    self.use_enhancer = False
        self._build_mode_string()

    def _build_mode_string(self):
        """:128-143 (the dual evaluator joins with ' + ')"""
        mode_parts = []
        if self.use_enhancer:
            mode_parts.append(f"BOUNDED-ENHANCE (<{self.face_size_threshold}px)" if self.bounded_enhancement else "FULL-ENHANCE")
        mode_parts.append(f"SAHI ({self.slicing_strategy})" if self.use_sahi else "BASELINE")
        self.mode_string = " + ".join(mode_parts) if mode_parts else "BASELINE"

    def quick_face_analysis(self, img):
        """:145-170"""
        if img is None:
            return False, "Image load failed", {}
        results = self.detection_model.model(img, conf=0.05, verbose=False)
        if len(results) == 0 or results[0].boxes is None or len(results[0].boxes) == 0:
            return True, "No faces detected", {}
        face_sizes = []
        for i in range(len(results[0].boxes)):
            xyxy = results[0].boxes.xyxy[i].cpu().numpy()
            face_sizes.append(max(xyxy[2] - xyxy[0], xyxy[3] - xyxy[1]))
        small_face_ratio = sum(1 for size in face_sizes if size < self.face_size_threshold) / len(face_sizes)
        if small_face_ratio > 0.5 or np.mean(face_sizes) < self.face_size_threshold:
            return True, "Small faces detected", {}
        return False, "Faces are large enough", {}

    def get_slice_size_adaptive(self, w, h):
        max_dim = max(w, h)
        return 512 if max_dim > 2500 else 416 if max_dim > 1500 else 320

    def run_inference(self, img_path):
        """:182-270: cached per path; [{'bbox': [x, y, w, h], 'confidence': c}] in the ORIGINAL picture's coordinates"""
        if img_path in self.prediction_cache:
            return self.prediction_cache[img_path]
        import cv2
        img = cv2.imread(img_path)
        if img is None:
            return []
        inference_img, was_enhanced = img, False
        if self.use_enhancer and self.face_enhancer:
            enhance_decision = self.quick_face_analysis(img)[0] if self.bounded_enhancement else True
            if enhance_decision:
                enhanced_image, success = self.face_enhancer.enhance_image(img)
                if success:
                    inference_img, was_enhanced = enhanced_image, True
                    self.enhancement_stats["enhanced_images"] += 1
            else:
                self.enhancement_stats["skipped_images"] += 1
        pred_boxes = []
        if self.use_sahi:
            from sahi.predict import get_sliced_prediction
            h, w = inference_img.shape[:2]
            if self.slicing_strategy == "uniform":
                slice_h, slice_w = self.sahi_config["slice_height"], self.sahi_config["slice_width"]
            else:
                slice_h = slice_w = self.get_slice_size_adaptive(w, h)
            result = get_sliced_prediction(inference_img, self.detection_model, slice_height=slice_h, slice_width=slice_w,
                                           overlap_height_ratio=self.sahi_config["overlap_ratio"], overlap_width_ratio=self.sahi_config["overlap_ratio"],
                                           postprocess_type=self.sahi_config["postprocess_type"], postprocess_match_metric=self.sahi_config["postprocess_match_metric"],
                                           postprocess_match_threshold=self.sahi_config["postprocess_match_thresholds"][0], postprocess_class_agnostic=True, verbose=0)
            for det in result.object_prediction_list:
                x, y, w, h = det.bbox.to_xywh()
                pred_boxes.append({"bbox": [x, y, w, h], "confidence": det.score.value})
        else:
            results = self.detection_model.model(inference_img, conf=self.inference_confidence, verbose=False)
            boxes = results[0].boxes
            if len(boxes) > 0:
                xyxy, confs = boxes.xyxy.cpu().numpy(), boxes.conf.cpu().numpy()
                for i in range(len(boxes)):
                    x1, y1, x2, y2 = xyxy[i]
                    pred_boxes.append({"bbox": [x1, y1, x2 - x1, y2 - y1], "confidence": confs[i]})
        if was_enhanced and self.face_enhancer.scale > 1:
            scale = self.face_enhancer.scale
            for pred in pred_boxes:
                pred["bbox"] = [coord / scale for coord in pred["bbox"]]
        self.prediction_cache[img_path] = pred_boxes
        return pred_boxes

    _DIFFICULTY = {"large_clear": ["easy", "medium", "hard"], "large_degraded": ["medium", "hard"], "medium_clear": ["medium", "hard"]}

    def map_subcategory_to_difficulty(self, category):
        """Which WIDER difficulty sets a sub-category counts towards (:317-332): everything is 'hard', three of them also 'medium', one 'easy'."""
        return list(self._DIFFICULTY.get(category, ["hard"]))

    def calculate_average_precision(self, all_detections, total_gt):
        """11-point interpolated AP (:293-315). The detections are ordered by a STABLE descending sort on confidence, like list.sort(reverse=True)."""
        if total_gt == 0 or not all_detections:
            return 0.0
        all_detections.sort(key=lambda det: det["confidence"], reverse=True)
        hit = np.fromiter((bool(det["is_tp"]) for det in all_detections), dtype=bool, count=len(all_detections))
        tp, fp = np.cumsum(hit), np.cumsum(~hit)
        recall, precision = tp / total_gt, tp / (tp + fp)
        ap = 0.0
        for level in np.arange(0., 1.1, 0.1):
            reached = recall >= level
            ap += (np.max(precision[reached]) if reached.any() else 0) / 11.0
        return ap

    def evaluate_single_set(self, category_type, category_name, valid_categories):
        preds, faces, valid, used = [], [], [], []
        total_gt = 0
        for img_path, gt_data in self.subcategory_gt.items():
            idx = sorted({int(i) for cat in valid_categories for i in gt_data.get(cat, [])})
            if not idx:
                continue
            all_faces = np.asarray([f["bbox"] for f in gt_data["all_faces"]], np.float64).reshape(-1, 4)
            flag = np.zeros(len(all_faces), np.uint8)
            flag[idx] = 1
            total_gt += len(idx)
            p = self.predictions.get(img_path, [])
            preds.append(np.asarray([[*d["bbox"], d["confidence"]] for d in p], np.float64).reshape(-1, 5))
            faces.append(all_faces); valid.append(flag); used.append(len(idx))
        flags = _lib.eval_dual_match(preds, faces, valid, self.iou_threshold) if preds else []
        all_detections, false_negatives = [], 0
        for p, fl, n_valid in zip(preds, flags, used):
            false_negatives += n_valid - int(np.count_nonzero(fl == 1))
            all_detections += [{"confidence": float(c), "is_tp": bool(f == 1)} for c, f in zip(p[:, 4], fl) if f != 2]
        ap = self.calculate_average_precision(all_detections, total_gt)
        filtered = [d for d in all_detections if d["confidence"] >= self.global_confidence]
        tp = sum(1 for d in filtered if d["is_tp"])
        precision = tp / len(filtered) if filtered else 0
        recall = tp / total_gt if total_gt > 0 else 0
        f1 = 2 * (precision * recall) / (precision + recall) if (precision + recall) > 0 else 0
        return {"category": category_name, "total_gt": total_gt, "total_pred": len(filtered), "true_positives": tp, "false_positives": len(filtered) - tp,
                "false_negatives": false_negatives, "precision": precision, "recall": recall, "f1_score": f1, "ap": ap}

    def run(self):
        sub = {c: self.evaluate_single_set("subcategory", c, [c]) for c in self.subcategories}
        diff_sets = {"easy": ["large_clear"], "medium": ["large_clear", "large_degraded", "medium_clear"], "hard": list(self.subcategories)}
        diff = {d: self.evaluate_single_set("difficulty", d, cats) for d, cats in diff_sets.items()}
        return sub, diff
