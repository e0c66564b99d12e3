"""eval.eval_dual's evaluation core with the reference's names (/root/reference/eval/eval_dual.py:272-433): sub-category and difficulty
metrics from `subcategory_gt.json`. The matching of every image (IoU of each prediction against the valid and the ignored faces,
first-best assignment, :369-399) is one GPU launch (ffp_eval_dual_match); the 11-point AP and precision / recall / F1 are the reference's
numpy expressions on the resulting flags. Predictions are handed in per image path ({path: [{'bbox': [x, y, w, h], 'confidence': c}]}),
e.g. from `OfficialWiderFaceEvaluator._run_single_inference` or the pipeline.
"""
import json

import numpy as np

from ffp_amd import _lib


class DualWiderFaceEvaluator:
    subcategories = ["large_clear", "large_degraded", "medium_clear", "medium_degraded", "small_clear", "small_degraded"]

    def __init__(self, subcategory_file=None, subcategory_gt=None, predictions=None, iou_threshold=0.5, global_confidence=0.25):
        if subcategory_gt is None:
            with open(subcategory_file, "r") as f:
                subcategory_gt = json.load(f)
        self.subcategory_gt = subcategory_gt
        self.predictions = predictions or {}
        self.iou_threshold, self.global_confidence = iou_threshold, global_confidence

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
