"""TEST INFRASTRUCTURE — CPU restatement of the reference's two WIDER FACE evaluation protocols.
Only tests/ and bench tooling may import this; the product path is csrc/eval.hip behind ffp_eval_*.

(E1) "official" protocol — /root/reference/eval/eval_official_widerface.py:
       _voc_ap :282-300, _image_eval :302-347, _img_pr_info :349-375, _dataset_pr_info :377-395, _evaluate_setting :397-453.
     `bbox_overlaps` is imported there from the WiderFace-Evaluation Cython extension (:24-33), which is NOT in /root/reference;
     it is restated here from the published box_overlaps.pyx (Fast R-CNN lineage): inclusive pixel boxes, +1 on widths/heights.
(E2) "dual" protocol — /root/reference/eval/eval_dual.py:
       calculate_iou :272-291, calculate_average_precision :293-315 (11-point), evaluate_single_set :334-419 (matching, ignore rule,
       precision / recall / F1 at the global confidence).

Pinning: tests/golden/make_eval_fixtures.py runs the reference's OWN pure-numpy methods of these files (compiled from the files'
syntax trees, the modules themselves cannot be imported: cv2, seaborn, sahi, ultralytics are absent) on seeded inputs;
tests/test_eval_oracle.py checks this restatement against those outputs. `_image_eval` needs the absent Cython function and
is therefore pinned only through its numpy parts; `bbox_overlaps` itself is parity-unpinned (upstream text, restated).
"""
import numpy as np


# ---- (E1) official protocol ---------------------------------------------------------------------------------------------------
def bbox_overlaps(boxes: np.ndarray, query: np.ndarray) -> np.ndarray:
    """IoU matrix [N boxes][K query], inclusive-pixel convention (x2 - x1 + 1), float64 (WiderFace-Evaluation box_overlaps.pyx)."""
    boxes = np.asarray(boxes, np.float64).reshape(-1, 4)
    query = np.asarray(query, np.float64).reshape(-1, 4)
    out = np.zeros((boxes.shape[0], query.shape[0]), np.float64)
    for k in range(query.shape[0]):
        qa = (query[k, 2] - query[k, 0] + 1) * (query[k, 3] - query[k, 1] + 1)
        for n in range(boxes.shape[0]):
            iw = min(boxes[n, 2], query[k, 2]) - max(boxes[n, 0], query[k, 0]) + 1
            if iw > 0:
                ih = min(boxes[n, 3], query[k, 3]) - max(boxes[n, 1], query[k, 1]) + 1
                if ih > 0:
                    ua = (boxes[n, 2] - boxes[n, 0] + 1) * (boxes[n, 3] - boxes[n, 1] + 1) + qa - iw * ih
                    out[n, k] = iw * ih / ua
    return out


def image_eval(pred: np.ndarray, gt: np.ndarray, ignore: np.ndarray, iou_thresh: float = 0.5):
    """eval_official_widerface.py:302-347. pred [N][5] = x, y, w, h, score (in the given order); gt [G][4] = x, y, w, h;
    ignore[g] == 1: evaluate this face, 0: a match with it drops the proposal. -> (pred_recall [N], proposal_list [N])."""
    _pred = np.array(pred, np.float64, copy=True)
    _gt = np.array(gt, np.float64, copy=True)
    pred_recall = np.zeros(_pred.shape[0])
    recall_list = np.zeros(_gt.shape[0])
    proposal_list = np.ones(_pred.shape[0])
    _pred[:, 2] += _pred[:, 0]
    _pred[:, 3] += _pred[:, 1]
    _gt[:, 2] += _gt[:, 0]
    _gt[:, 3] += _gt[:, 1]
    overlaps = bbox_overlaps(_pred[:, :4], _gt)
    for h in range(_pred.shape[0]):
        row = overlaps[h]
        mx, idx = row.max(), row.argmax()
        if mx >= iou_thresh:
            if ignore[idx] == 0:
                recall_list[idx] = -1
                proposal_list[h] = -1
            elif recall_list[idx] == 0:
                recall_list[idx] = 1
        pred_recall[h] = np.count_nonzero(recall_list == 1)
    return pred_recall, proposal_list


def img_pr_info(thresh_num: int, pred_info: np.ndarray, proposal_list: np.ndarray, pred_recall: np.ndarray) -> np.ndarray:
    """eval_official_widerface.py:349-375: per score threshold t, (#valid proposals, matched faces) up to the LAST prediction with
    score >= 1 - (t+1)/thresh_num."""
    out = np.zeros((thresh_num, 2), np.float64)
    for t in range(thresh_num):
        thresh = 1 - (t + 1) / thresh_num
        r = np.where(pred_info[:, 4] >= thresh)[0]
        if len(r):
            r = r[-1]
            out[t, 0] = np.count_nonzero(proposal_list[:r + 1] == 1)
            out[t, 1] = pred_recall[r]
    return out


def dataset_pr_info(thresh_num: int, pr_curve: np.ndarray, count_face: int) -> np.ndarray:
    """eval_official_widerface.py:377-395: column 0 precision = matched / proposals (0 when there are none), column 1 recall."""
    out = np.zeros((thresh_num, 2))
    for i in range(thresh_num):
        out[i, 0] = pr_curve[i, 1] / pr_curve[i, 0] if pr_curve[i, 0] != 0 else 0
        out[i, 1] = pr_curve[i, 1] / count_face
    return out


def voc_ap(rec: np.ndarray, prec: np.ndarray) -> float:
    """eval_official_widerface.py:282-300: area under the monotone precision envelope."""
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def evaluate_setting(images, thresh_num: int = 1000, iou_thresh: float = 0.5):
    """eval_official_widerface.py:397-453 over a flat list of images, each a dict {pred [N][5] | empty, gt [G][4], keep: 1-based
    indices of the faces this setting evaluates}. -> (ap, recall [T], precision [T], pr_counts [T][2], count_face)."""
    count_face = 0
    pr_curve = np.zeros((thresh_num, 2), np.float64)
    for im in images:
        pred, gt, keep = np.asarray(im["pred"], np.float64), np.asarray(im["gt"], np.float64), np.asarray(im["keep"], np.int64)
        count_face += len(keep)
        if len(gt) == 0 or len(pred) == 0:
            continue
        ignore = np.zeros(gt.shape[0])
        if len(keep):
            ignore[keep - 1] = 1
        pred_recall, proposal_list = image_eval(pred, gt, ignore, iou_thresh)
        pr_curve += img_pr_info(thresh_num, pred, proposal_list, pred_recall)
    counts = pr_curve.copy()
    pr = dataset_pr_info(thresh_num, pr_curve, count_face)
    return voc_ap(pr[:, 1], pr[:, 0]), pr[:, 1], pr[:, 0], counts, count_face


# ---- (E2) dual protocol -------------------------------------------------------------------------------------------------------
def calculate_iou(box1, box2) -> float:
    """eval_dual.py:272-291: x, y, w, h boxes, continuous coordinates (no +1), 0 when disjoint or the union is empty."""
    x1, y1, w1, h1 = box1
    x2, y2, w2, h2 = box2
    ix1, iy1 = max(x1, x2), max(y1, y2)
    ix2, iy2 = min(x1 + w1, x2 + w2), min(y1 + h1, y2 + h2)
    if ix2 < ix1 or iy2 < iy1:
        return 0.0
    inter = (ix2 - ix1) * (iy2 - iy1)
    union = (w1 * h1) + (w2 * h2) - inter
    return inter / union if union > 0 else 0.0


def match_image(pred: np.ndarray, gt_valid: np.ndarray, gt_ignored: np.ndarray, iou_thresh: float = 0.5) -> np.ndarray:
    """eval_dual.py:369-399 for one image: per prediction (in the given order) 1 = true positive, 0 = false positive,
    2 = dropped (not a TP and overlaps an ignored face). The best valid face is the FIRST one with the strictly largest IoU."""
    matched = np.zeros(len(gt_valid), bool)
    out = np.zeros(len(pred), np.int32)
    for i, p in enumerate(pred):
        best, bi = 0, -1
        for g, face in enumerate(gt_valid):
            iou = calculate_iou(p[:4], face)
            if iou > best:
                best, bi = iou, g
        if best >= iou_thresh and bi != -1 and not matched[bi]:
            matched[bi] = True
            out[i] = 1
        else:
            for face in gt_ignored:
                if calculate_iou(p[:4], face) >= iou_thresh:
                    out[i] = 2
                    break
    return out


def average_precision_11pt(conf: np.ndarray, is_tp: np.ndarray, total_gt: int) -> float:
    """eval_dual.py:293-315: stable sort by confidence (descending), cumulative TP / FP, 11-point interpolation."""
    if total_gt == 0 or len(conf) == 0:
        return 0.0
    order = sorted(range(len(conf)), key=lambda i: conf[i], reverse=True)      # list.sort(reverse=True) keeps equal keys in order
    tp = np.cumsum([bool(is_tp[i]) for i in order])
    fp = np.cumsum([not bool(is_tp[i]) for i in order])
    recalls = tp / total_gt
    precisions = tp / (tp + fp)
    ap = 0.0
    for t in np.arange(0., 1.1, 0.1):
        p = 0 if np.sum(recalls >= t) == 0 else np.max(precisions[recalls >= t])
        ap += p / 11.0
    return float(ap)


def evaluate_single_set(images, iou_thresh: float = 0.5, global_confidence: float = 0.25) -> dict:
    """eval_dual.py:334-419 over a list of images {pred [N][5] = x, y, w, h, confidence; faces [F][4]; valid: indices of the faces
    of this category set}. Images without a valid face are skipped entirely (their predictions do not count as false positives)."""
    total_gt, fn = 0, 0
    conf, tp = [], []
    for im in images:
        valid = sorted(set(int(i) for i in im["valid"]))            # eval_dual.py:352 list(set(...)): order is not defined there; see DESIGN.md
        if not valid:
            continue
        faces = np.asarray(im["faces"], np.float64).reshape(-1, 4)
        gt_valid = faces[valid]
        gt_ign = faces[[i for i in range(len(faces)) if i not in valid]]
        total_gt += len(gt_valid)
        pred = np.asarray(im["pred"], np.float64).reshape(-1, 5)
        flags = match_image(pred, gt_valid, gt_ign, iou_thresh)
        fn += len(gt_valid) - int(np.count_nonzero(flags == 1))
        for f, p in zip(flags, pred):
            if f != 2:
                conf.append(p[4])
                tp.append(f == 1)
    conf, tp = np.asarray(conf, np.float64), np.asarray(tp, bool)
    ap = average_precision_11pt(conf, tp, total_gt)
    keep = conf >= global_confidence
    n_keep, n_tp = int(keep.sum()), int(tp[keep].sum())
    precision = n_tp / n_keep if n_keep else 0
    recall = n_tp / total_gt if total_gt > 0 else 0
    f1 = 2 * (precision * recall) / (precision + recall) if (precision + recall) > 0 else 0
    return {"total_gt": total_gt, "total_pred": n_keep, "true_positives": n_tp, "false_positives": n_keep - n_tp, "false_negatives": fn,
            "precision": precision, "recall": recall, "f1_score": f1, "ap": ap}
