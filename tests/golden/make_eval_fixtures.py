"""Generates tests/golden/eval_expected.npz: outputs of the REFERENCE's own evaluation arithmetic on seeded inputs.

The two evaluation modules cannot be imported here (cv2, seaborn, matplotlib, sahi, ultralytics are absent — ordinary ImportError), so this
script parses the files, compiles ONLY the named methods from their syntax trees (unchanged, nothing added, no stand-in for a missing
library) and calls them on seeded inputs:
  /root/reference/eval/eval_dual.py               calculate_iou, calculate_average_precision, evaluate_single_set, map_subcategory_to_difficulty,
                                                  get_slice_size_adaptive
  /root/reference/eval/eval_official_widerface.py _voc_ap, _img_pr_info, _dataset_pr_info, _get_slice_size_adaptive
`evaluate_single_set` is driven through a plain object that carries the attributes the method reads (ground truth, thresholds) and whose
`run_inference` returns the seeded predictions — inputs, not arithmetic. `_image_eval` / `_evaluate_setting` call the Cython `bbox_overlaps`
that is not in the tree and are NOT run. Needs /root/reference; run here, never on the GPU box. Only data is written (inputs + outputs).
"""
import ast
import os
import sys
import types
from collections import defaultdict
from pathlib import Path

import numpy as np

REF = Path("/root/reference/eval")
OUT = Path(__file__).resolve().parent / "eval_expected.npz"


def methods_of(path: Path, cls: str, names):
    tree = ast.parse(path.read_text(encoding="utf-8"))
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    fns = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(f.name for f in fns) == sorted(names), [f.name for f in fns]
    mod = ast.Module(body=[ast.ClassDef(name=cls, bases=[], keywords=[], body=fns, decorator_list=[])], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np, "__name__": "ref_eval_subset"}
    exec(compile(mod, str(path), "exec"), ns)
    return ns[cls]


def random_image(rng, n_faces, n_pred, integer):
    faces = np.zeros((n_faces, 4))
    faces[:, 0] = rng.uniform(0, 900, n_faces)
    faces[:, 1] = rng.uniform(0, 600, n_faces)
    faces[:, 2:] = rng.uniform(6, 120, (n_faces, 2))
    pred = np.zeros((n_pred, 5))
    for i in range(n_pred):
        if n_faces and rng.random() < 0.7:                       # a jittered copy of a face (sometimes an exact duplicate of an earlier one)
            f = faces[rng.integers(n_faces)]
            pred[i, :4] = f + rng.normal(0, 0.08, 4) * f[[2, 3, 2, 3]] * (rng.random() < 0.8)
        else:
            pred[i, :2] = rng.uniform(0, 900, 2)
            pred[i, 2:4] = rng.uniform(6, 120, 2)
        pred[i, 4] = round(float(rng.uniform(0.01, 1.0)), 2 if rng.random() < 0.5 else 6)     # repeated confidences exercise the stable sort
    if integer:
        faces, pred[:, :4] = np.round(faces), np.round(pred[:, :4])
    pred[:, 2:4] = np.maximum(pred[:, 2:4], 1)
    return faces, pred


def main():
    rng = np.random.default_rng(20250607)
    out = {}
    Dual = methods_of(REF / "eval_dual.py", "DualWiderFaceEvaluator", ["calculate_iou", "calculate_average_precision", "evaluate_single_set",
                                                                        "map_subcategory_to_difficulty", "get_slice_size_adaptive"])
    Off = methods_of(REF / "eval_official_widerface.py", "OfficialWiderFaceEvaluator", ["_voc_ap", "_img_pr_info", "_dataset_pr_info", "_get_slice_size_adaptive"])
    dd, oo = Dual.__new__(Dual), Off.__new__(Off)
    dims = [(640, 480), (1500, 1500), (1501, 900), (2500, 100), (2501, 2000), (4160, 2340), (100, 2600)]
    out["adaptive_dims"] = np.asarray(dims, np.int64)
    out["adaptive_dual"] = np.asarray([dd.get_slice_size_adaptive(w, h) for w, h in dims], np.int64)
    out["adaptive_official"] = np.asarray([oo._get_slice_size_adaptive(w, h) for w, h in dims], np.int64)
    subcats = ["large_clear", "large_degraded", "medium_clear", "medium_degraded", "small_clear", "small_degraded"]
    order = ["easy", "medium", "hard"]
    out["difficulty_map"] = np.asarray([[int(d in dd.map_subcategory_to_difficulty(c)) for d in order] for c in subcats], np.int64)

    # ---- eval_dual: IoU pairs
    d = Dual.__new__(Dual)
    b1 = np.concatenate([rng.uniform(0, 50, (400, 2)), rng.uniform(0, 40, (400, 2))], 1)
    b2 = np.concatenate([rng.uniform(0, 50, (400, 2)), rng.uniform(0, 40, (400, 2))], 1)
    b2[:40] = b1[:40]                                  # identical boxes
    b2[40:60, :2] = b1[40:60, :2] + b1[40:60, 2:]      # touching corners
    b1[60:70, 2:] = 0                                  # empty boxes
    out["iou_b1"], out["iou_b2"] = b1, b2
    out["iou_out"] = np.asarray([d.calculate_iou(list(a), list(b)) for a, b in zip(b1, b2)], np.float64)

    # ---- eval_dual: whole evaluate_single_set on seeded datasets
    cats = ["large_clear", "large_degraded", "medium_clear", "small_hard"]
    for ds in range(4):
        n_img = [12, 40, 25, 6][ds]
        gt, canned, flat = {}, {}, []
        for k in range(n_img):
            nf = int(rng.integers(0, 9)) if ds != 3 else int(rng.integers(0, 3))
            npred = int(rng.integers(0, 14))
            faces, pred = random_image(rng, nf, npred, integer=(ds == 2))
            assign = rng.integers(0, len(cats), nf)
            name = f"ev{ds}/img{k}.jpg"
            gt[name] = {c: [int(i) for i in np.where(assign == ci)[0]] for ci, c in enumerate(cats)}
            gt[name]["all_faces"] = [{"bbox": [float(v) for v in f]} for f in faces]
            canned[str(Path("images") / name)] = [{"bbox": [float(v) for v in p[:4]], "confidence": float(p[4])} for p in pred]
            flat.append((name, faces, pred, assign))
        for si, valid_cats in enumerate([["large_clear"], ["large_clear", "large_degraded", "medium_clear"], cats]):
            ev = Dual.__new__(Dual)
            ev.use_enhancer = False
            ev.temp_enh_dir = Path("unused")
            ev.subcategory_gt = gt
            ev.images_path = Path("images")
            ev.prediction_cache = {}
            ev.enhancement_stats = defaultdict(int)
            ev.iou_threshold = 0.5
            ev.global_confidence = 0.25
            ev.run_inference = lambda p, canned=canned: canned[p]
            res = ev.evaluate_single_set("difficulty", f"set{si}", valid_cats)
            key = f"dual{ds}_{si}"
            out[key + "_res"] = np.asarray([res["total_gt"], res["total_pred"], res["true_positives"], res["false_positives"], res["false_negatives"],
                                            res["precision"], res["recall"], res["f1_score"], res["ap"]], np.float64)
            out[key + "_valid_cats"] = np.asarray([cats.index(c) for c in valid_cats], np.int64)
        out[f"dual{ds}_faces"] = np.concatenate([f for _, f, _, _ in flat] + [np.zeros((0, 4))], 0)
        out[f"dual{ds}_face_off"] = np.cumsum([0] + [len(f) for _, f, _, _ in flat]).astype(np.int64)
        out[f"dual{ds}_pred"] = np.concatenate([p for _, _, p, _ in flat] + [np.zeros((0, 5))], 0)
        out[f"dual{ds}_pred_off"] = np.cumsum([0] + [len(p) for _, _, p, _ in flat]).astype(np.int64)
        out[f"dual{ds}_assign"] = np.concatenate([a for _, _, _, a in flat] + [np.zeros((0,), np.int64)], 0).astype(np.int64)

    # ---- eval_dual: 11-point AP alone (ties in confidence, all-FP, all-TP)
    for k, (n, total) in enumerate([(50, 30), (7, 3), (200, 500), (5, 0), (12, 12)]):
        conf = np.round(rng.uniform(0, 1, n), 1 if k % 2 == 0 else 5)
        tp = rng.random(n) < (0.0 if k == 1 else 1.0 if k == 4 else 0.5)
        dets = [{"confidence": float(c), "is_tp": bool(t)} for c, t in zip(conf, tp)]
        out[f"ap11_{k}_conf"], out[f"ap11_{k}_tp"], out[f"ap11_{k}_total"] = conf, tp, np.int64(total)
        out[f"ap11_{k}_out"] = np.float64(d.calculate_average_precision(dets, total))

    # ---- eval_official_widerface: the pure-numpy parts
    o = Off.__new__(Off)
    for k, T in enumerate([1000, 1000, 37, 1000]):
        o.thresh_num = T
        n = [30, 1, 200, 64][k]
        score = rng.uniform(0, 1, n)
        if k != 2:
            score = np.sort(score)[::-1]                 # k == 2: unsorted scores ("last index with score >= t" then is not a count)
        if k == 3:
            score = np.round(score, 2)
        pred_info = np.concatenate([rng.uniform(0, 100, (n, 4)), score[:, None]], 1)
        proposal = np.where(rng.random(n) < 0.2, -1.0, 1.0)
        pred_recall = np.cumsum(rng.random(n) < 0.4).astype(np.float64)
        out[f"pr_{k}_T"], out[f"pr_{k}_pred"], out[f"pr_{k}_prop"], out[f"pr_{k}_rec"] = np.int64(T), pred_info, proposal, pred_recall
        out[f"pr_{k}_out"] = o._img_pr_info(pred_info, proposal, pred_recall)
        counts = np.stack([rng.integers(0, 50, T), rng.integers(0, 40, T)], 1).astype(np.float64)
        counts[::7, 0] = 0
        out[f"dpr_{k}_counts"], out[f"dpr_{k}_faces"] = counts, np.int64(123 + k)
        out[f"dpr_{k}_out"] = o._dataset_pr_info(counts, 123 + k)
        rec = np.sort(rng.uniform(0, 1, T)) if k != 1 else np.repeat(rng.uniform(0, 1, T // 4), 4)[:T]
        if k == 1:
            rec = np.sort(rec)
        prec = rng.uniform(0, 1, len(rec))
        out[f"ap_{k}_rec"], out[f"ap_{k}_prec"] = rec, prec
        out[f"ap_{k}_out"] = np.float64(o._voc_ap(rec, prec))

    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1e3:.1f} kB, {len(out)} arrays)")


if __name__ == "__main__":
    sys.exit(main())
