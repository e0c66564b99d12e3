"""Generates tests/golden/wrapper_expected.json: outputs of the REFERENCE's own pure-Python helpers on seeded inputs, for the rows of
SURVEY.md §8 whose arithmetic lives in /root/reference itself (a8 keypoint attach, f3 detection summary, a13 enhancement summary / model
table). The modules cannot be imported (cv2, sahi, ultralytics, basicsr absent — ordinary ImportError), so — as in make_eval_fixtures.py —
the named functions are compiled from the files' syntax trees, unchanged, with nothing added and no stand-in for a missing library:
  utils/yolo_wrapper.py    YOLOv11PoseDetectionModel.attach_keypoints_to_predictions, ._calculate_iou      (:168-217)
  utils/visualization.py   create_detection_summary (+ the FACE_KEYPOINT_NAMES table it reads)             (:5-12, 225-285)
  utils/enhancer.py        create_enhancement_summary, get_available_models                                (:409-480)
They are driven with plain Python objects that carry the attributes they read (bbox.to_voc_bbox(), score.value, keypoints, ...).
Needs /root/reference; run here, never on the GPU box. Only data is written (inputs + outputs).
"""
import ast
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

REF = Path("/root/reference/utils")
OUT = Path(__file__).resolve().parent / "wrapper_expected.json"


def compile_subset(path: Path, functions=(), cls=None, methods=(), assigns=()):
    tree = ast.parse(path.read_text(encoding="utf-8"))
    body = []
    for n in tree.body:
        if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id in assigns for t in n.targets):
            body.append(n)
        if isinstance(n, ast.FunctionDef) and n.name in functions:
            body.append(n)
        if isinstance(n, ast.ClassDef) and n.name == cls:
            fns = [m for m in n.body if isinstance(m, ast.FunctionDef) and m.name in methods]
            assert sorted(f.name for f in fns) == sorted(methods)
            body.append(ast.ClassDef(name=cls, bases=[], keywords=[], body=fns, decorator_list=[]))
    mod = ast.Module(body=body, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"os": os, "np": np, "__name__": "ref_subset"}
    exec(compile(mod, str(path), "exec"), ns)
    return ns


class Box:
    def __init__(self, xyxy):
        self.xyxy = [int(v) for v in xyxy]

    def to_voc_bbox(self):
        return list(self.xyxy)

    def to_xyxy(self):
        return list(self.xyxy)


class Score:
    def __init__(self, v):
        self.value = float(v)


class Pred:
    def __init__(self, xyxy, score, keypoints=None):
        self.bbox, self.score = Box(xyxy), Score(score)
        if keypoints is not None:
            self.keypoints = keypoints


class Result:
    def __init__(self, preds):
        self.object_prediction_list = preds


def main():
    rng = np.random.default_rng(424242)
    out = {}
    # ---- a8: keypoint attach ------------------------------------------------------------------------------------------------------
    W = compile_subset(REF / "yolo_wrapper.py", cls="YOLOv11PoseDetectionModel", methods=["attach_keypoints_to_predictions", "_calculate_iou"])["YOLOv11PoseDetectionModel"]
    cases = []
    for k in range(6):
        n_cache, n_pred = int(rng.integers(0, 12)), int(rng.integers(0, 10))
        boxes = []
        for _ in range(n_cache):
            x, y, w, h = int(rng.integers(0, 500)), int(rng.integers(0, 300)), int(rng.integers(5, 80)), int(rng.integers(5, 80))
            boxes.append([x, y, x + w, y + h])
        if n_cache > 2 and k % 2:
            boxes[1] = list(boxes[0])                                   # duplicate key: the later entry overwrites the earlier one
        cache, order = {}, []
        for i, b in enumerate(boxes):
            cache[f"{b[0]}_{b[1]}_{b[2]}_{b[3]}"] = i                    # the "keypoints" are the entry's index: identity is what is recorded
        preds = []
        for _ in range(n_pred):
            if n_cache and rng.random() < 0.75:
                b = list(boxes[int(rng.integers(n_cache))])
                if rng.random() < 0.6:                                  # merged boxes rarely equal a slice box: shift / grow it
                    d = rng.integers(-12, 13, 4)
                    b = [b[0] + int(d[0]), b[1] + int(d[1]), max(b[0] + int(d[0]) + 1, b[2] + int(d[2])), max(b[1] + int(d[1]) + 1, b[3] + int(d[3]))]
            else:
                x, y = int(rng.integers(0, 500)), int(rng.integers(0, 300))
                b = [x, y, x + int(rng.integers(5, 80)), y + int(rng.integers(5, 80))]
            preds.append(b)
        m = W.__new__(W)
        m.keypoints_cache = dict(cache)
        objs = [Pred(b, 0.9) for b in preds]
        m.attach_keypoints_to_predictions(objs)
        cases.append({"cache_keys": list(cache.keys()), "cache_vals": list(cache.values()), "preds": preds,
                      "attached": [getattr(o, "keypoints", -1) for o in objs]})
    pairs = [[[int(v) for v in rng.integers(0, 60, 2)] + [int(v) for v in rng.integers(60, 120, 2)], [int(v) for v in rng.integers(0, 100, 2)] + [int(v) for v in rng.integers(100, 160, 2)]]
             for _ in range(40)] + [[[0, 0, 10, 10], [10, 10, 20, 20]], [[0, 0, 10, 10], [0, 0, 10, 10]], [[5, 5, 5, 5], [5, 5, 5, 5]]]
    m = W.__new__(W)
    out["attach"] = cases
    out["iou_pairs"] = pairs
    out["iou"] = [m._calculate_iou(a, b) for a, b in pairs]

    # ---- f3: detection summary text ---------------------------------------------------------------------------------------------------
    V = compile_subset(REF / "visualization.py", functions=["create_detection_summary"], assigns=["FACE_KEYPOINT_NAMES"])
    summaries = []
    with tempfile.TemporaryDirectory() as td:
        for k, n in enumerate([0, 1, 4]):
            dets = []
            for i in range(n):
                x, y = float(rng.uniform(0, 900)), float(rng.uniform(0, 500))
                box = [x, y, x + float(rng.uniform(10, 90)), y + float(rng.uniform(10, 90))]
                kp = [[float(v) for v in rng.uniform(0, 1000, 2)] + [float(rng.uniform(0, 1))] for _ in range(5)] if i % 2 == 0 else None
                dets.append({"box": box, "score": float(rng.uniform(0.3, 0.99)), "kpts": kp})
            preds = [Pred(d["box"], d["score"], None if d["kpts"] is None else np.asarray(d["kpts"])) for d in dets]
            for p, d in zip(preds, dets):
                p.bbox.xyxy = d["box"]                                  # floats: the function applies int() itself
                p.bbox.to_xyxy = (lambda b=d["box"]: list(b))
            path = os.path.join(td, "out", f"summary{k}.txt")
            V["create_detection_summary"](Result(preds), f"/data/images/photo_{k}.jpg", 1.2345 + k, path, 1920 + k, 1080, 512, 512)
            summaries.append({"dets": dets, "image_path": f"/data/images/photo_{k}.jpg", "time": 1.2345 + k, "size": [1920 + k, 1080], "slice": [512, 512],
                              "text": open(path, encoding="utf-8").read()})
    out["detection_summary"] = summaries

    # ---- a13: enhancement summary text + model table ------------------------------------------------------------------------------------
    E = compile_subset(REF / "enhancer.py", functions=["create_enhancement_summary", "get_available_models"])
    out["available_models"] = E["get_available_models"]()
    results = {"statistics": {"total_files": 3, "successful": 2, "failed": 1, "total_time": 4.5678},
               "enhancement_info": [{"original_size": (32, 40), "enhanced_size": (128, 160), "original_path": "/tmp/crops/face_crop_1_conf_0.91.jpg", "scale_factor": 4,
                                     "output_path": "/tmp/crops_enhanced/enhanced_face_crop_1_conf_0.91.jpg"},
                                    {"original_size": (24, 24), "enhanced_size": (96, 96), "original_path": "/tmp/crops/face_crop_2_conf_0.55.jpg", "scale_factor": 4,
                                     "output_path": "/tmp/crops_enhanced/enhanced_face_crop_2_conf_0.55.jpg"}],
               "failed_files": ["/tmp/crops/broken.jpg"], "enhanced_files": []}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sub", "enh.txt")
        E["create_enhancement_summary"](results, path)
        text = open(path, encoding="utf-8").read()
    out["enhancement_summary"] = {"results": json.loads(json.dumps(results)), "text": "\n".join(l for l in text.split("\n") if not l.startswith("Generated:"))}

    OUT.write_text(json.dumps(out, indent=1, ensure_ascii=False), encoding="utf-8")
    print(f"wrote {OUT} ({OUT.stat().st_size / 1e3:.1f} kB)")


if __name__ == "__main__":
    sys.exit(main())
