"""The parts of the path whose arithmetic lives in /root/reference ITSELF, pinned by the reference's own functions run on seeded inputs
(tests/golden/wrapper_expected.json, written by tests/golden/make_wrapper_fixtures.py): the keypoint attach of the SAHI adapter (a8:
utils/yolo_wrapper.py:168-217), the detection summary text (f3: utils/visualization.py:225-285), the enhancement summary text and the
model table (a13: utils/enhancer.py:409-480). Both the oracle (oracle/wrapper_ref.py) and the shipped compat modules must reproduce them."""
import json
import os

import numpy as np
import pytest

import ffp_amd.compat

FX = os.path.join(os.path.dirname(__file__), "golden", "wrapper_expected.json")


@pytest.fixture(scope="module")
def fx():
    with open(FX, encoding="utf-8") as fh:
        return json.load(fh)


class Box:
    def __init__(self, b):
        self.b = list(b)

    def to_voc_bbox(self):
        return list(self.b)

    def to_xyxy(self):
        return list(self.b)


class Pred:
    def __init__(self, b, score=0.9, keypoints=None):
        self.bbox = Box(b)
        self.score = type("S", (), {"value": float(score)})()
        if keypoints is not None:
            self.keypoints = keypoints


def test_iou_and_attach_match_the_reference(fx):
    from oracle import wrapper_ref
    ffp_amd.compat.install()
    from utils.yolo_wrapper import YOLOv11PoseDetectionModel
    m = YOLOv11PoseDetectionModel.__new__(YOLOv11PoseDetectionModel)
    got = [wrapper_ref.iou(a, b) for a, b in fx["iou_pairs"]]
    assert got == fx["iou"]
    assert [m._calculate_iou(a, b) for a, b in fx["iou_pairs"]] == fx["iou"]
    for case in fx["attach"]:
        cache = dict(zip(case["cache_keys"], case["cache_vals"]))
        exp = case["attached"]
        o = wrapper_ref.attach(case["preds"], cache)
        assert [(-1 if v is None else v) for v in o] == exp
        m.keypoints_cache = dict(cache)
        preds = [Pred(b) for b in case["preds"]]
        m.attach_keypoints_to_predictions(preds)
        assert [getattr(p, "keypoints", -1) for p in preds] == exp
    assert any(v != -1 for c in fx["attach"] for v in c["attached"]) and any(v == -1 for c in fx["attach"] for v in c["attached"])


def test_detection_summary_text_is_the_references(fx, tmp_path):
    ffp_amd.compat.install()
    from utils.visualization import create_detection_summary
    for k, s in enumerate(fx["detection_summary"]):
        preds = [Pred(d["box"], d["score"], None if d["kpts"] is None else np.asarray(d["kpts"])) for d in s["dets"]]
        res = type("R", (), {"object_prediction_list": preds})()
        out = tmp_path / "o" / f"s{k}.txt"
        create_detection_summary(res, s["image_path"], s["time"], str(out), s["size"][0], s["size"][1], s["slice"][0], s["slice"][1])
        assert out.read_text(encoding="utf-8") == s["text"]


def test_enhancement_summary_and_model_table_are_the_references(fx, tmp_path):
    ffp_amd.compat.install()
    from utils.enhancer import create_enhancement_summary, get_available_models
    out = tmp_path / "e" / "enh.txt"
    create_enhancement_summary(fx["enhancement_summary"]["results"], str(out))
    text = "\n".join(l for l in out.read_text(encoding="utf-8").split("\n") if not l.startswith("Generated:"))
    assert text == fx["enhancement_summary"]["text"]
    mine = get_available_models()
    for name, info in fx["available_models"].items():
        assert name in mine
        for key, v in info.items():
            assert mine[name][key] == v, (name, key)
