"""CPU tests of the drop-in boundary that need no GPU: the LITERAL import blocks of the two callers north_star names resolve
against the compat shims, the adapter's result conversion + keypoint side channel agree with the oracle
(/root/reference/utils/yolo_wrapper.py:84-217), and the presentation helpers write what the reference's scripts expect
(/root/reference/utils/visualization.py:78-285)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# pipeline_v4_yolo/app_yolo_sahi.py:1-17 (the sys.path line points at the reference's root; here: the shim directory)
APP_YOLO_SAHI_IMPORTS = """
import sys
import os
import time
import cv2
from pathlib import Path
from glob import glob

from sahi.predict import get_sliced_prediction
from utils.yolo_wrapper import YOLOv11PoseDetectionModel
from utils.visualization import (
    draw_detections, 
    save_face_crops, 
    create_detection_summary
)
"""
# pipeline_v1_detection_first/app_v1.py:1-14
APP_V1_IMPORTS = """
import os
import re
import time
import os, sys

from sahi.predict import get_sliced_prediction
from utils.insightface_wrapper import InsightFaceDetectionModel
from utils.visualization import draw_detections, save_face_crops, create_detection_summary
from utils.enhancer import FaceEnhancer, enhance_face_crops_batch, create_enhancement_summary
from PIL import Image
"""


@pytest.mark.parametrize("block", [APP_YOLO_SAHI_IMPORTS, APP_V1_IMPORTS], ids=["app_yolo_sahi", "app_v1"])
def test_literal_import_blocks_resolve(block):
    code = ("import sys; sys.path.insert(0, %r)\nimport ffp_amd\nfrom ffp_amd import compat\ncompat.install()\n" % ROOT) + block + (
        "\nimport cv2 as _c\nassert hasattr(_c, 'imread') and hasattr(_c, 'imwrite') and hasattr(_c, 'IMWRITE_JPEG_QUALITY')\nprint('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


@pytest.fixture(scope="module")
def shims():
    import ffp_amd  # noqa: F401
    from ffp_amd import compat
    compat.install()
    return compat


def fake_results(rng, n, w, h):
    from utils.yolo_wrapper import Results
    rows = np.zeros((n, 21), np.float32)
    x1 = rng.uniform(0, w - 30, n); y1 = rng.uniform(0, h - 30, n)
    rows[:, 0], rows[:, 1] = x1, y1
    rows[:, 2], rows[:, 3] = x1 + rng.uniform(8, 29.9, n), y1 + rng.uniform(8, 29.9, n)
    rows[:, 4] = rng.uniform(0.05, 0.99, n)
    rows[:, 6:] = rng.uniform(0, 100, (n, 15))
    return Results(rows, 5, (h, w), {0: "face"}), rows


def test_adapter_conversion_and_keypoint_attach_match_oracle(shims):
    """a5 + a8: truncation, shift, no second confidence filter, cache keys, exact-key and best-IoU (> 0.5, first wins) attach."""
    from oracle import wrapper_ref
    from sahi.prediction import ObjectPrediction
    from utils.yolo_wrapper import YOLOv11PoseDetectionModel
    rng = np.random.default_rng(11)
    m = YOLOv11PoseDetectionModel(model_path="unused", confidence_threshold=0.5, device="cuda:0", image_size=256, load_at_init=False)
    cache = {}
    all_preds, all_ref = [], []
    for shift in ([0, 0], [410, 0], [820, 410], [3328, 1648]):
        res, rows = fake_results(rng, 7, 512, 512)
        m._original_predictions = [res]
        m.convert_original_predictions(shift_amount=shift, full_shape=[2160, 3840])       # SAHI passes flat lists (docs sahi/base.py:176-179)
        got = m.object_prediction_list
        ref = wrapper_ref.convert(rows[:, :4], rows[:, 4], rows[:, 6:].reshape(-1, 5, 3), shift, [2160, 3840], cache)
        assert len(got) == len(ref) == 7                                                     # scores below the threshold are NOT dropped here
        for p, (bx, sc, sh) in zip(got, ref):
            assert p.bbox.to_xyxy() == bx and p.score.value == pytest.approx(sc) and p.bbox.shift_amount == sh
            assert p.category.id == 0 and p.category.name == "face"
        all_preds += [p.get_shifted_object_prediction() for p in got]
        all_ref += [[b[0] + s[0], b[1] + s[1], b[2] + s[0], b[3] + s[1]] for b, _, s in ref]
    assert set(m.keypoints_cache) == set(cache)
    for k in cache:
        assert np.array_equal(m.keypoints_cache[k], cache[k])
    # nested-list form of the arguments (utils/yolo_wrapper.py:99-118) gives the same list
    m2 = YOLOv11PoseDetectionModel(model_path="unused", load_at_init=False)
    res, rows = fake_results(rng, 3, 512, 512)
    m2._original_predictions = [res]
    m2._create_object_prediction_list_from_original_predictions([[5, 6]], [[2160, 3840]])
    a = [p.bbox.to_xyxy() for p in m2.object_prediction_list]
    m2._original_predictions = [res]
    m2._create_object_prediction_list_from_original_predictions([5, 6], [2160, 3840])
    assert a == [p.bbox.to_xyxy() for p in m2.object_prediction_list]
    # attach: exact boxes, grown (merged) boxes that still overlap > 0.5, boxes that overlap nothing
    boxes = [p.bbox.to_voc_bbox() for p in all_preds]
    assert boxes == all_ref
    grown = [[b[0] - 2, b[1] - 1, b[2] + 3, b[3] + 2] for b in boxes[::3]]
    far = [[3000, 2000, 3040, 2050], [1, 1, 9, 9]]
    query = boxes + [[max(v, 0) for v in g] for g in grown] + far
    preds = [ObjectPrediction(bbox=q, category_id=0, category_name="face", score=0.9) for q in query]
    out = m.attach_keypoints_to_predictions(preds)
    ref = wrapper_ref.attach(query, cache)
    assert out is preds
    n_att = 0
    for p, r in zip(out, ref):
        if r is None:
            assert not hasattr(p, "keypoints")
        else:
            assert p.keypoints.shape == (5, 3) and np.array_equal(p.keypoints, r)
            n_att += 1
    assert n_att >= len(boxes) + len(grown) - 2 and not hasattr(out[-1], "keypoints")
    for q in (query[0], query[len(boxes)], far[0]):                                         # the lookup app_yolo_sahi.py:80-84 calls
        r = wrapper_ref.attach([q], cache)[0]
        k = m.get_keypoints_for_bbox(q)
        assert (k is None) == (r is None) and (k is None or np.array_equal(k, r))
    # empty result convention (:95-97) and cache reset on unload (:58-61)
    m._original_predictions = [fake_results(rng, 0, 64, 64)[0]]
    m.convert_original_predictions(shift_amount=[0, 0], full_shape=[64, 64])
    assert m.object_prediction_list == []
    m.unload_model()
    assert m.keypoints_cache == {} and m.model is None


def test_presentation_helpers_write_the_reference_outputs(shims, tmp_path):
    from PIL import Image
    from sahi.prediction import ObjectPrediction, PredictionResult
    from utils.visualization import (FACE_KEYPOINT_NAMES, create_detection_summary, draw_detections, draw_detections_on_image,
                                     save_face_crops)
    import cv2
    img = (np.random.default_rng(3).integers(0, 255, (120, 160, 3))).astype(np.uint8)
    path = str(tmp_path / "in.png")
    assert cv2.imwrite(path, img) and np.array_equal(cv2.imread(path), img)                 # shim round trip, BGR in / BGR out
    assert cv2.imread(str(tmp_path / "missing.png")) is None
    preds = [ObjectPrediction(bbox=[10, 20, 50, 70], category_id=0, category_name="face", score=0.876),
             ObjectPrediction(bbox=[140, 100, 200, 150], category_id=0, category_name="face", score=0.512),     # clamped to 160 x 120
             ObjectPrediction(bbox=[30, 30, 30, 60], category_id=0, category_name="face", score=0.7)]           # empty: skipped
    preds[0].keypoints = np.asarray([[20, 30, 0.9], [40, 30, 0.8], [30, 45, 0.7], [22, 60, 0.2], [38, 60, 0.95]], np.float32)
    res = PredictionResult(object_prediction_list=preds, image=path, durations_in_seconds={})
    crops = save_face_crops(path, res, str(tmp_path / "crop"), prefix="t_face")
    assert [os.path.basename(c) for c in crops] == ["t_face_1_conf_0.88.jpg", "t_face_2_conf_0.51.jpg"]
    assert Image.open(crops[0]).size == (40, 50) and Image.open(crops[1]).size == (20, 20)
    out = str(tmp_path / "viz" / "o.jpg")
    draw_detections(path, res, out, show_confidence=True, show_keypoints=True, box_color=(0, 255, 0), text_color=(255, 255, 255),
                    kpt_conf_threshold=0.3)
    assert Image.open(out).size == (160, 120)
    over = draw_detections_on_image(img, res, draw_skeleton=True)
    assert over.shape == img.shape and (over != img).any() and (over[25:70, 10] == (0, 255, 0)).all(1).sum() >= 40      # left edge of box 1, BGR green
    s = str(tmp_path / "sum" / "s.txt")
    create_detection_summary(res, path, 1.234, s, 160, 120, 640, 640)
    text = open(s, encoding="utf-8").read().splitlines()
    assert text[1] == "=== Ringkasan Deteksi Wajah dengan Keypoints ===" and "Gambar Sumber: in.png" in text
    assert "Ukuran Gambar Asli: 160x120 px" in text and "Ukuran Slice: 640x640 px" in text and "Waktu Proses Total: 1.23 detik" in text
    assert "Total Wajah Ditemukan: 3" in text and "Rata-rata Skor Kepercayaan: 0.696" in text
    assert "  - Bounding Box: [x1: 10, y1: 20, x2: 50, y2: 70]" in text and "  - Skor Kepercayaan: 0.876" in text
    assert f"      {FACE_KEYPOINT_NAMES[0]}: (20.0, 30.0) [conf: 0.900]" in text
    empty = PredictionResult(object_prediction_list=[], image=path, durations_in_seconds={})
    create_detection_summary(empty, path, 0.5, s, 160, 120, 640, 640)
    assert "Tidak ada wajah yang terdeteksi." in open(s, encoding="utf-8").read()


def test_cv2_shim_imread_applies_exif_orientation(shims, tmp_path):
    """`cv2.imread` rotates / mirrors by the EXIF orientation tag unless told not to; the shim does the same on both of its paths
    (phone photographs came back in sensor orientation)."""
    from PIL import Image
    import cv2
    if not hasattr(cv2, "_oriented"):
        pytest.skip("a real OpenCV is installed")
    rng = np.random.default_rng(11)
    rgb = rng.integers(0, 255, (24, 40, 3)).astype(np.uint8)
    for o in range(1, 9):
        p = str(tmp_path / f"o{o}.png")
        ex = Image.Exif()
        ex[0x0112] = o
        Image.fromarray(rgb).save(p, exif=ex)
        from PIL import ImageOps
        want = np.asarray(ImageOps.exif_transpose(Image.open(p)).convert("RGB"))[..., ::-1]
        got = cv2.imread(p)
        assert got.shape == want.shape and np.array_equal(got, want), o
        assert cv2.imread(p, cv2.IMREAD_UNCHANGED).shape == rgb.shape


def test_official_evaluator_hands_the_detector_what_the_reference_does(shims, tmp_path, monkeypatch):
    """eval/eval_official_widerface.py:166-243: the image is read with cv2.imread (BGR) and, without SAHI, goes to the YOLO object
    itself — `self.detection_model.model(img, conf=0.01, verbose=False)`, Ultralytics' default imgsz — whose FLOAT xyxy boxes become
    top-left x, y, w, h; with SAHI it goes to get_sliced_prediction as the same BGR array."""
    from PIL import Image
    from eval.eval_official_widerface import OfficialWiderFaceEvaluator
    import torch
    rgb = np.zeros((48, 64, 3), np.uint8)
    rgb[..., 0] = 200                                                  # a red picture: BGR has it in channel 2
    p = str(tmp_path / "red.png")
    Image.fromarray(rgb).save(p)
    ev = OfficialWiderFaceEvaluator.__new__(OfficialWiderFaceEvaluator)
    ev.use_sahi, ev.inference_confidence = False, 0.01
    ev.sahi_config = {"slice_height": 640, "slice_width": 640, "overlap_ratio": 0.2}
    seen = {}

    class _B:
        xyxy = torch.tensor([[1.25, 2.5, 11.75, 22.0]])
        conf = torch.tensor([0.625])
        def __len__(self):
            return 1

    class _R:
        boxes = _B()

    class _Yolo:
        def __call__(self, img, **kw):
            seen["img"], seen["kw"] = img, kw
            return [_R()]

    class _DM:
        model = _Yolo()

    ev.detection_model = _DM()
    out = ev._run_single_inference(p)
    assert seen["img"].shape == (48, 64, 3) and seen["img"][0, 0].tolist() == [0, 0, 200]          # BGR
    assert seen["kw"] == {"conf": 0.01, "verbose": False}                                           # no imgsz: the YOLO default
    assert out.dtype == np.float64 and np.allclose(out, [[1.25, 2.5, 10.5, 19.5, 0.625]])
    assert ev._run_single_inference(str(tmp_path / "missing.jpg")).size == 0
    # SAHI branch: same array, the reference's fixed arguments
    import sahi.predict as sp
    def fake_gsp(image, model, **kw):
        seen["sahi_img"], seen["sahi_kw"] = image, kw
        class _Res:
            object_prediction_list = []
        return _Res()
    monkeypatch.setattr(sp, "get_sliced_prediction", fake_gsp)
    ev.use_sahi = True
    assert ev._run_single_inference(p).size == 0
    assert seen["sahi_img"][0, 0].tolist() == [0, 0, 200]
    assert seen["sahi_kw"] == dict(slice_height=640, slice_width=640, overlap_height_ratio=0.2, overlap_width_ratio=0.2, postprocess_type="NMS",
                                   postprocess_match_threshold=0.5, postprocess_class_agnostic=True, verbose=0)


def test_evaluator_enhancement_modes_plumbing(shims, tmp_path, monkeypatch):
    """eval/eval_official_widerface.py:166-255 and eval/eval_dual.py:182-270, the four pipelines: the enhancement phase (always, or when
    the quick analysis at conf 0.05 finds no / mostly small faces), detection on the enhanced picture, boxes divided by the enhancer's
    scale; mode strings (' -> ' in the official evaluator, ' + ' in the dual one); statistics and the per-path cache of the dual one."""
    from PIL import Image
    import torch
    from eval.eval_dual import DualWiderFaceEvaluator
    from eval.eval_official_widerface import OfficialWiderFaceEvaluator
    p = str(tmp_path / "pic.png")
    Image.fromarray(np.full((40, 60, 3), 90, np.uint8)).save(p)
    calls = []

    class _B:
        def __init__(self, rows):
            self.xyxy = torch.tensor(rows, dtype=torch.float32).reshape(-1, 5)[:, :4]
            self.conf = torch.tensor(rows, dtype=torch.float32).reshape(-1, 5)[:, 4]
        def __len__(self):
            return self.xyxy.shape[0]

    class _Yolo:
        faces = [[10.0, 10.0, 30.0, 40.0, 0.9]]                        # what the quick analysis sees
        def __call__(self, img, **kw):
            calls.append((img.shape, kw["conf"]))
            if kw["conf"] == 0.05:
                rows = self.faces
            else:                                                      # the real pass: one box that scales with the picture
                rows = [[img.shape[1] / 4, img.shape[0] / 4, img.shape[1] / 2, img.shape[0] / 2, 0.75]]
            class _R:
                boxes = _B(rows)
            return [_R()]

    class _DM:
        model = _Yolo()

    class _Enh:
        scale = 2
        n = 0
        def enhance_image(self, img):
            _Enh.n += 1
            return np.repeat(np.repeat(img, 2, 0), 2, 1), True

    def official(**kw):
        ev = OfficialWiderFaceEvaluator.__new__(OfficialWiderFaceEvaluator)
        ev.use_sahi, ev.slicing_strategy, ev.inference_confidence, ev.face_size_threshold = False, "uniform", 0.01, 50
        ev.use_enhancer, ev.bounded_enhancement = kw.get("enh", False), kw.get("bounded", False)
        ev.sahi_config = {"slice_height": 640, "slice_width": 640, "overlap_ratio": 0.2}
        ev.detection_model, ev.face_enhancer = _DM(), (_Enh() if ev.use_enhancer else None)
        ev._build_mode_string()
        return ev

    base = official()._run_single_inference(p)
    assert np.allclose(base, [[15, 10, 15, 10, 0.75]]) and official().mode_string == "BASELINE"
    ev = official(enh=True)
    n0 = _Enh.n
    out = ev._run_single_inference(p)
    assert _Enh.n == n0 + 1 and calls[-1] == ((80, 120, 3), 0.01)      # detection ran on the x2 picture ...
    assert np.allclose(out, base) and ev.mode_string == "FULL-ENHANCE -> BASELINE"          # ... and its boxes came back to the original's coordinates
    ev = official(enh=True, bounded=True)
    assert ev.mode_string == "BOUNDED-ENHANCE (<50px) -> BASELINE"
    _Yolo.faces = [[0.0, 0.0, 60.0, 70.0, 0.9], [0.0, 0.0, 80.0, 55.0, 0.8]]       # large faces: no enhancement
    n0 = _Enh.n
    out = ev._run_single_inference(p)
    assert _Enh.n == n0 and calls[-2:] == [((40, 60, 3), 0.05), ((40, 60, 3), 0.01)] and np.allclose(out, base)
    _Yolo.faces = [[0.0, 0.0, 20.0, 30.0, 0.9], [0.0, 0.0, 80.0, 55.0, 0.8], [0.0, 0.0, 10.0, 12.0, 0.8]]     # two of three are small
    out = ev._run_single_inference(p)
    assert _Enh.n == n0 + 1 and np.allclose(out, base)
    assert ev._quick_face_analysis(None)[0] is False
    _Yolo.faces = []
    assert ev._quick_face_analysis(np.zeros((8, 8, 3), np.uint8))[:2] == (True, "No faces detected")
    official_sahi = official(enh=True)
    official_sahi.use_sahi = True
    official_sahi._build_mode_string()
    assert official_sahi.mode_string == "FULL-ENHANCE -> SAHI (uniform)"

    # the dual evaluator: same phases, dict rows, statistics, cache, ' + '
    _Yolo.faces = [[0.0, 0.0, 20.0, 30.0, 0.9]]
    d = DualWiderFaceEvaluator(subcategory_gt={}, use_enhancer=True, bounded_enhancement=True)
    assert d.mode_string == "BOUNDED-ENHANCE (<50px) + BASELINE" and d.inference_confidence == 0.5
    d.detection_model, d.face_enhancer = _DM(), _Enh()
    rows = d.run_inference(p)
    assert len(rows) == 1 and np.allclose(rows[0]["bbox"], [15, 10, 15, 10]) and abs(rows[0]["confidence"] - 0.75) < 1e-6
    assert d.enhancement_stats["enhanced_images"] == 1 and d.run_inference(p) is rows            # cached
    _Yolo.faces = [[0.0, 0.0, 90.0, 90.0, 0.9]]
    p2 = str(tmp_path / "pic2.png")
    Image.fromarray(np.full((40, 60, 3), 30, np.uint8)).save(p2)
    d.run_inference(p2)
    assert d.enhancement_stats == {"total_images": 0, "enhanced_images": 1, "skipped_images": 1}
    assert d.run_inference(str(tmp_path / "missing.jpg")) == []
    ds = DualWiderFaceEvaluator(subcategory_gt={}, use_sahi=True, use_enhancer=True, slicing_strategy="adaptive", sahi_match_thresholds=[0.4])
    assert ds.mode_string == "FULL-ENHANCE + SAHI (adaptive)" and ds.inference_confidence == 0.01
    assert ds.sahi_config["postprocess_match_thresholds"] == [0.4] and ds.sahi_config["overlap_ratio"] == 0.2 and "slice_height" not in ds.sahi_config
    assert DualWiderFaceEvaluator(subcategory_gt={}).sahi_config["overlap_ratio"] == 0.25
    assert [ds.get_slice_size_adaptive(w, 10) for w in (2501, 2500, 1501, 1500)] == [512, 416, 416, 320]
    seen = {}
    import sahi.predict as sp
    def fake_gsp(image, model, **kw):
        seen["shape"], seen["kw"] = image.shape, kw
        class _Res:
            object_prediction_list = []
        return _Res()
    monkeypatch.setattr(sp, "get_sliced_prediction", fake_gsp)
    ds.detection_model, ds.face_enhancer = _DM(), _Enh()
    assert ds.run_inference(p) == []
    assert seen["shape"] == (80, 120, 3) and seen["kw"]["slice_height"] == 320 and seen["kw"]["postprocess_match_threshold"] == 0.4
    assert seen["kw"]["postprocess_type"] == "NMS" and seen["kw"]["postprocess_match_metric"] == "IOS" and seen["kw"]["postprocess_class_agnostic"] is True
