"""End-to-end replays of the reference's two named callers on the GPU, through the import-compatible shims only (restated call
sequences — the reference's files are not read or executed):
  * pipeline_v4_yolo/app_yolo_sahi.py:19-119 `process_single_image`: cv2.imread -> get_sliced_prediction(path) -> get_keypoints_for_bbox
    per detection -> draw_detections -> save_face_crops -> create_detection_summary;
  * pipeline_v1_detection_first/app_v1.py:44-104: adaptive slice size -> get_sliced_prediction -> draw_detections -> save_face_crops ->
    create_detection_summary -> FaceEnhancer(x4, tile 400, half) -> enhance_face_crops_batch -> create_enhancement_summary
    (the YOLO wrapper stands in for the out-of-scope InsightFace detector of that script);
  * the enhance-first ordering at the reference's settings (pipeline_v4_yolo/app_yolo_full.py:87-123): a 23-block x2 model, tile 400 /
    pad 10, on a picture larger than two tiles, then sliced detection on the enhanced picture.
JPEG files in, JPEG files out; boxes, crops and enhanced crops against the oracle, file names and report text against the reference's
conventions (the text functions themselves are pinned byte for byte by tests/test_wrapper_pinned.py)."""
import os
import re

import numpy as np
import pytest

from util import match_by_iou, psnr_u8

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def compat(gpu_lib):
    from ffp_amd import compat as c
    c.install()
    return c


def _jpeg_frame(path, h, w, seed):
    from PIL import Image
    from ffp_amd import synth
    Image.fromarray(synth.synthetic_frame(h, w, seed=seed)).save(path, quality=95)
    return np.asarray(Image.open(path).convert("RGB"))              # what every reader of the file sees (libjpeg-turbo)


def _oracle_boxes(rgb, imgsz, sh, sw, conf=0.5):
    from ffp_amd import synth
    from oracle import sahi_ref, ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    ref = Yolo11PoseRef(synth.yolo11_pose_weights("n"), "n")
    bgr = np.ascontiguousarray(rgb[..., ::-1])                        # the wrapper hands Ultralytics a BGR array (cv2.imread / read_image_as_pil -> BGR flip)
    exp = sahi_ref.get_sliced_prediction(bgr[..., ::-1], lambda im: ultra_post.predict(ref, im, imgsz, conf, 0.7, 300), sh, sw, 0.2, 0.2)
    return np.asarray([d.bbox for d in exp], np.float32).reshape(-1, 4)


def test_process_single_image_sequence(compat, tmp_path):
    import cv2
    from PIL import Image
    from sahi.predict import get_sliced_prediction
    from utils.visualization import create_detection_summary, draw_detections, save_face_crops
    from utils.yolo_wrapper import YOLOv11PoseDetectionModel
    image_path = str(tmp_path / "input" / "crowd_01.jpg")
    os.makedirs(os.path.dirname(image_path))
    rgb = _jpeg_frame(image_path, 700, 1100, seed=5)
    config = {"slice_height": 320, "slice_width": 320, "overlap_ratio": 0.2, "confidence_threshold": 0.5, "device": "cuda:0",
              "show_keypoints": True, "show_confidence": True, "kpt_conf_threshold": 0.3}
    detection_model = YOLOv11PoseDetectionModel(model_path="synthetic:yolo11n-pose", confidence_threshold=config["confidence_threshold"],
                                                device=config["device"], image_size=512, load_at_init=True)
    # ---- process_single_image(image_path, detection_model, output_base_dir, config) ----
    base_name = "crowd_01"
    output_folder = str(tmp_path / "output" / base_name)
    crops_folder = os.path.join(output_folder, "crop")
    os.makedirs(crops_folder)
    image = cv2.imread(image_path)
    assert image is not None and np.array_equal(image, rgb[..., ::-1])          # shim: BGR, pixel-identical to libjpeg-turbo
    img_height, img_width = image.shape[:2]
    result = get_sliced_prediction(image_path, detection_model, slice_height=config["slice_height"], slice_width=config["slice_width"],
                                   overlap_height_ratio=config["overlap_ratio"], overlap_width_ratio=config["overlap_ratio"])
    num = len(result.object_prediction_list)
    assert num > 0
    for detection in result.object_prediction_list:
        keypoints = detection_model.get_keypoints_for_bbox(detection.bbox.to_xyxy())
        if keypoints is not None:
            detection.keypoints = keypoints
    with_kpts = [d for d in result.object_prediction_list if hasattr(d, "keypoints")]      # exact key, else best IoU > 0.5 (utils/yolo_wrapper.py:168-200): merged unions may find none
    assert len(with_kpts) >= num // 2 and all(np.asarray(d.keypoints).shape == (5, 3) for d in with_kpts)
    viz = os.path.join(output_folder, f"{base_name}_with_keypoints.jpg")
    draw_detections(image_path, result, viz, show_confidence=True, show_keypoints=True, box_color=(0, 255, 0), text_color=(255, 255, 255),
                    kpt_conf_threshold=config["kpt_conf_threshold"])
    saved = save_face_crops(image_path, result, crops_folder, prefix=f"{base_name}_face")
    summary = os.path.join(output_folder, f"{base_name}_summary.txt")
    create_detection_summary(result, image_path, 1.234, summary, img_width, img_height, config["slice_width"], config["slice_height"])
    # ---- against the oracle on the decoded file ----
    got = np.asarray([p.bbox.to_xyxy() for p in result.object_prediction_list], np.float32).reshape(-1, 4)
    eb = _oracle_boxes(rgb, 512, 320, 320)
    assert abs(len(got) - len(eb)) <= 1 and len(eb) > 0
    assert np.mean([np.array_equal(eb[i], got[j]) for i, j, _ in match_by_iou(eb, got)]) >= 0.95
    # drawn picture: same size, differs from the input where boxes are
    assert Image.open(viz).size == (img_width, img_height)
    assert (np.asarray(Image.open(viz).convert("RGB")).astype(int) - rgb).any()
    # crops: reference file names (utils/visualization.py:215), and the bytes cv2.imwrite would leave for that crop
    assert len(saved) == num and all(os.path.dirname(p) == crops_folder for p in saved)
    for k, (p, det) in enumerate(zip(saved, result.object_prediction_list)):
        assert os.path.basename(p) == f"{base_name}_face_{k + 1}_conf_{det.score.value:.2f}.jpg"
        x1, y1, x2, y2 = [int(c) for c in det.bbox.to_xyxy()]
        x1, y1, x2, y2 = max(0, x1), max(0, y1), min(img_width, x2), min(img_height, y2)
        want = tmp_path / "want.jpg"
        Image.fromarray(rgb[y1:y2, x1:x2]).save(want, quality=95)
        assert open(p, "rb").read() == want.read_bytes(), f"crop {k} differs from libjpeg-turbo's file"
    text = open(summary, encoding="utf-8").read()
    assert f"Gambar Sumber: {os.path.basename(image_path)}" in text and f"Ukuran Gambar Asli: {img_width}x{img_height} px" in text
    assert f"Ukuran Slice: 320x320 px" in text and "Waktu Proses Total: 1.23 detik" in text and f"Total Wajah Ditemukan: {num}" in text
    assert len(re.findall(r"^Wajah #\d+:$", text, re.M)) == num
    for k, det in enumerate(result.object_prediction_list):
        x1, y1, x2, y2 = [int(c) for c in det.bbox.to_xyxy()]
        assert f"Wajah #{k + 1}:\n  - Bounding Box: [x1: {x1}, y1: {y1}, x2: {x2}, y2: {y2}]\n  - Skor Kepercayaan: {det.score.value:.3f}" in text


def test_detect_then_enhance_crops_sequence(compat, tmp_path):
    from PIL import Image
    from ffp_amd import synth
    from oracle import rrdbnet_ref
    from sahi.predict import get_sliced_prediction
    from utils.enhancer import FaceEnhancer, create_enhancement_summary, enhance_face_crops_batch
    from utils.visualization import create_detection_summary, draw_detections, save_face_crops
    from utils.yolo_wrapper import YOLOv11PoseDetectionModel
    source_dir, output_dir = tmp_path / "data" / "input", tmp_path / "data" / "output"
    os.makedirs(source_dir)
    os.makedirs(output_dir)
    test_image_name = "13_Interview On-Location 13.jpg"                        # the clean_name rule replaces blanks
    test_image_path = str(source_dir / test_image_name)
    rgb = _jpeg_frame(test_image_path, 600, 760, seed=33)
    detection_model = YOLOv11PoseDetectionModel(model_path="synthetic:yolo11n-pose", confidence_threshold=0.5, device="cuda:0", image_size=512, load_at_init=True)
    clean_name = re.sub(r"[^a-zA-Z0-9_-]", "_", os.path.splitext(test_image_name)[0])
    assert clean_name == "13_Interview_On-Location_13"
    result_dir = str(output_dir / f"result_{clean_name}")
    os.makedirs(result_dir)
    crops_dir = os.path.join(result_dir, "face_crops")
    visual_output_path = os.path.join(result_dir, f"visual_{clean_name}.png")
    base_slice_size = 512
    with Image.open(test_image_path) as img:
        img_width, img_height = img.size
    slice_height = img_height // 2 if img_height < base_slice_size * 1.5 else base_slice_size
    slice_width = img_width // 2 if img_width < base_slice_size * 1.5 else base_slice_size
    assert (slice_height, slice_width) == (300, 380)
    detection_result = get_sliced_prediction(image=test_image_path, detection_model=detection_model, slice_height=slice_height, slice_width=slice_width,
                                             overlap_height_ratio=0.2, overlap_width_ratio=0.2)
    preds = detection_result.object_prediction_list
    assert len(preds) > 0
    draw_detections(test_image_path, detection_result, visual_output_path)
    saved_crops = save_face_crops(test_image_path, detection_result, crops_dir, prefix=clean_name)
    create_detection_summary(result=detection_result, image_path=test_image_path, processing_time=0.5, output_path=os.path.join(result_dir, "detection_summary.txt"),
                             img_width=img_width, img_height=img_height, slice_width=slice_width, slice_height=slice_height)
    assert Image.open(visual_output_path).size == (img_width, img_height) and len(saved_crops) == len(preds)
    got = np.asarray([p.bbox.to_xyxy() for p in preds], np.float32).reshape(-1, 4)
    eb = _oracle_boxes(rgb, 512, slice_height, slice_width)
    assert abs(len(got) - len(eb)) <= 1 and np.mean([np.array_equal(eb[i], got[j]) for i, j, _ in match_by_iou(eb, got)]) >= 0.95
    enhancer = FaceEnhancer(model_name="RealESRGAN_x4plus", model_path="synthetic:RealESRGAN_x4plus", scale=4, tile=400, half=True)
    enhancement_results = enhance_face_crops_batch(crops_dir=crops_dir, enhancer=enhancer, prefix=clean_name)
    summary_path = os.path.join(result_dir, "enhancement_summary.txt")
    create_enhancement_summary(enhancement_results, summary_path)
    stats = enhancement_results["statistics"]
    big_enough = [p for p in saved_crops if min(Image.open(p).size) >= 4]
    assert stats["total_files"] == len(saved_crops) and stats["successful"] == len(big_enough) and stats["failed"] == len(saved_crops) - len(big_enough)
    enh_dir = os.path.join(result_dir, f"{clean_name}_enhanced")                # utils/enhancer.py:344-350: next to the crops, "<prefix>_enhanced"
    net = rrdbnet_ref.RRDBNetRef(synth.rrdbnet_weights(4, 23), 4, 23)
    checked = 0
    for info in enhancement_results["enhancement_info"]:
        src, dst = info["original_path"], info["output_path"]
        assert os.path.dirname(dst) == enh_dir and os.path.basename(dst) == f"{clean_name}_{os.path.basename(src)}"
        crop_bgr = np.asarray(Image.open(src).convert("RGB"))[..., ::-1].copy()   # the decoded crop FILE is what the reference enhances
        assert info["original_size"] == (crop_bgr.shape[1], crop_bgr.shape[0]) and info["enhanced_size"] == (4 * crop_bgr.shape[1], 4 * crop_bgr.shape[0])
        if checked < 3:                                                              # CPU oracle: three crops are enough
            want = rrdbnet_ref.enhance(net, crop_bgr)
            ref_file = tmp_path / "ref.jpg"
            Image.fromarray(np.ascontiguousarray(want[..., ::-1])).save(ref_file, quality=95)
            got_px = np.asarray(Image.open(dst).convert("RGB"))
            assert psnr_u8(got_px, np.asarray(Image.open(ref_file).convert("RGB"))) >= 45.0     # fp16 network (>= 50 dB before JPEG) through the same quality-95 file format
            checked += 1
    assert checked >= 1
    text = open(summary_path, encoding="utf-8").read()
    assert "=== LAPORAN ENHANCEMENT WAJAH ===" in text and f"Total File Diproses: {stats['total_files']}" in text and f" Berhasil: {stats['successful']}" in text


def test_enhance_first_at_reference_settings(gpu_lib):
    """23-block RealESRGAN_x2plus, tile 400 / pad 10 (utils/enhancer.py:21,135-142) on a picture that needs 3 x 3 tiles, then sliced
    detection on the enhanced picture: SR <= 1 LSB in fp32 against the oracle's tiled enhance, detections against the oracle run on the
    oracle's enhanced picture."""
    import torch
    from ffp_amd import pipeline, synth
    from oracle import rrdbnet_ref, sahi_ref, ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    H, W = 820, 900
    frame = synth.synthetic_frame(H, W, seed=5, n_blobs=60)
    W2 = synth.rrdbnet_weights(2, 23)
    Wd = synth.yolo11_pose_weights("n")
    cfg = pipeline.PipeConfig(slice_h=512, slice_w=512, overlap=0.2, imgsz=512, conf=0.01, sr_crops=0)      # random-init SR output is low-contrast: a low threshold gives ~250 boxes to compare
    pipe = pipeline.FramePipeline(Wd, None, cfg, arch="n", device=0, det_precision=gpu_lib.PREC_F32X3)
    x2 = gpu_lib.Enhancer(W2, 2, 23, device=0, half=False)
    d_frame = torch.from_numpy(frame).cuda()
    enh, rows, n = pipe.enhance_first(d_frame, H, W, enhancer=x2, tile=400, tile_pad=10)
    got_sr = enh.cpu().numpy()
    want_sr = rrdbnet_ref.enhance(rrdbnet_ref.RRDBNetRef(W2, 2, 23), frame, tile=400, tile_pad=10)
    assert got_sr.shape == want_sr.shape == (2 * H, 2 * W, 3)
    diff = np.abs(got_sr.astype(int) - want_sr.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02, (int(diff.max()), float((diff > 0).mean()))
    k = pipe.merged_count(n)
    got = rows[:k, :4].cpu().numpy()
    ref = Yolo11PoseRef(Wd, "n")
    # the detections against the oracle on THAT enhanced picture (a 1-LSB input difference could move an int-truncated box edge)
    exp = sahi_ref.get_sliced_prediction(got_sr, lambda im: ultra_post.predict(ref, im, 512, 0.01, 0.7, 300), 512, 512, 0.2, 0.2, True,
                                         "GREEDYNMM", "IOS", 0.5, False)
    eb = np.asarray([d.bbox for d in exp], np.float32).reshape(-1, 4)
    assert eb.shape[0] > 0 and abs(k - eb.shape[0]) <= max(1, eb.shape[0] // 50), (k, eb.shape[0])
    ious = np.array([x[2] for x in match_by_iou(eb, got)])
    assert (ious >= 0.999).mean() >= 0.95, ious


def test_evaluator_full_pipeline_mode_through_the_shims(compat, tmp_path):
    """The evaluators' fourth pipeline, 'Enhance + SAHI + YOLO' (eval/eval_dual.py:24-31,185-265; eval_official_widerface.py:166-255):
    JPEG file -> cv2.imread -> FaceEnhancer('RealESRGAN_x2plus').enhance_image (23 blocks, tile 400, fp16) -> get_sliced_prediction on the
    2x picture (NMS / IOS / 0.5, class-agnostic, 640 slices) -> boxes / 2. Checked against the same steps made by hand through the
    library, and against the oracle's detector run on the enhanced picture the device produced."""
    import cv2
    from eval.eval_dual import DualWiderFaceEvaluator
    from eval.eval_official_widerface import OfficialWiderFaceEvaluator
    from sahi.predict import get_sliced_prediction
    from utils.enhancer import FaceEnhancer
    from utils.yolo_wrapper import YOLOv11PoseDetectionModel
    from oracle import sahi_ref, ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    from ffp_amd import synth
    p = str(tmp_path / "0_Parade_marchingband_1_5.jpg")
    _jpeg_frame(p, 360, 480, seed=5)
    model = YOLOv11PoseDetectionModel(model_path="synthetic:yolo11n-pose", confidence_threshold=0.01, device="cuda:0", load_at_init=True)
    enhancer = FaceEnhancer(model_name="RealESRGAN_x2plus", model_path="synthetic:RealESRGAN_x2plus")
    assert enhancer.scale == 2

    d = DualWiderFaceEvaluator(subcategory_gt={}, use_sahi=True, use_enhancer=True)
    d.detection_model, d.face_enhancer = model, enhancer
    assert d.mode_string == "FULL-ENHANCE + SAHI (uniform)"
    rows = d.run_inference(p)
    assert d.enhancement_stats["enhanced_images"] == 1 and len(rows) > 0

    o = OfficialWiderFaceEvaluator.__new__(OfficialWiderFaceEvaluator)
    o.use_sahi, o.slicing_strategy, o.inference_confidence, o.face_size_threshold = True, "uniform", 0.01, 50
    o.use_enhancer, o.bounded_enhancement = True, False
    o.sahi_config = {"slice_height": 640, "slice_width": 640, "overlap_ratio": 0.2}
    o.detection_model, o.face_enhancer = model, enhancer
    o._build_mode_string()
    assert o.mode_string == "FULL-ENHANCE -> SAHI (uniform)"
    out = o._run_single_inference(p)

    # by hand, the same library calls
    img = cv2.imread(p)
    big, ok = enhancer.enhance_image(img)
    assert ok and big.shape == (720, 960, 3)
    for overlap, got in ((0.25, np.asarray([[*r["bbox"], r["confidence"]] for r in rows], np.float64)), (0.2, out)):
        res = get_sliced_prediction(big, model, slice_height=640, slice_width=640, overlap_height_ratio=overlap, overlap_width_ratio=overlap,
                                    postprocess_type="NMS", postprocess_match_metric="IOS", postprocess_match_threshold=0.5, postprocess_class_agnostic=True, verbose=0)
        want = np.asarray([[*q.bbox.to_xywh(), q.score.value] for q in res.object_prediction_list], np.float64).reshape(-1, 5)
        want[:, :4] /= 2
        assert got.shape == want.shape and np.array_equal(got, want)
    # the oracle's detector + SAHI on that enhanced picture. The evaluator hands sahi the BGR array cv2 produced; sahi takes arrays for RGB
    # (read_image_as_pil), the wrapper flips to BGR, Ultralytics flips back: the network sees the array AS IT IS (R and B swapped w.r.t.
    # the file — the reference's behaviour, kept)
    ref = Yolo11PoseRef(synth.yolo11_pose_weights("n"), "n")
    exp = sahi_ref.get_sliced_prediction(np.ascontiguousarray(big), lambda im: ultra_post.predict(ref, im, 1024, 0.01, 0.7, 300), 640, 640, 0.2, 0.2, True,
                                         "NMS", "IOS", 0.5, True)                # image_size 1024: the wrapper's default (utils/yolo_wrapper.py:13)
    eb = np.asarray([q.bbox for q in exp], np.float32).reshape(-1, 4)
    gb = out[:, :4].copy() * 2
    gb[:, 2:] += gb[:, :2]
    assert eb.shape[0] > 0 and abs(gb.shape[0] - eb.shape[0]) <= max(1, eb.shape[0] // 25), (gb.shape, eb.shape)
    ious = np.array([x[2] for x in match_by_iou(eb, gb.astype(np.float32))])
    assert (ious >= 0.999).mean() >= 0.95, ious
