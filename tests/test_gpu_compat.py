"""GPU test of the drop-in surface: the reference's own import lines and call sequence
(pipeline_v4_yolo/app_yolo_sahi.py:9-17,50-57; pipeline_v1_detection_first/app_v1.py:91-104) run against this build,
and agree with the oracle."""
import os

import numpy as np
import pytest

from util import match_by_iou, psnr_u8

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def compat(gpu_lib):
    from ffp_amd import compat as c
    c.install()
    return c


def test_reference_import_lines_and_sliced_call(compat, tmp_path):
    from sahi.predict import get_sliced_prediction, get_prediction
    from sahi.prediction import ObjectPrediction
    from utils.yolo_wrapper import YOLOv11PoseDetectionModel
    from ffp_amd import synth
    from oracle import sahi_ref, ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    from PIL import Image
    frame = synth.synthetic_frame(420, 560, seed=9)
    path = str(tmp_path / "in.png")
    Image.fromarray(frame).save(path)
    model = YOLOv11PoseDetectionModel(model_path="synthetic:yolo11n-pose", confidence_threshold=0.5, device="cuda:0", image_size=256, load_at_init=True)
    res = get_sliced_prediction(path, model, slice_height=256, slice_width=256, overlap_height_ratio=0.2, overlap_width_ratio=0.2)
    assert set(res.durations_in_seconds) == {"slice", "prediction", "postprocess"}
    assert all(isinstance(p, ObjectPrediction) for p in res.object_prediction_list)
    ref = Yolo11PoseRef(synth.yolo11_pose_weights("n"), "n")
    exp = sahi_ref.get_sliced_prediction(frame, lambda im: ultra_post.predict(ref, im, 256, 0.5, 0.7, 300), 256, 256, 0.2, 0.2)
    got = np.asarray([p.bbox.to_xyxy() for p in res.object_prediction_list], np.float32).reshape(-1, 4)
    eb = np.asarray([d.bbox for d in exp], np.float32).reshape(-1, 4)
    assert abs(len(got) - len(eb)) <= 1 and len(eb) > 0
    assert np.mean([np.array_equal(eb[i], got[j]) for i, j, _ in match_by_iou(eb, got)]) >= 0.95
    # the generic (sequential, one perform_inference per slice) path gives the same list as the batched one
    batch_fn = YOLOv11PoseDetectionModel.perform_inference_batch
    try:
        del YOLOv11PoseDetectionModel.perform_inference_batch
        model.keypoints_cache = {}
        seq = get_sliced_prediction(frame, model, slice_height=256, slice_width=256, overlap_height_ratio=0.2, overlap_width_ratio=0.2, verbose=0)
    finally:
        YOLOv11PoseDetectionModel.perform_inference_batch = batch_fn
    assert [p.bbox.to_xyxy() for p in seq.object_prediction_list] == [p.bbox.to_xyxy() for p in res.object_prediction_list]
    assert [p.score.value for p in seq.object_prediction_list] == [p.score.value for p in res.object_prediction_list]
    # keypoint side channel
    out = model.attach_keypoints_to_predictions(res.object_prediction_list)
    assert sum(hasattr(p, "keypoints") for p in out) >= max(1, len(out) - 1)
    assert all(p.keypoints.shape == (5, 3) for p in out if hasattr(p, "keypoints"))
    # direct call form used by eval/eval_dual.py:245
    r = model.model(frame[..., ::-1].copy(), conf=0.5, verbose=False)
    assert hasattr(r[0].boxes, "xyxy") and r[0].boxes.xyxy.shape[1] == 4
    # single prediction, empty result convention
    model.confidence_threshold = 0.9999
    e = get_prediction(frame[:64, :64], model)
    assert e.object_prediction_list == []
    with pytest.raises(ValueError):
        YOLOv11PoseDetectionModel(model_path=None)
    with pytest.raises(ValueError):
        get_sliced_prediction(frame, model, 256, 256, postprocess_type="BOGUS")


def test_face_enhancer_surface(compat, tmp_path):
    from utils.enhancer import FaceEnhancer, enhance_face_crops_batch, create_enhancement_summary, get_available_models
    from ffp_amd import synth
    from oracle import rrdbnet_ref
    from PIL import Image
    enh = FaceEnhancer(model_name="RealESRGAN_x4plus", model_path="synthetic:RealESRGAN_x4plus", scale=4, tile=400, half=True)
    crop = synth.synthetic_frame(64, 64, seed=2, n_blobs=3)[8:40, 10:46, ::-1].copy()
    out, ok = enh.enhance_image(crop)
    assert ok and out.shape == (128, 144, 3)
    ref = rrdbnet_ref.enhance(rrdbnet_ref.RRDBNetRef(synth.rrdbnet_weights(4, 23), 4, 23), crop)
    assert psnr_u8(out, ref) >= 50.0
    tiny, ok = enh.enhance_image(crop[:3, :3])
    assert not ok and tiny.shape == (3, 3, 3)                     # never raises, returns the input (utils/enhancer.py:205-208)
    out2, ok = enh.enhance_image(Image.fromarray(crop[..., ::-1]))
    assert ok and np.array_equal(out2, out)
    d = tmp_path / "crops"
    d.mkdir()
    Image.fromarray(crop[..., ::-1]).save(str(d / "a_face_1_conf_0.90.png"))
    (d / "broken.jpg").write_bytes(b"not an image")
    res = enhance_face_crops_batch(str(d), enh, prefix="t")
    assert res["statistics"]["total_files"] == 2 and res["statistics"]["successful"] == 1 and res["statistics"]["failed"] == 1
    create_enhancement_summary(res, str(tmp_path / "s.txt"))
    assert os.path.getsize(str(tmp_path / "s.txt")) > 0
    assert "RealESRGAN_x4plus" in get_available_models() and enh.get_model_info()["is_loaded"]
    x2 = FaceEnhancer(model_name="RealESRGAN_x2plus", model_path="synthetic:x2", scale=4)
    assert x2.scale == 2                                          # reference quirk: 'x2' in the name forces scale 2


def test_batched_crop_enhancement_equals_the_per_file_loop(compat, tmp_path):
    """enhance_face_crops_batch (pipeline_v1_detection_first/app_v1.py:100-104 -> utils/enhancer.py:344-391) sends every readable crop through ONE
    ragged GPU batch; the files it writes, the statistics and the order are those of the reference's per-file loop (run here with the batch
    switched off), and FaceEnhancer.enhance_images == a list of enhance_image calls."""
    import utils.enhancer as ue
    from ffp_amd import synth
    from PIL import Image
    enh = ue.FaceEnhancer(model_name="RealESRGAN_x4plus", model_path="synthetic:RealESRGAN_x4plus", scale=4, tile=400, half=True)
    f = synth.synthetic_frame(256, 256, seed=9, n_blobs=9)
    crops = [f[0:24, 0:24], f[10:58, 30:62], f[100:196, 50:146], f[5:8, 5:8], f[40:73, 90:107]]
    d = tmp_path / "crops"
    d.mkdir()
    for k, c in enumerate(crops):
        Image.fromarray(c).save(str(d / f"img_face_{k}_conf_0.9{k}.{'jpg' if k % 2 else 'png'}"), quality=95)
    (d / "broken.jpg").write_bytes(b"\xff\xd8 not a jpeg")
    seen = []
    res_b = ue.enhance_face_crops_batch(str(d), enh, prefix="b", progress_callback=lambda i, n, name: seen.append((i, n, name)))

    class PerFile(ue.FaceEnhancer):                 # a subclass: the batch path steps aside (type(enhancer) is not FaceEnhancer)
        pass
    enh.__class__ = PerFile
    res_p = ue.enhance_face_crops_batch(str(d), enh, prefix="p")
    enh.__class__ = ue.FaceEnhancer
    assert res_b["statistics"]["total_files"] == 6 and [s[0] for s in seen] == [1, 2, 3, 4, 5, 6]
    for key in ("successful", "failed", "total_files"):
        assert res_b["statistics"][key] == res_p["statistics"][key], key
    assert res_b["statistics"]["successful"] == 4 and res_b["statistics"]["failed"] == 2          # the 3 x 3 crop and the broken file
    assert [os.path.basename(p)[2:] for p in res_b["enhanced_files"]] == [os.path.basename(p)[2:] for p in res_p["enhanced_files"]]
    for a, b in zip(res_b["enhanced_files"], res_p["enhanced_files"]):
        assert open(a, "rb").read() == open(b, "rb").read(), a
    for ia, ib in zip(res_b["enhancement_info"], res_p["enhancement_info"]):
        assert ia["original_size"] == ib["original_size"] and ia["enhanced_size"] == ib["enhanced_size"] and ia["success"] and ib["success"]
    bgr = [np.ascontiguousarray(c[..., ::-1]) for c in crops]
    many = enh.enhance_images(bgr)
    for c, (o, ok) in zip(bgr, many):
        o1, ok1 = enh.enhance_image(c)
        assert ok == ok1 and np.array_equal(o, o1)


def test_sliced_prediction_reads_jpg_and_ndarray_without_pil_round_trips(compat, tmp_path):
    """The ndarray fast path and the codec's .jpg path give what the PIL path gives (same pixels -> same detections), and PredictionResult.image
    is still the PIL picture the reference exposes (docs sahi/prediction.py:160-165), made on first use."""
    from PIL import Image
    from sahi.predict import get_sliced_prediction
    from sahi.utils.cv import read_image_as_array, read_image_as_pil
    from utils.yolo_wrapper import YOLOv11PoseDetectionModel
    from ffp_amd import synth, _lib
    f = synth.synthetic_frame(700, 900, seed=4, n_blobs=20)
    p = tmp_path / "f.jpg"
    p.write_bytes(_lib.jpeg_encode(f, 95, bgr=False))
    assert np.array_equal(read_image_as_array(str(p)), np.asarray(read_image_as_pil(str(p))))
    assert read_image_as_array(f) is f
    m = YOLOv11PoseDetectionModel(model_path="synthetic:yolo11n-pose", confidence_threshold=0.25, device="cuda:0", image_size=512)
    kw = dict(slice_height=512, slice_width=512, overlap_height_ratio=0.2, overlap_width_ratio=0.2, verbose=0)
    r_path = get_sliced_prediction(str(p), m, **kw)
    r_arr = get_sliced_prediction(np.asarray(Image.open(str(p)).convert("RGB")), m, **kw)
    key = lambda r: [(o.bbox.to_xyxy(), round(o.score.value, 6)) for o in r.object_prediction_list]
    assert key(r_path) == key(r_arr)
    assert (r_path.image_width, r_path.image_height) == (900, 700) and (r_arr.image_width, r_arr.image_height) == (900, 700)
    assert isinstance(r_arr.image, Image.Image) and r_arr.image.size == (900, 700)
