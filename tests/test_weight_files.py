"""Weights arrive as FILES in the reference: `RealESRGANer(model_path=…pth)` after FaceEnhancer's search over seven relative paths
(utils/enhancer.py:61-83,132-156) and `YOLO(model_path)` (utils/yolo_wrapper.py:47-56). These tests write real files — a `torch.save`d
`{"params_ema": …}` / `{"params": …}` checkpoint and an `.ffpw` container — and load them BY PATH. The CPU half checks the readers and the
search order; the GPU half (`-m gpu`) checks that a model loaded from a file gives the outputs of the same weights passed in memory."""
import os

import numpy as np
import pytest


def _sd_small():
    from ffp_amd import synth
    return synth.rrdbnet_weights(4, 1)


def test_pth_reader_prefers_params_ema_then_params(tmp_path):
    import torch
    from ffp_amd import weights_io
    W = _sd_small()
    other = {k: v + 1.0 for k, v in W.items()}
    as_t = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
    p1, p2, p3 = tmp_path / "a.pth", tmp_path / "b.pth", tmp_path / "c.pth"
    torch.save({"params_ema": as_t(W), "params": as_t(other)}, p1)           # RealESRGANer: params_ema wins
    torch.save({"params": as_t(W)}, p2)
    torch.save(as_t(W), p3)                                                     # a bare state dict
    for p in (p1, p2, p3):
        got = weights_io.load_esrgan_pth(str(p))
        assert set(got) == set(W)
        for k in W:
            assert got[k].dtype == np.float32 and np.array_equal(got[k], W[k]), k


def test_ffpw_file_round_trip(tmp_path):
    from ffp_amd import synth, weights_io
    W = synth.yolo11_pose_weights("n")
    p = tmp_path / "yolo11n-pose-face.ffpw"
    weights_io.save(str(p), W)
    got = weights_io.load(str(p))
    assert list(got) == list(W)
    for k in W:
        assert np.array_equal(got[k], W[k]), k


def test_face_enhancer_search_order(tmp_path, monkeypatch):
    """utils/enhancer.py:61-83: models/ before weights/ before the working directory; the first hit wins, as an absolute path."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "face-detection-with-yolov11-sahi-and-real-esrgan_amd", "compat"))
    from utils.enhancer import FaceEnhancer
    monkeypatch.chdir(tmp_path)
    fe = FaceEnhancer.__new__(FaceEnhancer)
    assert fe._find_model_path("RealESRGAN_x4plus") is None
    (tmp_path / "RealESRGAN_x4plus.pth").write_bytes(b"x")
    assert fe._find_model_path("RealESRGAN_x4plus") == str(tmp_path / "RealESRGAN_x4plus.pth")
    (tmp_path / "weights").mkdir()
    (tmp_path / "weights" / "RealESRGAN_x4plus.pth").write_bytes(b"x")
    assert fe._find_model_path("RealESRGAN_x4plus") == str(tmp_path / "weights" / "RealESRGAN_x4plus.pth")
    (tmp_path / "models").mkdir()
    (tmp_path / "models" / "RealESRGAN_x4plus.pth").write_bytes(b"x")
    assert fe._find_model_path("RealESRGAN_x4plus") == str(tmp_path / "models" / "RealESRGAN_x4plus.pth")
    assert fe._find_model_path("RealESRGAN_x2plus") is None


@pytest.mark.gpu
def test_face_enhancer_loads_a_pth_by_path_and_by_search(gpu_lib, tmp_path, monkeypatch):
    """A torch.save'd {"params_ema": …} found through the search order == the same weights handed over in memory, byte for byte;
    no file anywhere -> the AttributeError('startswith') the reference's callers test for (utils/enhancer.py:169)."""
    import sys
    import torch
    from ffp_amd import synth
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "face-detection-with-yolov11-sahi-and-real-esrgan_amd", "compat"))
    from utils.enhancer import FaceEnhancer
    W = synth.rrdbnet_weights(4, 23)
    monkeypatch.chdir(tmp_path)
    with pytest.raises(AttributeError, match="startswith"):
        FaceEnhancer("RealESRGAN_x4plus")                                  # nothing local: realesrgan 0.3.0 would try model_path.startswith
    (tmp_path / "weights").mkdir()
    torch.save({"params_ema": {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in W.items()}}, tmp_path / "weights" / "RealESRGAN_x4plus.pth")
    img = synth.synthetic_frame(96, 96, seed=5, n_blobs=4)[:41, :37][..., ::-1].copy()
    by_search = FaceEnhancer("RealESRGAN_x4plus")                          # model_path=None -> weights/RealESRGAN_x4plus.pth
    by_path = FaceEnhancer("RealESRGAN_x4plus", model_path=str(tmp_path / "weights" / "RealESRGAN_x4plus.pth"))
    in_memory = FaceEnhancer("RealESRGAN_x4plus", model_path=W)
    a, ok_a = by_search.enhance_image(img)
    b, ok_b = by_path.enhance_image(img)
    c, ok_c = in_memory.enhance_image(img)
    assert ok_a and ok_b and ok_c and a.shape == (164, 148, 3)
    assert np.array_equal(a, c) and np.array_equal(b, c)
    synth_named = FaceEnhancer("RealESRGAN_x4plus", model_path="synthetic:RealESRGAN_x4plus")
    d, ok_d = synth_named.enhance_image(img)
    assert ok_d and np.array_equal(d, c)


@pytest.mark.gpu
def test_detection_model_loads_an_ffpw_by_path(gpu_lib, tmp_path):
    """YOLOv11PoseDetectionModel(model_path=<file>.ffpw) (the converted checkpoint) == model_path="synthetic:…" of the same weights."""
    import sys
    from ffp_amd import synth, weights_io
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "face-detection-with-yolov11-sahi-and-real-esrgan_amd", "compat"))
    from utils.yolo_wrapper import YOLOv11PoseDetectionModel
    p = tmp_path / "yolo11n-pose-face.ffpw"
    weights_io.save(str(p), synth.yolo11_pose_weights("n"))
    kw = dict(confidence_threshold=0.25, device="cuda:0", image_size=256)
    from_file = YOLOv11PoseDetectionModel(model_path=str(p), **kw)
    from_name = YOLOv11PoseDetectionModel(model_path="synthetic:yolo11n-pose", **kw)
    img = synth.synthetic_frame(256, 256, seed=3, n_blobs=8)
    outs = []
    for m in (from_file, from_name):
        m.perform_inference(img)
        m.convert_original_predictions(shift_amount=[0, 0], full_shape=[256, 256])
        outs.append([(o.bbox.to_xyxy(), round(o.score.value, 6)) for o in m.object_prediction_list])
    assert outs[0] == outs[1]
    with pytest.raises(ValueError):
        YOLOv11PoseDetectionModel(model_path=None, **kw)                   # utils/yolo_wrapper.py:49-50
    with pytest.raises(ValueError):
        YOLOv11PoseDetectionModel(model_path=str(tmp_path / "x.pt"), **kw)  # pickled Ultralytics graphs are converted offline
