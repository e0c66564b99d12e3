"""CPU tests of the oracle: structural known-answer tests and hand-computed micro-cases.
These are what pins the oracle — the reference holds no golden vectors for this path (parity unpinned)."""
import numpy as np
import pytest
import torch

import ffp_amd  # noqa: F401
from ffp_amd import arch, synth
from oracle import rrdbnet_ref, sahi_ref, ultra_post
from oracle.yolo11_ref import Yolo11PoseRef, conv_flops


def test_parameter_counts_match_ultralytics():
    # face-pose variants (nc=1, 5 kpts) and the published nc=80 detect counts minus pose head
    assert arch.yolo11_unfused_param_count("n", 1) == 2_662_416
    assert arch.yolo11_unfused_param_count("s", 1) == 9_715_744


def test_oracle_consumes_exactly_the_inventory():
    for sc in "ns":
        W = synth.yolo11_pose_weights(sc)
        names = {s.name for s in arch.yolo11_pose_convs(sc)}
        assert {k.rsplit(".", 1)[0] for k in W} == names
        m = Yolo11PoseRef(W, sc)
        m.keep_taps = True
        out = m.forward(torch.zeros(1, 3, 64, 96))
        assert set(m.taps) == names                      # every conv is used exactly where the graph says
        assert out.shape == (1, 20, (8 * 12) + (4 * 6) + (2 * 3))


def test_anchor_counts():
    W = synth.yolo11_pose_weights("n")
    m = Yolo11PoseRef(W, "n")
    for (h, w), a in {(640, 640): 8400, (512, 512): 5376, (288, 512): 3024}.items():
        assert sum((h // s) * (w // s) for s in (8, 16, 32)) == a
    assert m.forward(torch.zeros(1, 3, 128, 128)).shape[-1] == 336


def test_flops_kats():
    # BASELINE.md §2
    assert abs(conv_flops(synth.yolo11_pose_weights("s"), "s", 512, 512) / 1e9 - 14.312) < 2e-3
    assert abs(conv_flops(synth.yolo11_pose_weights("n"), "n", 640, 640) / 1e9 - 6.620) < 2e-3
    assert rrdbnet_ref.flops_per_input_pixel(4, 23) == 35_853_696
    assert rrdbnet_ref.flops_per_input_pixel(2, 23) == 8_966_016
    assert sum(s.n_params_fused for s in arch.rrdbnet_convs()) == 16_697_987


def test_slice_grid_kats():
    b = np.asarray(sahi_ref.get_slice_bboxes(2160, 3840, 512, 512, 0.2, 0.2))
    assert len(b) == 60
    assert sorted(set(b[:, 0])) == [0, 410, 820, 1230, 1640, 2050, 2460, 2870, 3280, 3328]
    assert sorted(set(b[:, 1])) == [0, 410, 820, 1230, 1640, 1648]
    assert (b[:, 2] - b[:, 0] == 512).all() and (b[:, 3] - b[:, 1] == 512).all()
    assert len(sahi_ref.get_slice_bboxes(2160, 3840, 640, 640, 0.2, 0.2)) == 32
    assert len(sahi_ref.get_slice_bboxes(2160, 3840, 640, 640, 0.25, 0.25)) == 40
    b = np.asarray(sahi_ref.get_slice_bboxes(4320, 7680, 640, 640, 0.25, 0.25))
    assert len(b) == 144 and b[:, 0].max() == 7040 and b[:, 1].max() == 3680
    assert len(sahi_ref.get_slice_bboxes(4320, 7680, 512, 512, 0.2, 0.2)) == 209
    assert sahi_ref.get_slice_bboxes(300, 400, 512, 512, 0.2, 0.2) == [[0, 0, 400, 300]]


def test_letterbox_geometry_kats():
    assert ultra_post.letterbox_geometry(2160, 3840, 1024) == (1024, 576, 0, 0, 0, 0)
    assert ultra_post.letterbox_geometry(2160, 3840, 512) == (512, 288, 0, 0, 0, 0)
    assert ultra_post.letterbox_geometry(512, 512, 1024) == (1024, 1024, 0, 0, 0, 0)
    assert ultra_post.letterbox_geometry(200, 300, 256) == (256, 171, 10, 11, 0, 0)   # dh = 85 % 32 = 21 -> 10 / 11
    img = np.full((200, 300, 3), 7, np.uint8)
    lb = ultra_post.letterbox(img, 256)
    assert lb.shape == (192, 256, 3) and (lb[:10] == 114).all() and (lb[10:181] == 7).all() and (lb[181:] == 114).all()


def test_resize_fixed_point():
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    assert np.array_equal(ultra_post.resize_linear_u8(src, 60, 40), src)
    flat = np.full((33, 47, 3), 200, np.uint8)
    assert (ultra_post.resize_linear_u8(flat, 100, 70) == 200).all()          # constant image stays constant
    up = ultra_post.resize_linear_u8(src, 120, 80)                              # exact x2: interior = 3/4, 1/4 blends
    x = src.astype(np.float64)
    ref = 0.75 * (0.75 * x[0, 0] + 0.25 * x[0, 1]) + 0.25 * (0.75 * x[1, 0] + 0.25 * x[1, 1])
    assert np.abs(up[1, 1].astype(np.float64) - ref).max() <= 1.0
    assert np.array_equal(up[0, 0], src[0, 0])                                  # border taps clamp


def test_nms_micro_cases():
    b = np.asarray([[0, 0, 10, 10], [0, 0, 10, 7.0], [20, 20, 30, 30], [0, 0, 10, 7.1]], np.float32)
    s = np.asarray([0.9, 0.8, 0.7, 0.6], np.float32)
    # IoU(0,1) = 0.7 exactly: NOT suppressed (strict >), IoU(0,3) = 0.71 suppressed
    assert ultra_post.nms_torchvision(b, s, 0.7).tolist() == [0, 1, 2]
    # ties: stable order (lower index first)
    assert ultra_post.nms_torchvision(b[[2, 2]], np.asarray([0.5, 0.5], np.float32), 0.7).tolist() == [0]


def test_sahi_postprocess_micro_cases():
    D = sahi_ref.Det
    a, b = D([0, 0, 10, 10], 0.9), D([5, 0, 15, 10], 0.8)      # IOS exactly 0.5
    assert len(sahi_ref.postprocess([a, b], "NMS", "IOS", 0.5)) == 1              # >= thr suppresses
    out = sahi_ref.postprocess([a, b], "GREEDYNMM", "IOS", 0.5)
    assert len(out) == 1 and out[0].bbox == [0, 0, 10, 10]                        # absorbed, not merged (has_match is strict >)
    c = D([4, 0, 14, 10], 0.8)
    out = sahi_ref.postprocess([a, c], "GREEDYNMM", "IOS", 0.5)
    assert out[0].bbox == [0, 0, 14, 10] and out[0].score == pytest.approx(0.9)
    assert len(sahi_ref.postprocess([a, D([6, 0, 16, 10], 0.8)], "GREEDYNMM", "IOS", 0.5)) == 2
    # different categories never match unless class_agnostic
    e = D([0, 0, 10, 10], 0.5, cat=1)
    assert len(sahi_ref.postprocess([a, e], "NMS", "IOU", 0.5, class_agnostic=False)) == 2
    assert len(sahi_ref.postprocess([a, e], "NMS", "IOU", 0.5, class_agnostic=True)) == 1
    # output order = score descending
    out = sahi_ref.postprocess([D([0, 0, 5, 5], 0.2), D([50, 50, 60, 60], 0.9), D([100, 0, 110, 10], 0.5)], "NMS", "IOU", 0.5)
    assert [d.score for d in out] == [0.9, 0.5, 0.2]


def test_det_shift_and_clip_semantics():
    d = sahi_ref.Det([3, 4, 700, 900], 0.5, shift=[100, 200], full_shape=[600, 650])
    assert d.bbox == [3, 4, 650, 600]                                             # ObjectAnnotation clips to full_shape
    s = d.shifted()
    assert s.bbox == [103, 204, 750, 800] and s.shift == [0, 0] and s.full_shape is None
    with pytest.raises(ValueError):
        sahi_ref.Det([-1, 0, 5, 5], 0.5)


def test_sliced_prediction_orchestration_counts():
    calls = []

    def fake(im):
        calls.append(im.shape[:2])
        return ultra_post.PredictResult(np.asarray([[1.9, 2.9, 10.2, 12.7]], np.float32), np.asarray([0.8], np.float32),
                                        np.zeros(1, np.float32), np.zeros((1, 5, 3), np.float32))
    img = np.zeros((300, 500, 3), np.uint8)
    out = sahi_ref.get_sliced_prediction(img, fake, 256, 256, 0.2, 0.2, postprocess_type="NMS", postprocess_match_metric="IOU")
    assert len(calls) == 6 + 1 and calls[-1] == (300, 500)                        # 3x2 slices + the full-frame pass
    assert all(d.bbox[0] == int(d.bbox[0]) for d in out) and [1, 2, 10, 12] in [d.bbox for d in out]
    calls.clear()
    sahi_ref.get_sliced_prediction(img[:200, :200], fake, 256, 256)
    assert len(calls) == 1                                                        # single slice -> no standard prediction


def test_esrgan_enhance_shapes_and_tiling_consistency():
    W = synth.rrdbnet_weights(4, 23)
    net = rrdbnet_ref.RRDBNetRef(W, 4, 23)
    img = synth.synthetic_frame(64, 64, seed=1, n_blobs=3)[:18, :22]
    a = rrdbnet_ref.enhance(net, img)
    assert a.shape == (72, 88, 3) and a.dtype == np.uint8
    b = rrdbnet_ref.enhance(net, img, tile=64)           # tile larger than the image == untiled
    assert np.array_equal(a, b)
