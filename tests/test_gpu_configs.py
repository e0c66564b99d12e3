"""GPU parity on the configurations BASELINE.json names and the reference actually runs, which the round-1 suite did
not reach:

  * the wrapper default `image_size=1024` (/root/reference/utils/yolo_wrapper.py:13,78; pipeline_v4_yolo/1_Inference.py:34):
    every slice is letterbox-UPscaled before the forward pass (cv2 INTER_LINEAR fixed point);
  * config 1: one 640x640 image, YOLO11n, no SAHI (pipeline_v4_yolo/inference_direct.py:15-30 plumbing), A = 8400 anchors;
  * config 5 shape on one GPU: 7680x4320, SAHI 640/0.25 (144 slices + full frame), 128 crops — through size-independent
    properties plus a one-item oracle spot check (the oracle needs minutes for a whole 8K frame).

Bars as in test_gpu_detector.py: float boxes IoU >= 0.999, identical count and classes, score +-2e-4.
"""
import numpy as np
import pytest

from util import match_by_iou, psnr_u8

pytestmark = pytest.mark.gpu


def check_item(d, r, tag):
    assert d.shape[0] == len(r), (tag, d.shape[0], len(r))
    if len(r) == 0:
        return 0, 0
    m = match_by_iou(r.xyxy, d[:, :4], r.conf, d[:, 4])
    ious = np.array([x[2] for x in m])
    assert ious.min() >= 0.999, (tag, ious.min())
    j = np.array([x[1] for x in m])
    np.testing.assert_allclose(d[j, 4], r.conf, atol=2e-4)
    assert np.array_equal(d[j, 5].astype(int), r.cls.astype(int))
    np.testing.assert_allclose(d[j, 6:].reshape(-1, 5, 3), r.kpts, atol=5e-2, rtol=1e-4)
    return len(r), int((d[j, :4].astype(int) == r.xyxy.astype(int)).all(1).sum())


@pytest.mark.parametrize("mode", ["f32", "f32x3"])
def test_upscaling_letterbox_128_to_256(gpu_lib, mode):
    """Tiles smaller than the network input: 128x128 and 100x77 crops letterboxed UP to 256 (gain 2 and 2.56)."""
    from ffp_amd import synth
    from oracle import ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    Wn = synth.yolo11_pose_weights("n")
    ref = Yolo11PoseRef(Wn, "n")
    det = gpu_lib.Detector(Wn, arch="n", precision=gpu_lib.PREC_F32 if mode == "f32" else gpu_lib.PREC_F32X3)
    frame = synth.synthetic_frame(300, 400, seed=21)
    tiles = [(0, 0, 128, 128), (200, 100, 328, 228), (50, 60, 150, 137), (272, 172, 400, 300)]
    # the resized network input itself (fixed-point bilinear) is compared through the raw head output
    for t, o in zip(tiles, det.forward_raw(frame, tiles, 256)):
        r = ref.forward(ultra_post.preprocess(frame[t[1]:t[3], t[0]:t[2]], 256))[0].numpy()
        assert o.shape == r.shape
        np.testing.assert_allclose(o[4], r[4], atol=2e-4, rtol=0)
        np.testing.assert_allclose(o[:4], r[:4], atol=2e-2, rtol=0)
    tot = same = 0
    for conf in (0.5, 0.05):
        for t, d in zip(tiles, det.infer_tiles(frame, tiles, 256, conf, 0.7, 300)):
            n, s = check_item(d, ultra_post.predict(ref, frame[t[1]:t[3], t[0]:t[2]], 256, conf, 0.7, 300), (t, conf))
            tot += n; same += s
    assert tot > 0 and same >= 0.97 * tot, (same, tot)


def test_reference_default_512_slice_at_imgsz_1024(gpu_lib):
    """One 512x512 slice through YOLO11s at the wrapper's default image_size=1024 (A = 21504 anchors) + the 4K-shaped
    full-frame letterbox case at reduced size (960x540 @ 1024 -> 1024x576: upscale, no padding)."""
    from ffp_amd import synth
    from oracle import ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    Ws = synth.yolo11_pose_weights("s")
    ref = Yolo11PoseRef(Ws, "s")
    det = gpu_lib.Detector(Ws, arch="s", precision=gpu_lib.PREC_F32X3)
    frame = synth.synthetic_frame(540, 960, seed=22)
    assert gpu_lib.letterbox_geometry(540, 960, 1024) == (1024, 576, 0, 0, 0, 0)
    tiles = [(200, 10, 712, 522), (0, 0, 960, 540)]
    raw = det.forward_raw(frame, tiles[:1], 1024)[0]
    assert raw.shape == (20, 21504)
    tot = same = 0
    for t, d in zip(tiles, det.infer_tiles(frame, tiles, 1024, 0.25, 0.7, 300)):
        n, s = check_item(d, ultra_post.predict(ref, frame[t[1]:t[3], t[0]:t[2]], 1024, 0.25, 0.7, 300), t)
        tot += n; same += s
    assert tot > 0 and same >= 0.97 * tot, (same, tot)


@pytest.mark.parametrize("mode", ["f32", "f32x3"])
def test_config1_640_yolo11n_no_sahi(gpu_lib, mode):
    """BASELINE config 1: single 640x640 image, YOLO11n, one letterboxed forward (no resize: native), NMS, boxes."""
    from ffp_amd import synth
    from oracle import ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    Wn = synth.yolo11_pose_weights("n")
    ref = Yolo11PoseRef(Wn, "n")
    det = gpu_lib.Detector(Wn, arch="n", precision=gpu_lib.PREC_F32 if mode == "f32" else gpu_lib.PREC_F32X3)
    img = synth.synthetic_frame(640, 640, seed=23)
    raw = det.forward_raw(img, [(0, 0, 640, 640)], 640)[0]
    r = ref.forward(ultra_post.preprocess(img, 640))[0].numpy()
    assert raw.shape == r.shape == (20, 8400)
    np.testing.assert_allclose(raw[4], r[4], atol=2e-4, rtol=0)
    np.testing.assert_allclose(raw[:4], r[:4], atol=2e-2, rtol=0)
    tot = same = 0
    for conf in (0.5, 0.25):
        d = det.infer_tiles(img, [(0, 0, 640, 640)], 640, conf, 0.7, 300)[0]
        n, s = check_item(d, ultra_post.predict(ref, img, 640, conf, 0.7, 300), conf)
        tot += n; same += s
    assert tot > 0 and same >= 0.97 * tot


def test_config5_shape_8k_640_on_one_gpu(gpu_lib):
    """7680x4320, SAHI 640/0.25 -> 144 slices + the full-frame pass at net input 640, GREEDYNMM/IOS/0.5; Real-ESRGAN x4 on
    128 crops. Properties: slice grid KAT, determinism, fused == tile-wise composition, sharded == unsharded, score order;
    spot check: one slice and 2 crops against the oracle."""
    import torch
    from ffp_amd import pipeline, synth
    from oracle import rrdbnet_ref, ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    H, W = 4320, 7680
    sl = gpu_lib.slice_bboxes(H, W, 640, 640, 0.25, 0.25)
    assert len(sl) == 144 and tuple(sl[-1]) == (7040, 3680, 7680, 4320)
    Ws = synth.yolo11_pose_weights("s")
    det = gpu_lib.Detector(Ws, arch="s", precision=gpu_lib.PREC_F32)      # bitwise batch-independence is an exact-fp32 property (see test_gpu_fullsize)
    frame = synth.synthetic_frame(H, W, seed=5)
    kw = dict(imgsz=640, conf=0.25, iou=0.7, max_det=300, pp_type="GREEDYNMM", pp_metric="IOS", pp_thr=0.5)
    a = det.sliced_predict(frame, 640, 640, 0.25, 0.25, True, **kw)
    b = det.sliced_predict(frame, 640, 640, 0.25, 0.25, True, **kw)
    assert a.shape[0] > 0 and np.array_equal(a, b)
    assert np.all(a[:, :4] == np.trunc(a[:, :4])) and a[:, 2].max() <= W and a[:, 3].max() <= H and np.all(np.diff(a[:, 4]) <= 0)
    tiles = [tuple(t) for t in sl] + [(0, 0, W, H)]
    per = det.infer_tiles(frame, tiles, 640, 0.25, 0.7, 300)
    third = len(tiles) // 3
    parts = det.infer_tiles(frame, tiles[:third], 640, 0.25, 0.7, 300) + det.infer_tiles(frame, tiles[third:], 640, 0.25, 0.7, 300)
    for x, y in zip(per, parts):
        assert np.array_equal(x, y)                      # what two ranks would compute == what one computes
    rows = []
    for t, d in zip(tiles, per):
        d = d.copy()
        bx = np.trunc(d[:, :4])
        bx[:, 2] = np.minimum(bx[:, 2], W); bx[:, 3] = np.minimum(bx[:, 3], H)
        d[:, 0] = bx[:, 0] + t[0]; d[:, 1] = bx[:, 1] + t[1]; d[:, 2] = bx[:, 2] + t[0]; d[:, 3] = bx[:, 3] + t[1]
        d[:, 6::3] += t[0]; d[:, 7::3] += t[1]
        rows.append(d)
    merged, _ = gpu_lib.merge(np.concatenate(rows, 0), "GREEDYNMM", "IOS", 0.5, False)
    assert np.array_equal(merged, a)
    # oracle spot check on one interior slice
    ref = Yolo11PoseRef(Ws, "s")
    k = 70
    t = tiles[k]
    check_item(per[k], ultra_post.predict(ref, frame[t[1]:t[3], t[0]:t[2]], 640, 0.25, 0.7, 300), t)
    # 128 crops through the device-resident crop path; batch == single, two of them against the oracle
    Wsr = synth.rrdbnet_weights(4, 23)
    enh = gpu_lib.Enhancer(Wsr, 4, 23, half=True)
    sizes = pipeline.sr_crop_sizes(128, seed=0)
    boxes = pipeline.crop_boxes_for_sr(a, H, W, 128, sizes, seed=0)
    bgr = np.ascontiguousarray(frame[..., ::-1])
    d_frame = torch.from_numpy(bgr).cuda()
    tot = pipeline.FramePipeline.sr_out_bytes(boxes, H, W)
    out = torch.zeros((tot,), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    offs = enh.enhance_crops_dev([d_frame.data_ptr()], H, W, boxes, out.data_ptr(), tot)
    host = out.cpu().numpy()
    small = [i for i in range(128) if sizes[i] <= 32][:2]
    netref = rrdbnet_ref.RRDBNetRef(Wsr, 4, 23)
    for i in range(0, 128, 9):
        x1, y1, x2, y2 = boxes[i]
        g = host[offs[i]:offs[i] + (y2 - y1) * (x2 - x1) * 48].reshape((y2 - y1) * 4, (x2 - x1) * 4, 3)
        assert np.array_equal(g, enh.enhance(np.ascontiguousarray(bgr[y1:y2, x1:x2])))
    for i in small:
        x1, y1, x2, y2 = boxes[i]
        g = host[offs[i]:offs[i] + (y2 - y1) * (x2 - x1) * 48].reshape((y2 - y1) * 4, (x2 - x1) * 4, 3)
        assert psnr_u8(g, rrdbnet_ref.enhance(netref, bgr[y1:y2, x1:x2])) >= 50.0
