"""GPU parity of the detector path (C-ABI ffp_det_*) against the CPU oracle on the same seeded inputs.

Tolerances (north_star): box IoU >= 0.999 and identical class ids vs the CPU path — asserted for the fp32 (exact-f32
MFMA) mode on float, pre-truncation boxes. The fp16 mode is a speed mode: asserted at IoU >= 0.90 / score 2e-2."""
import numpy as np
import pytest
import torch

from util import iou_xyxy, match_by_iou

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["n-f32", "s-f32", "n-f32x3", "s-f32x3"])
def models(request, gpu_lib):
    """Both fp32-grade arithmetic modes must meet the north_star bar: exact-fp32 MFMA and the fp16 hi/lo split (f32x3)."""
    from ffp_amd import synth
    from oracle.yolo11_ref import Yolo11PoseRef
    sc, mode = request.param.split("-")
    W = synth.yolo11_pose_weights(sc)
    prec = gpu_lib.PREC_F32 if mode == "f32" else gpu_lib.PREC_F32X3
    return sc, Yolo11PoseRef(W, sc), gpu_lib.Detector(W, arch=sc, precision=prec), W


@pytest.fixture(scope="module")
def frame():
    from ffp_amd import synth
    return synth.synthetic_frame(480, 640, seed=7)


TILES = [(0, 0, 256, 256), (100, 50, 356, 306), (0, 0, 640, 480), (20, 30, 320, 230), (384, 224, 640, 480)]


def test_forward_raw_matches_oracle(models, frame):
    from oracle import ultra_post
    sc, ref, det, _ = models
    outs = det.forward_raw(frame, TILES, 256)
    for t, o in zip(TILES, outs):
        crop = frame[t[1]:t[3], t[0]:t[2]]
        r = ref.forward(ultra_post.preprocess(crop, 256))[0].numpy()
        assert o.shape == r.shape, (t, o.shape, r.shape)
        np.testing.assert_allclose(o[4], r[4], atol=2e-4, rtol=0)            # class sigmoid
        np.testing.assert_allclose(o[:4], r[:4], atol=2e-2, rtol=0)          # cx,cy,w,h in net pixels
        np.testing.assert_allclose(o[5:], r[5:], atol=2e-2, rtol=1e-4)       # keypoints


@pytest.mark.parametrize("conf", [0.5, 0.05])
def test_infer_tiles_matches_oracle(models, frame, conf):
    from oracle import ultra_post
    sc, ref, det, _ = models
    res = det.infer_tiles(frame, TILES, 256, conf, 0.7, 300)
    total, same_total = 0, 0
    for t, d in zip(TILES, res):
        crop = frame[t[1]:t[3], t[0]:t[2]]
        r = ultra_post.predict(ref, crop, 256, conf, 0.7, 300)
        assert d.shape[0] == len(r), (t, d.shape[0], len(r))
        if len(r) == 0:
            continue
        total += len(r)
        # same order except for near-ties in score: match by IoU
        m = match_by_iou(r.xyxy, d[:, :4])
        ious = np.array([x[2] for x in m])
        assert ious.min() >= 0.999, (t, ious.min())
        j = np.array([x[1] for x in m])
        np.testing.assert_allclose(d[j, 4], r.conf, atol=2e-4)
        assert np.array_equal(d[j, 5].astype(int), r.cls.astype(int))
        np.testing.assert_allclose(d[j, 6:].reshape(-1, 5, 3), r.kpts, atol=5e-2, rtol=1e-4)
        # exact-int agreement of the wrapper's truncation: a <= 1.3e-3 px float difference flips an integer only when a
        # coordinate sits that close to one — floor 97 % over all tiles
        same_total += int((d[j, :4].astype(int) == r.xyxy.astype(int)).all(1).sum())
    assert total > 0
    assert same_total >= 0.97 * total, (same_total, total)


def test_fp16_mode_close(models, frame, gpu_lib):
    from oracle import ultra_post
    sc, ref, _, W = models
    det16 = gpu_lib.Detector(W, arch=sc, precision=gpu_lib.PREC_F16)
    outs = det16.forward_raw(frame, TILES[:2], 256)
    for t, o in zip(TILES[:2], outs):
        crop = frame[t[1]:t[3], t[0]:t[2]]
        r = ref.forward(ultra_post.preprocess(crop, 256))[0].numpy()
        np.testing.assert_allclose(o[4], r[4], atol=3e-2)
        hot = r[4] > 0.25
        if hot.any():
            a = np.stack([r[0] - r[2] / 2, r[1] - r[3] / 2, r[0] + r[2] / 2, r[1] + r[3] / 2], 1)[hot]
            b = np.stack([o[0] - o[2] / 2, o[1] - o[3] / 2, o[0] + o[2] / 2, o[1] + o[3] / 2], 1)[hot]
            assert np.median(iou_xyxy(a, b)) >= 0.97


def test_letterbox_geometry_kat(gpu_lib):
    # SURVEY.md §8c KAT: 3840x2160 @1024 -> 1024x576, no pad; @512 -> 512x288
    assert gpu_lib.letterbox_geometry(2160, 3840, 1024) == (1024, 576, 0, 0, 0, 0)
    assert gpu_lib.letterbox_geometry(2160, 3840, 512) == (512, 288, 0, 0, 0, 0)
    from oracle import ultra_post
    for (h, w, s) in [(200, 300, 256), (480, 640, 256), (333, 777, 640), (512, 512, 1024), (1080, 1920, 640)]:
        assert gpu_lib.letterbox_geometry(h, w, s) == ultra_post.letterbox_geometry(h, w, s)


@pytest.mark.parametrize("K", [4, 11], ids=["chan_scales_2^-4..2^4", "chan_scales_2^-11..2^11"])
def test_f32x3_detector_with_bn_fold_like_channel_scales(gpu_lib, K):
    """The whole detector in the default arithmetic (scaled fp16 hi/lo split) on weights whose output channels carry scales
    spanning 2^-K..2^K per layer (RMS-normalised so the network stays finite) — what BatchNorm folding of a trained checkpoint
    produces, unlike the unit-variance synthetic weights of the other tests. Raw head outputs vs the oracle at the usual bars, and
    against the exact-fp32 kernel's own error."""
    from ffp_amd import synth
    from oracle import ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    W = synth.yolo11_pose_weights("n")
    rng = np.random.default_rng(5)
    W2 = {}
    for n in [k[:-7] for k in W if k.endswith(".weight")]:
        w, b = W[n + ".weight"].copy(), W[n + ".bias"].copy()
        if not (n.startswith("model.23") and n.endswith(".2")) and w.shape[0] > 1:      # head output convs keep their meaning
            s = 2.0 ** rng.integers(-K, K + 1, size=w.shape[0]).astype(np.float64)
            s = (s / np.sqrt(np.mean(s ** 2))).astype(np.float32)
            w *= s[:, None, None, None]
            b *= s
        W2[n + ".weight"], W2[n + ".bias"] = w, b
    ref = Yolo11PoseRef(W2, "n")
    frame = synth.synthetic_frame(300, 400, seed=21)
    tiles = [(0, 0, 256, 256), (100, 30, 356, 286)]
    outs = {m: gpu_lib.Detector(W2, arch="n", precision=p).forward_raw(frame, tiles, 256)
            for m, p in (("f32x3", gpu_lib.PREC_F32X3), ("f32", gpu_lib.PREC_F32))}
    for i, t in enumerate(tiles):
        r = ref.forward(ultra_post.preprocess(frame[t[1]:t[3], t[0]:t[2]], 256))[0].numpy()
        assert np.isfinite(outs["f32x3"][i]).all()
        e3, e1 = np.abs(outs["f32x3"][i] - r), np.abs(outs["f32"][i] - r)
        np.testing.assert_allclose(outs["f32x3"][i][4], r[4], atol=2e-4, rtol=0)
        np.testing.assert_allclose(outs["f32x3"][i][:4], r[:4], atol=2e-2, rtol=0)
        assert e3[:5].max() <= max(3 * e1[:5].max(), 5e-3), (K, e3[:5].max(), e1[:5].max())


@pytest.mark.parametrize("arch", ["s", "n"])
@pytest.mark.parametrize("order", [0, 1], ids=["as_bgr", "as_rgb"])
def test_stem_fused_into_first_conv_runs_and_matches_oracle(gpu_lib, order, arch):
    """YOLO11s / YOLO11n in the default arithmetic compute model.0 inside model.1's loader (conv_mfma_kernel<..., STEM>, no stored stem
    output). The profile must name the fused launch (so this cannot pass on the two-kernel path), and the raw head outputs must
    meet the usual bars for a native-size slice (the dword fast path, image border on every side), an interior slice, a
    down-scaled full frame (resize + letterbox padding), an up-scaled crop and an odd-offset slice, in both channel orders."""
    from ffp_amd import synth
    from oracle import ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    W = synth.yolo11_pose_weights(arch)
    ref = Yolo11PoseRef(W, arch)
    det = gpu_lib.Detector(W, arch=arch, precision=gpu_lib.PREC_F32X3)
    frame = synth.synthetic_frame(301, 421, seed=33)              # 301 x 421 x 3 bytes is not a whole number of dwords
    tiles = [(0, 0, 256, 256), (165, 45, 421, 301), (0, 0, 421, 301), (30, 40, 190, 140), (5, 7, 261, 263)]
    det.set_profile(True)
    outs = det.forward_raw(frame, tiles, 256, chan_order=order)
    names = [e["name"] for e in det.profile_detail()]
    det.set_profile(False)
    assert any("stem_fused" in n and "model.1.conv" in n for n in names), names[:4]
    assert not any(n.endswith(" model.1.conv") for n in names)
    for t, o in zip(tiles, outs):
        crop = frame[t[1]:t[3], t[0]:t[2]]
        if order == gpu_lib.CHAN_AS_RGB:
            crop = crop[..., ::-1]              # the oracle's preprocess takes BGR like cv2.imread
        r = ref.forward(ultra_post.preprocess(np.ascontiguousarray(crop), 256))[0].numpy()
        np.testing.assert_allclose(o[4], r[4], atol=2e-4, rtol=0)
        np.testing.assert_allclose(o[:4], r[:4], atol=2e-2, rtol=0)
        np.testing.assert_allclose(o[5:], r[5:], atol=2e-2, rtol=1e-4)


def test_lanes_parallel_graph_branches_change_nothing(gpu_lib):
    """ffp_det_set_lanes(1): the head's towers and C3k's side convs become parallel branches of the captured graph. Same kernels,
    same operands, disjoint outputs: the raw head outputs must be bit-identical to the single-stream plan, on the eager first
    run and on graph replays, and the graph must still be captured."""
    from ffp_amd import synth
    W = synth.yolo11_pose_weights("s")
    frame = synth.synthetic_frame(300, 420, seed=41)
    tiles = [(0, 0, 256, 256), (164, 44, 420, 300), (0, 0, 420, 300)]
    det = gpu_lib.Detector(W, arch="s", precision=gpu_lib.PREC_F32X3)
    base = det.forward_raw(frame, tiles, 256)
    det.set_lanes(1)
    for it in range(4):
        outs = det.forward_raw(frame, tiles, 256)
        for a, b in zip(base, outs):
            assert np.array_equal(a, b), it
    assert det.graph_status() == 1
    det.set_lanes(0)
    outs = det.forward_raw(frame, tiles, 256)
    for a, b in zip(base, outs):
        assert np.array_equal(a, b)


def test_frames_stack_beyond_2_gib(gpu_lib):
    """A stack of frames larger than 2 GiB (the harness stacks the frames of a detection group into one tall frame): crops near its end
    must read the same pixels as the same crops of a small frame — the stem-fused loader rebases its 32-bit buffer offsets per item."""
    from ffp_amd import synth
    W = synth.yolo11_pose_weights("s")
    det = gpu_lib.Detector(W, arch="s", precision=gpu_lib.PREC_F32X3)
    band = synth.synthetic_frame(600, 4096, seed=5)
    rows = 180_000                                            # 180,000 x 4096 x 3 B = 2.21 GB
    big = np.zeros((rows, 4096, 3), np.uint8)
    y0 = rows - 600
    big[y0:] = band
    tiles_small = [(0, 0, 512, 512), (3584, 88, 4096, 600), (1001, 43, 1513, 555), (2000, 100, 2300, 400)]    # native slices + an up-scaled crop
    tiles_big = [(x0, y0 + a, x1, y0 + b) for x0, a, x1, b in tiles_small]
    ref = det.forward_raw(band, tiles_small, 512)
    out = det.forward_raw(big, tiles_big, 512)
    for t, a, b in zip(tiles_small, ref, out):                # same items, same batch mates: the same bits
        assert np.array_equal(a, b), t
