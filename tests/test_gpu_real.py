"""GPU parity on REAL image content: photographs and face crops cut from the reference's leftover run data
(tests/golden/real/*.png, see tests/golden/make_real_fixtures.py) against oracle outputs committed in
tests/golden/real_expected.npz. The synthetic frames of the other tests are smooth; these put JPEG block edges, sensor noise
and skin texture through the fixed-point LetterBox resize (down- AND up-scaling), the stem, the whole detector and the SR net.
Bars as elsewhere: float boxes IoU >= 0.999, same count / classes, score +-2e-4; SR fp32 <= 1 LSB, fp16 >= 50 dB."""
import os

import numpy as np
import pytest
from PIL import Image

from util import match_by_iou, psnr_u8

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("mode", ["f32", "f32x3"])
def test_detector_on_photographs(gpu_lib, mode):
    from ffp_amd import synth
    z = np.load(os.path.join(G, "real_expected.npz"))
    det = gpu_lib.Detector(synth.yolo11_pose_weights("n"), arch="n", precision=gpu_lib.PREC_F32 if mode == "f32" else gpu_lib.PREC_F32X3)
    tot = same = 0
    for k, c in enumerate(z["cases"]):
        name, sub, imgsz = str(c).split("|")
        img = np.asarray(Image.open(os.path.join(G, "real", name + ".png")).convert("RGB"))
        h, w = img.shape[:2]
        tile = tuple(int(v) for v in sub.split(",")) if sub else (0, 0, w, h)
        raw = det.forward_raw(img, [tile], int(imgsz))[0]
        np.testing.assert_allclose(raw[4], z[f"case{k}_cls_row"], atol=2e-4, rtol=0)
        np.testing.assert_allclose(raw[:4], z[f"case{k}_box_rows"].astype(np.float32), atol=4e-2, rtol=1e-3)   # golden boxes stored as fp16
        d = det.infer_tiles(img, [tile], int(imgsz), 0.25, 0.7, 300)[0]
        exy, econf, ek = z[f"case{k}_xyxy"], z[f"case{k}_conf"], z[f"case{k}_kpts"]
        assert d.shape[0] == exy.shape[0], (c, d.shape[0], exy.shape[0])
        if exy.shape[0] == 0:
            continue
        keep = np.ones(len(exy), bool)
        if len(exy) == 300:                    # max_det cap: the last few survivors depend on sub-tolerance score differences
            keep = econf > econf.min() + 1e-3
        m = [x for x in match_by_iou(exy, d[:, :4], econf, d[:, 4]) if keep[x[0]]]
        ious = np.array([x[2] for x in m])
        assert ious.min() >= 0.999, (c, ious.min())
        i = np.array([x[0] for x in m]); j = np.array([x[1] for x in m])
        np.testing.assert_allclose(d[j, 4], econf[i], atol=2e-4)
        assert np.all(d[j, 5] == 0)
        np.testing.assert_allclose(d[j, 6:].reshape(-1, 5, 3), ek[i], atol=5e-2, rtol=1e-4)
        tot += len(m); same += int((d[j, :4].astype(int) == exy[i].astype(int)).all(1).sum())
    assert tot > 100 and same >= 0.97 * tot, (same, tot)


@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
def test_enhancer_on_real_face_crops(gpu_lib, half):
    from ffp_amd import synth
    z = np.load(os.path.join(G, "real_expected.npz"))
    enh = gpu_lib.Enhancer(synth.rrdbnet_weights(4, 23), 4, 23, half=half)
    crops = []
    for name in ("sr_face_20x28", "sr_face_23x27", "sr_face_47x54"):
        bgr = np.asarray(Image.open(os.path.join(G, "real", name + ".png")).convert("RGB"))[..., ::-1].copy()
        crops.append((name, bgr))
    outs = enh.enhance_batch([b for _, b in crops])
    for (name, bgr), out in zip(crops, outs):
        ref = z[name]
        assert out.shape == ref.shape == (bgr.shape[0] * 4, bgr.shape[1] * 4, 3)
        if half:
            assert psnr_u8(out, ref) >= 50.0, (name, psnr_u8(out, ref))
        else:
            assert np.abs(out.astype(int) - ref.astype(int)).max() <= 1 and psnr_u8(out, ref) >= 55.0, name
