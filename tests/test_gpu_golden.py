"""The HIP path against the committed golden fixtures (tests/golden/*.npz) — no oracle in the loop at run time."""
import os

import numpy as np
import pytest

from util import psnr_u8

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _frame(z):
    from ffp_amd import synth
    h, w = [int(v) for v in z["frame_hw"]]
    return synth.synthetic_frame(h, w, seed=int(z["frame_seed"]))


@pytest.fixture(scope="module")
def det(gpu_lib):
    from ffp_amd import synth
    return gpu_lib.Detector(synth.yolo11_pose_weights("n"), arch="n", precision=gpu_lib.PREC_F32)


def test_raw_forward(det):
    z = np.load(os.path.join(G, "yolo11n_raw.npz"))
    out = det.forward_raw(_frame(z), [tuple(int(v) for v in z["tile"])], int(z["imgsz"]))[0]
    np.testing.assert_allclose(out[4], z["raw"][4], atol=2e-4)
    np.testing.assert_allclose(out[:4], z["raw"][:4], atol=2e-2)
    np.testing.assert_allclose(out[5:], z["raw"][5:], atol=2e-2, rtol=1e-4)


def test_predict(det):
    z = np.load(os.path.join(G, "yolo11n_predict.npz"))
    res = det.infer_tiles(_frame(z), z["tiles"].tolist(), int(z["imgsz"]), float(z["conf"]), 0.7, 300)
    for i, r in enumerate(res):
        assert r.shape[0] == z[f"xyxy{i}"].shape[0]
        np.testing.assert_allclose(r[:, :4], z[f"xyxy{i}"], atol=2e-2)
        np.testing.assert_allclose(r[:, 4], z[f"conf{i}"], atol=2e-4)
        np.testing.assert_allclose(r[:, 6:].reshape(-1, 5, 3), z[f"kpts{i}"], atol=5e-2, rtol=1e-4)


@pytest.mark.parametrize("name,cfg", [("nmm", ("GREEDYNMM", "IOS", False)), ("nms", ("NMS", "IOS", True))])
def test_sliced(det, name, cfg):
    z = np.load(os.path.join(G, f"sliced_{name}.npz"))
    out = det.sliced_predict(_frame(z), 128, 128, 0.2, 0.2, True, 128, 0.25, 0.7, 300, cfg[0], cfg[1], 0.5, cfg[2])
    assert np.array_equal(out[:, :4].astype(np.int32), z["boxes"])
    np.testing.assert_allclose(out[:, 4], z["scores"], atol=2e-4)


def test_merge_exact(gpu_lib):
    z = np.load(os.path.join(G, "merge_200.npz"))
    for pt in ("NMS", "GREEDYNMM"):
        for m in ("IOU", "IOS"):
            out, src = gpu_lib.merge(z["rows"], pt, m, 0.5)
            assert np.array_equal(out[:, :4], z[f"{pt}_{m}_boxes"])
            assert np.array_equal(out[:, 4], z[f"{pt}_{m}_scores"])
            assert np.array_equal(src, z[f"{pt}_{m}_src"])


def test_esrgan(gpu_lib):
    from ffp_amd import synth
    z = np.load(os.path.join(G, "esrgan_x4.npz"))
    W = synth.rrdbnet_weights(4, 23)
    e32 = gpu_lib.Enhancer(W, 4, 23, half=False)
    a = e32.enhance(z["img"])
    assert np.abs(a.astype(int) - z["out"].astype(int)).max() <= 1 and psnr_u8(a, z["out"]) >= 55
    b = e32.enhance(z["img"], tile=16, tile_pad=4)
    assert psnr_u8(b, z["out_tiled"]) >= 55
    e16 = gpu_lib.Enhancer(W, 4, 23, half=True)
    assert psnr_u8(e16.enhance(z["img"]), z["out"]) >= 50
