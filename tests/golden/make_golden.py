"""Generate the committed golden fixtures (tests/golden/*.npz) from the CPU oracle.

    python tests/golden/make_golden.py

The reference itself cannot produce vectors here (its third-party engines and weights are absent: SURVEY.md §8c), so
these fixtures pin the ORACLE and the HIP path against regressions on seeded inputs; they do not pin the oracle to
the reference (parity unpinned, see oracle/__init__.py). Inputs are regenerated from seeds by the tests; only the
expected outputs (and tiny inputs) are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import ffp_amd  # noqa: E402,F401
from ffp_amd import synth  # noqa: E402
from oracle import rrdbnet_ref, sahi_ref, ultra_post  # noqa: E402
from oracle.yolo11_ref import Yolo11PoseRef  # noqa: E402


def main():
    torch.set_num_threads(4)
    # 1. YOLO11n-pose raw forward on one 96x128 crop of a seeded frame (net input 128 -> letterbox 96x128)
    W = synth.yolo11_pose_weights("n")
    ref = Yolo11PoseRef(W, "n")
    frame = synth.synthetic_frame(160, 224, seed=21)
    tile = (40, 30, 168, 126)
    crop = frame[tile[1]:tile[3], tile[0]:tile[2]]
    raw = ref.forward(ultra_post.preprocess(crop, 128))[0].numpy()
    np.savez_compressed(os.path.join(HERE, "yolo11n_raw.npz"), frame_seed=21, frame_hw=(160, 224), tile=tile, imgsz=128, raw=raw)
    # 2. predict() results (post NMS, un-letterboxed) for three tiles incl. a resized full frame
    tiles = [(0, 0, 128, 128), (60, 20, 188, 148), (0, 0, 224, 160)]
    out = {}
    for i, t in enumerate(tiles):
        r = ultra_post.predict(ref, frame[t[1]:t[3], t[0]:t[2]], 128, 0.25, 0.7, 300)
        out[f"xyxy{i}"] = r.xyxy; out[f"conf{i}"] = r.conf; out[f"kpts{i}"] = r.kpts
    np.savez_compressed(os.path.join(HERE, "yolo11n_predict.npz"), frame_seed=21, frame_hw=(160, 224), tiles=np.asarray(tiles), imgsz=128,
                        conf=0.25, **out)
    # 3. sliced prediction (GREEDYNMM/IOS/0.5 and NMS/IOS/0.5 agnostic) on the same frame, 128 slices / 0.2
    for name, (pt, m, ag) in {"nmm": ("GREEDYNMM", "IOS", False), "nms": ("NMS", "IOS", True)}.items():
        d = sahi_ref.get_sliced_prediction(frame, lambda im: ultra_post.predict(ref, im, 128, 0.25, 0.7, 300), 128, 128, 0.2, 0.2, True, pt, m, 0.5, ag)
        np.savez_compressed(os.path.join(HERE, f"sliced_{name}.npz"), frame_seed=21, frame_hw=(160, 224), boxes=np.asarray([x.bbox for x in d], np.int32).reshape(-1, 4),
                            scores=np.asarray([x.score for x in d], np.float32))
    # 4. merge micro-fixture: rows in, rows out (exact)
    rng = np.random.default_rng(5)
    n = 200
    c = rng.uniform(0, 600, (n, 2)); s = np.exp(rng.uniform(np.log(8), np.log(100), n))
    b = np.trunc(np.stack([c[:, 0] - s / 2, c[:, 1] - s / 2, c[:, 0] + s / 2, c[:, 1] + s / 2], 1)).clip(0)
    b[:60] = b[rng.integers(60, n, 60)] + rng.integers(-3, 4, (60, 4)); b = b.clip(0)
    b[:, 2] = np.maximum(b[:, 2], b[:, 0] + 1); b[:, 3] = np.maximum(b[:, 3], b[:, 1] + 1)
    rows = np.concatenate([b, rng.uniform(0.05, 1, (n, 1)), np.zeros((n, 1))], 1).astype(np.float32)
    save = {"rows": rows}
    for pt in ("NMS", "GREEDYNMM"):
        for m in ("IOU", "IOS"):
            dets = [sahi_ref.Det(r[:4].tolist(), r[4], 0, src=i) for i, r in enumerate(rows)]
            o = sahi_ref.postprocess(dets, pt, m, 0.5, False)
            save[f"{pt}_{m}_boxes"] = np.asarray([x.bbox for x in o], np.float32)
            save[f"{pt}_{m}_scores"] = np.asarray([x.score for x in o], np.float32)
            save[f"{pt}_{m}_src"] = np.asarray([x.src for x in o], np.int32)
    np.savez_compressed(os.path.join(HERE, "merge_200.npz"), **save)
    # 5. Real-ESRGAN x4 on a 20x28 BGR crop (whole image and tiled 16/pad 4)
    Ws = synth.rrdbnet_weights(4, 23)
    net = rrdbnet_ref.RRDBNetRef(Ws, 4, 23)
    img = synth.synthetic_frame(64, 64, seed=33, n_blobs=4)[10:30, 20:48, ::-1].copy()
    np.savez_compressed(os.path.join(HERE, "esrgan_x4.npz"), img=img, out=rrdbnet_ref.enhance(net, img), out_tiled=rrdbnet_ref.enhance(net, img, tile=16, tile_pad=4))
    # 6. letterbox + resize fixture (uint8 exact)
    lb = ultra_post.letterbox(frame, 128)
    small = ultra_post.resize_linear_u8(frame, 100, 70)
    np.savez_compressed(os.path.join(HERE, "letterbox.npz"), frame_seed=21, frame_hw=(160, 224), letterbox128=lb, resize_100x70=small)
    # 7. real-image fixtures (tests/golden/real/*.png, cut by make_real_fixtures.py from the reference's leftover run data): detector
    #    on photographs (native, down-scaling and up-scaling letterbox), Real-ESRGAN x4 on face crops the reference's own run saved
    from PIL import Image
    real = os.path.join(HERE, "real")
    save = {}
    cases = [("det_test1", None, 256), ("det_parade", None, 256), ("det_family", None, 256), ("det_photo4k", None, 256),
             ("det_test1", (96, 40, 224, 168), 256), ("det_parade", (20, 60, 180, 200), 320)]
    for k, (name, sub, imgsz) in enumerate(cases):
        img = np.asarray(Image.open(os.path.join(real, name + ".png")).convert("RGB"))       # what SAHI hands the wrapper: RGB ndarray
        if sub is not None:
            img = img[sub[1]:sub[3], sub[0]:sub[2]]
        r = ultra_post.predict(ref, img, imgsz, 0.25, 0.7, 300)
        raw = ref.forward(ultra_post.preprocess(img, imgsz))[0].numpy()
        save[f"case{k}_xyxy"] = r.xyxy; save[f"case{k}_conf"] = r.conf; save[f"case{k}_kpts"] = r.kpts
        save[f"case{k}_cls_row"] = raw[4].astype(np.float32)
        save[f"case{k}_box_rows"] = raw[:4].astype(np.float16)        # fp16 storage: tolerance 2e-2 px in the test
    save["cases"] = np.asarray([f"{n}|{'' if s_ is None else ','.join(map(str, s_))}|{i}" for n, s_, i in cases])
    for name in ("sr_face_20x28", "sr_face_23x27", "sr_face_47x54"):
        bgr = np.asarray(Image.open(os.path.join(real, name + ".png")).convert("RGB"))[..., ::-1].copy()
        save[name] = rrdbnet_ref.enhance(net, bgr)
    np.savez_compressed(os.path.join(HERE, "real_expected.npz"), **save)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
