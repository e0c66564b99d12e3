"""Generate <pkg>/synth_calib.json: per-conv scalar multipliers that play the role BatchNorm plays in a trained
checkpoint — each convolution's pre-activation output has unit standard deviation on a synthetic frame, so the
random-init networks used for benchmarks/parity are numerically well conditioned (no saturated scores, no
exploding activations). Uses the CPU oracle as the forward pass; the product package only reads the JSON.

    python tests/golden/make_synth_calibration.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ffp_amd  # noqa: E402
from ffp_amd import synth  # noqa: E402
from oracle import rrdbnet_ref, ultra_post, yolo11_ref  # noqa: E402

TARGET = {".cv2.{l}.2": 2.5, ".cv3.{l}.2": 2.0, ".cv4.{l}.2": 1.0}


def calib_yolo(scale):
    W = synth.yolo11_pose_weights(scale, seed=0, calibrated=False)
    m = yolo11_ref.Yolo11PoseRef(W, scale)
    factors = {}

    def hook(name, y):
        tgt = 1.0
        for l in range(3):
            for k, v in TARGET.items():
                if name.endswith(k.format(l=l)):
                    tgt = v
        b = m.w[name + ".bias"].view(1, -1, 1, 1)
        s = tgt / float((y - b).std())
        factors[name] = s
        return (y - b) * s + b          # only the weight is rescaled, the bias is kept

    m.pre_hook = hook
    img = synth.synthetic_frame(384, 384, seed=123)
    m.forward(ultra_post.preprocess(img, 384))
    return factors


def calib_rrdb(scale):
    W = synth.rrdbnet_weights(scale, 23, seed=0, calibrated=False)
    net = rrdbnet_ref.RRDBNetRef(W, scale, 23)
    factors = {}

    def hook(name, y):
        tgt = 0.18 if name == "conv_last" else 1.0
        b = net.w[name + ".bias"].view(1, -1, 1, 1)
        s = tgt / float((y - b).std())
        factors[name] = s
        return (y - b) * s + b

    net.pre_hook = hook
    img = synth.synthetic_frame(64, 64, seed=321)
    x = torch.from_numpy(np.ascontiguousarray(img[..., ::-1].transpose(2, 0, 1))).float().unsqueeze(0) / 255
    net.forward(x)
    return factors


if __name__ == "__main__":
    out = {"yolo11n-pose": calib_yolo("n"), "yolo11s-pose": calib_yolo("s"),
           "rrdbnet_x4": calib_rrdb(4), "rrdbnet_x2": calib_rrdb(2)}
    path = os.path.join(os.path.dirname(ffp_amd.__file__), "synth_calib.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, {k: len(v) for k, v in out.items()})
