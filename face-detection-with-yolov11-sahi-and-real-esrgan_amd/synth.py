"""Seeded random-init weights and synthetic frames.

No checkpoint of the reference exists offline (its `models/` is git-ignored: /root/reference/.gitignore:7-8),
so benchmarks and parity tests run the real architectures with deterministic random weights, as
BASELINE.md §3 / SURVEY.md §8(d) prescribe: variance-preserving conv init with BatchNorm folded, a classifier
prior so that ~1 % of anchors clear conf=0.5, and residual-scaled ESRGAN weights whose output stays inside [0,1].
"""
from __future__ import annotations

import json
import os
from typing import Dict

import numpy as np

from . import arch


_CALIB = None


def _calib(key: str) -> Dict[str, float]:
    """Per-conv scalar multipliers (tests/golden/make_synth_calibration.py) standing in for trained BatchNorm statistics."""
    global _CALIB
    if _CALIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_calib.json")
        with open(path) as f:
            _CALIB = json.load(f)
    return _CALIB[key]


def yolo11_pose_weights(scale: str = "s", nc: int = 1, kpt_shape=(5, 3), seed: int = 0,
                        calibrated: bool = True) -> Dict[str, np.ndarray]:
    """Fused (conv.weight, conv.bias) tensors for every conv of YOLO11{n,s}-pose, He-style init, then (calibrated)
    each conv rescaled so its pre-activation output has unit std on a synthetic frame; class-logit bias -4.6
    (~1 % of anchors clear conf=0.5)."""
    rng = np.random.default_rng(seed)
    cal = _calib(f"yolo11{scale}-pose") if calibrated else None
    out: Dict[str, np.ndarray] = {}
    for s in arch.yolo11_pose_convs(scale, nc, kpt_shape):
        fan_in = (s.c1 // s.g) * s.k * s.k
        std = np.sqrt((2.0 if s.act else 1.0) / fan_in)
        w = rng.standard_normal(s.weight_shape, dtype=np.float32) * np.float32(std)
        b = rng.standard_normal((s.c2,), dtype=np.float32) * np.float32(0.1)
        if ".cv3." in s.name and s.name.endswith(".2"):
            b = b * 0 + np.float32(-4.6)
        if cal is not None:
            w = w * np.float32(cal[s.name])
        out[s.name + ".weight"] = w
        out[s.name + ".bias"] = b
    return out


def rrdbnet_weights(scale: int = 4, num_block: int = 23, seed: int = 0, calibrated: bool = True) -> Dict[str, np.ndarray]:
    """RRDBNet conv weights/biases; (calibrated) unit-std conv outputs, conv_last centred on 0.5 so the uint8 output
    is neither black nor saturated."""
    rng = np.random.default_rng(seed + 1000)
    cal = _calib(f"rrdbnet_x{scale}") if calibrated else None
    out: Dict[str, np.ndarray] = {}
    for s in arch.rrdbnet_convs(scale, num_block):
        fan_in = s.c1 * 9
        w = rng.standard_normal(s.weight_shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
        b = rng.standard_normal((s.c2,), dtype=np.float32) * np.float32(0.05)
        if s.name == "conv_last":
            b = b * 0 + np.float32(0.5)
        if cal is not None:
            key = s.name if s.name in cal else s.name.replace(f"body.{s.name.split('.')[1]}.", "body.0.") if s.name.startswith("body.") else s.name
            w = w * np.float32(cal[key])
        out[s.name + ".weight"] = w
        out[s.name + ".bias"] = b
    return out


def _octave_noise(rng: np.random.Generator, h: int, w: int, octaves=(64, 32, 16, 8)) -> np.ndarray:
    """Sum of bilinearly upsampled uniform noise grids -> float32 (h, w) in ~[0, 1]."""
    acc = np.zeros((h, w), np.float32)
    amp, tot = 1.0, 0.0
    for cell in octaves:
        gh, gw = h // cell + 2, w // cell + 2
        g = rng.random((gh, gw), dtype=np.float32)
        ys = (np.arange(h, dtype=np.float32) / cell)
        xs = (np.arange(w, dtype=np.float32) / cell)
        y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
        acc += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
        tot += amp
        amp *= 0.5
    return acc / tot


def synthetic_frame(h: int, w: int, seed: int = 0, n_blobs: int = 48) -> np.ndarray:
    """uint8 (h, w, 3) frame: low-frequency background + face-like elliptical blobs."""
    rng = np.random.default_rng(seed)
    img = np.stack([_octave_noise(rng, h, w) for _ in range(3)], axis=-1)
    img = 0.15 + 0.7 * img
    for _ in range(n_blobs):
        s = int(np.exp(rng.uniform(np.log(16), np.log(min(160, h // 2, w // 2)))))
        cy = int(rng.integers(s, max(s + 1, h - s))); cx = int(rng.integers(s, max(s + 1, w - s)))
        y0, y1, x0, x1 = max(0, cy - s), min(h, cy + s), max(0, cx - s), min(w, cx + s)
        yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float32)
        r2 = ((yy - cy) / (0.5 * s)) ** 2 + ((xx - cx) / (0.38 * s)) ** 2
        m = np.clip(1.2 - r2, 0, 1)[..., None]
        tone = np.array([0.85, 0.65, 0.55], np.float32) * rng.uniform(0.6, 1.1)
        eyes = ((np.abs(yy - (cy - 0.12 * s)) < 0.05 * s) & (np.abs(np.abs(xx - cx) - 0.15 * s) < 0.06 * s))[..., None]
        patch = img[y0:y1, x0:x1]
        patch[:] = patch * (1 - m) + (tone * (1 - 0.7 * eyes)) * m
    return np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8)
