"""ORACLE (test infrastructure): YOLO11{n,s}-pose forward pass, torch CPU fp32, fused Conv+BN weights.

Restates the Ultralytics graph that the reference runs through `self.model.predict(...)`
(/root/reference/utils/yolo_wrapper.py:55,74-80). The module source is upstream `ultralytics`
(unpinned, not vendored; SURVEY.md Appendix A gives the layer table this follows).
Input : float32 tensor (B, 3, H, W), already letterboxed / normalised (see ultra_post.preprocess).
Output: (B, 4 + nc + nk, A) float32 — the inference-mode output of the Pose head
        rows 0..3 = cx, cy, w, h (net-input pixels), 4..4+nc-1 = class sigmoid, rest = keypoints (x, y, sigmoid(v)).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SCALES = {"n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024)}


class Yolo11PoseRef:
    def __init__(self, weights: Dict[str, np.ndarray], scale: str = "s", nc: int = 1, kpt_shape=(5, 3)):
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in weights.items()}
        self.scale, self.nc, self.kpt_shape = scale, nc, tuple(kpt_shape)
        self.nk = kpt_shape[0] * kpt_shape[1]
        self.reg_max = 16
        self.strides = (8, 16, 32)
        self.taps: Dict[str, torch.Tensor] = {}      # optional activation taps for layer-wise parity
        self.keep_taps = False
        self.pre_hook = None                          # optional fn(name, pre_activation) -> pre_activation (calibration)

    # ---- primitive blocks --------------------------------------------------------------------
    def conv(self, x, name, k=1, s=1, g=1, act=True):
        """Ultralytics `Conv` after fuse(): conv2d(+bias) -> SiLU (autopad = k//2)."""
        y = F.conv2d(x, self.w[name + ".weight"], self.w[name + ".bias"], stride=s, padding=k // 2, groups=g)
        if self.pre_hook is not None:
            y = self.pre_hook(name, y)
        y = F.silu(y) if act else y
        if self.keep_taps:
            self.taps[name] = y
        return y

    def bottleneck(self, x, p, shortcut=True):
        y = self.conv(self.conv(x, p + ".cv1.conv", 3), p + ".cv2.conv", 3)
        return x + y if shortcut else y

    def c3k(self, x, p):
        a = self.conv(x, p + ".cv1.conv", 1)
        for j in range(2):
            a = self.bottleneck(a, f"{p}.m.{j}")
        b = self.conv(x, p + ".cv2.conv", 1)
        return self.conv(torch.cat((a, b), 1), p + ".cv3.conv", 1)

    def c3k2(self, x, p, c3k: bool):
        y = list(self.conv(x, p + ".cv1.conv", 1).chunk(2, 1))
        y.append(self.c3k(y[-1], p + ".m.0") if c3k else self.bottleneck(y[-1], p + ".m.0"))
        return self.conv(torch.cat(y, 1), p + ".cv2.conv", 1)

    def sppf(self, x, p):
        y = [self.conv(x, p + ".cv1.conv", 1)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        return self.conv(torch.cat(y, 1), p + ".cv2.conv", 1)

    def attention(self, x, p):
        B, C, H, W = x.shape
        nh = C // 64
        hd = C // nh
        kd = hd // 2
        N = H * W
        qkv = self.conv(x, p + ".qkv.conv", 1, act=False)
        q, k, v = qkv.view(B, nh, 2 * kd + hd, N).split([kd, kd, hd], dim=2)
        attn = (q.transpose(-2, -1) @ k) * (kd ** -0.5)
        attn = attn.softmax(dim=-1)
        o = (v @ attn.transpose(-2, -1)).view(B, C, H, W) + self.conv(v.reshape(B, C, H, W), p + ".pe.conv", 3, g=C, act=False)
        return self.conv(o, p + ".proj.conv", 1, act=False)

    def c2psa(self, x, p):
        y = self.conv(x, p + ".cv1.conv", 1)
        a, b = y.chunk(2, 1)
        b = b + self.attention(b, p + ".m.0.attn")
        b = b + self.conv(self.conv(b, p + ".m.0.ffn.0.conv", 1), p + ".m.0.ffn.1.conv", 1, act=False)
        return self.conv(torch.cat((a, b), 1), p + ".cv2.conv", 1)

    # ---- backbone + neck ------------------------------------------------------------------------
    def features(self, x) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        x0 = self.conv(x, "model.0.conv", 3, 2)
        x1 = self.conv(x0, "model.1.conv", 3, 2)
        x2 = self.c3k2(x1, "model.2", False)
        x3 = self.conv(x2, "model.3.conv", 3, 2)
        x4 = self.c3k2(x3, "model.4", False)
        x5 = self.conv(x4, "model.5.conv", 3, 2)
        x6 = self.c3k2(x5, "model.6", True)
        x7 = self.conv(x6, "model.7.conv", 3, 2)
        x8 = self.c3k2(x7, "model.8", True)
        x9 = self.sppf(x8, "model.9")
        x10 = self.c2psa(x9, "model.10")
        x12 = torch.cat((F.interpolate(x10, scale_factor=2.0, mode="nearest"), x6), 1)
        x13 = self.c3k2(x12, "model.13", False)
        x15 = torch.cat((F.interpolate(x13, scale_factor=2.0, mode="nearest"), x4), 1)
        x16 = self.c3k2(x15, "model.16", False)
        x18 = torch.cat((self.conv(x16, "model.17.conv", 3, 2), x13), 1)
        x19 = self.c3k2(x18, "model.19", False)
        x21 = torch.cat((self.conv(x19, "model.20.conv", 3, 2), x10), 1)
        x22 = self.c3k2(x21, "model.22", True)
        return x16, x19, x22

    # ---- Pose head --------------------------------------------------------------------------------
    def head_raw(self, feats) -> List[torch.Tensor]:
        """Per level: (B, 64 + nc + nk, H, W) raw head maps (box DFL logits | class logits | kpt raw)."""
        outs = []
        for l, x in enumerate(feats):
            C = x.shape[1]
            p = "model.23"
            box = self.conv(self.conv(self.conv(x, f"{p}.cv2.{l}.0.conv", 3), f"{p}.cv2.{l}.1.conv", 3),
                            f"{p}.cv2.{l}.2", 1, act=False)
            c = self.conv(self.conv(x, f"{p}.cv3.{l}.0.0.conv", 3, g=C), f"{p}.cv3.{l}.0.1.conv", 1)
            c3 = c.shape[1]
            c = self.conv(self.conv(c, f"{p}.cv3.{l}.1.0.conv", 3, g=c3), f"{p}.cv3.{l}.1.1.conv", 1)
            cls = self.conv(c, f"{p}.cv3.{l}.2", 1, act=False)
            kp = self.conv(self.conv(self.conv(x, f"{p}.cv4.{l}.0.conv", 3), f"{p}.cv4.{l}.1.conv", 3),
                           f"{p}.cv4.{l}.2", 1, act=False)
            outs.append(torch.cat((box, cls, kp), 1))
        return outs

    @staticmethod
    def make_anchors(shapes, strides, offset=0.5):
        pts, st = [], []
        for (h, w), s in zip(shapes, strides):
            sx = torch.arange(w, dtype=torch.float32) + offset
            sy = torch.arange(h, dtype=torch.float32) + offset
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            pts.append(torch.stack((xx, yy), -1).view(-1, 2))
            st.append(torch.full((h * w, 1), float(s), dtype=torch.float32))
        return torch.cat(pts).transpose(0, 1), torch.cat(st).transpose(0, 1)

    def decode(self, raws: List[torch.Tensor]) -> torch.Tensor:
        """Detect._inference + Pose.kpts_decode: DFL softmax expectation, dist2bbox(xywh) * stride, sigmoid."""
        B = raws[0].shape[0]
        no = 4 * self.reg_max + self.nc
        x_cat = torch.cat([r[:, :no].reshape(B, no, -1) for r in raws], 2)
        kpt = torch.cat([r[:, no:].reshape(B, self.nk, -1) for r in raws], 2)
        anchors, strides = self.make_anchors([r.shape[2:] for r in raws], self.strides)
        box, cls = x_cat.split((4 * self.reg_max, self.nc), 1)
        b, _, a = box.shape
        proj = torch.arange(self.reg_max, dtype=torch.float32).view(1, self.reg_max, 1, 1)
        dist = F.conv2d(box.view(b, 4, self.reg_max, a).transpose(2, 1).softmax(1), proj).view(b, 4, a)
        lt, rb = dist.chunk(2, 1)
        x1y1 = anchors.unsqueeze(0) - lt
        x2y2 = anchors.unsqueeze(0) + rb
        dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * strides
        y = kpt.clone()
        nd = self.kpt_shape[1]
        if nd == 3:
            y[:, 2::nd] = y[:, 2::nd].sigmoid()
        y[:, 0::nd] = (y[:, 0::nd] * 2.0 + (anchors[0] - 0.5)) * strides
        y[:, 1::nd] = (y[:, 1::nd] * 2.0 + (anchors[1] - 0.5)) * strides
        return torch.cat((dbox, cls.sigmoid(), y), 1)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.decode(self.head_raw(self.features(x)))

    @torch.no_grad()
    def forward_raw(self, x: torch.Tensor) -> List[torch.Tensor]:
        return self.head_raw(self.features(x))


def conv_flops(weights: Dict[str, np.ndarray], scale: str, h: int, w: int, nc=1, kpt_shape=(5, 3)) -> int:
    """2*MAC over convolutions + attention matmuls for one (h, w) pass (BASELINE.md §2 KAT: 14.312 GFLOP @512²)."""
    m = Yolo11PoseRef(weights, scale, nc, kpt_shape)
    m.keep_taps = True
    m.forward(torch.zeros(1, 3, h, w))
    total = 0
    for name, y in m.taps.items():
        wt = m.w[name + ".weight"]
        total += 2 * wt[0].numel() * wt.shape[0] * y.shape[2] * y.shape[3]
    # attention matmuls: q^T k (N*N*kd) and v attn^T (N*N*hd) per head
    C = m.w["model.10.m.0.attn.proj.conv.weight"].shape[0]
    nh = C // 64; hd = C // nh; kd = hd // 2
    N = (h // 32) * (w // 32)
    total += 2 * nh * N * N * (kd + hd)
    return total
