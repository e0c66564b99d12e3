"""Tensor inventories of the two networks on the hot path.

The native engine (csrc/graph_yolo11.cpp, csrc/graph_rrdb.cpp) looks weights up BY NAME in an
FFPW container (weights_io.py); this module is the single Python-side statement of which names and
shapes exist, used by synth.py (random-init weights, there are no checkpoints offline) and by the
checkpoint converters.

Names follow the upstream state-dict keys so that a real checkpoint converts 1:1:
  * YOLO11-pose (Ultralytics `yolo11-pose.yaml`; reference loads it at utils/yolo_wrapper.py:55):
    `model.{i}...conv.weight` + `model.{i}...conv.bias` where the bias is the BatchNorm folded into the
    conv (Ultralytics fuses Conv+BN at inference); plain `nn.Conv2d` heads keep `.weight/.bias`.
  * RRDBNet (basicsr 1.4.2; reference builds it at utils/enhancer.py:121-128): `conv_first`,
    `body.{b}.rdb{r}.conv{c}`, `conv_body`, `conv_up1`, `conv_up2`, `conv_hr`, `conv_last`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

# [depth, width, max_channels] of yolo11-pose.yaml
YOLO11_SCALES = {"n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024)}


@dataclass(frozen=True)
class ConvSpec:
    name: str      # prefix; tensors are name+'.weight' (c2, c1//g, k, k) and name+'.bias' (c2,)
    c1: int
    c2: int
    k: int
    s: int = 1
    g: int = 1
    act: bool = True   # SiLU after (folded) BN; False = linear

    @property
    def weight_shape(self) -> Tuple[int, int, int, int]:
        return (self.c2, self.c1 // self.g, self.k, self.k)

    @property
    def n_params_fused(self) -> int:
        return self.c2 * (self.c1 // self.g) * self.k * self.k + self.c2


def _make_divisible(x: float, d: int = 8) -> int:
    return int(math.ceil(x / d) * d)


def yolo11_channels(scale: str):
    _, width, max_ch = YOLO11_SCALES[scale]
    return lambda c: _make_divisible(min(c, max_ch) * width, 8)


def _c3k2(specs: List[ConvSpec], p: str, c1: int, c2: int, c3k: bool, e: float):
    c = int(c2 * e)
    specs.append(ConvSpec(f"{p}.cv1.conv", c1, 2 * c, 1))
    specs.append(ConvSpec(f"{p}.cv2.conv", 3 * c, c2, 1))
    if c3k:   # C3k(c, c, n=2): two 1x1 in, two k3 Bottlenecks (e=1.0), 1x1 out
        c_ = int(c * 0.5)
        specs.append(ConvSpec(f"{p}.m.0.cv1.conv", c, c_, 1))
        specs.append(ConvSpec(f"{p}.m.0.cv2.conv", c, c_, 1))
        specs.append(ConvSpec(f"{p}.m.0.cv3.conv", 2 * c_, c, 1))
        for j in range(2):
            specs.append(ConvSpec(f"{p}.m.0.m.{j}.cv1.conv", c_, c_, 3))
            specs.append(ConvSpec(f"{p}.m.0.m.{j}.cv2.conv", c_, c_, 3))
    else:     # Bottleneck(c, c, e=0.5)
        c_ = int(c * 0.5)
        specs.append(ConvSpec(f"{p}.m.0.cv1.conv", c, c_, 3))
        specs.append(ConvSpec(f"{p}.m.0.cv2.conv", c_, c, 3))


def yolo11_pose_convs(scale: str = "s", nc: int = 1, kpt_shape=(5, 3)) -> List[ConvSpec]:
    """Every (fused) convolution of YOLO11{n,s}-pose, SURVEY.md Appendix A."""
    ch = yolo11_channels(scale)
    S: List[ConvSpec] = []
    c64, c128, c256, c512, c1024 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    S.append(ConvSpec("model.0.conv", 3, c64, 3, 2))
    S.append(ConvSpec("model.1.conv", c64, c128, 3, 2))
    _c3k2(S, "model.2", c128, c256, False, 0.25)
    S.append(ConvSpec("model.3.conv", c256, c256, 3, 2))
    _c3k2(S, "model.4", c256, c512, False, 0.25)
    S.append(ConvSpec("model.5.conv", c512, c512, 3, 2))
    _c3k2(S, "model.6", c512, c512, True, 0.5)
    S.append(ConvSpec("model.7.conv", c512, c1024, 3, 2))
    _c3k2(S, "model.8", c1024, c1024, True, 0.5)
    # SPPF
    S.append(ConvSpec("model.9.cv1.conv", c1024, c1024 // 2, 1))
    S.append(ConvSpec("model.9.cv2.conv", c1024 // 2 * 4, c1024, 1))
    # C2PSA(c1024, n=1)
    h = c1024 // 2
    nh = h // 64
    kd = (h // nh) // 2
    S.append(ConvSpec("model.10.cv1.conv", c1024, 2 * h, 1))
    S.append(ConvSpec("model.10.cv2.conv", 2 * h, c1024, 1))
    S.append(ConvSpec("model.10.m.0.attn.qkv.conv", h, h + 2 * nh * kd, 1, act=False))
    S.append(ConvSpec("model.10.m.0.attn.proj.conv", h, h, 1, act=False))
    S.append(ConvSpec("model.10.m.0.attn.pe.conv", h, h, 3, 1, h, act=False))
    S.append(ConvSpec("model.10.m.0.ffn.0.conv", h, 2 * h, 1))
    S.append(ConvSpec("model.10.m.0.ffn.1.conv", 2 * h, h, 1, act=False))
    # neck
    _c3k2(S, "model.13", c1024 + c512, c512, False, 0.5)
    _c3k2(S, "model.16", c512 + c512, c256, False, 0.5)
    S.append(ConvSpec("model.17.conv", c256, c256, 3, 2))
    _c3k2(S, "model.19", c256 + c512, c512, False, 0.5)
    S.append(ConvSpec("model.20.conv", c512, c512, 3, 2))
    _c3k2(S, "model.22", c512 + c1024, c1024, True, 0.5)
    # Pose head on (P3, P4, P5)
    chs = (c256, c512, c1024)
    nk = kpt_shape[0] * kpt_shape[1]
    c2 = max(16, chs[0] // 4, 64)
    c3 = max(chs[0], min(nc, 100))
    c4 = max(chs[0] // 4, nk)
    for l, x in enumerate(chs):
        S.append(ConvSpec(f"model.23.cv2.{l}.0.conv", x, c2, 3))
        S.append(ConvSpec(f"model.23.cv2.{l}.1.conv", c2, c2, 3))
        S.append(ConvSpec(f"model.23.cv2.{l}.2", c2, 64, 1, act=False))
        S.append(ConvSpec(f"model.23.cv3.{l}.0.0.conv", x, x, 3, 1, x))
        S.append(ConvSpec(f"model.23.cv3.{l}.0.1.conv", x, c3, 1))
        S.append(ConvSpec(f"model.23.cv3.{l}.1.0.conv", c3, c3, 3, 1, c3))
        S.append(ConvSpec(f"model.23.cv3.{l}.1.1.conv", c3, c3, 1))
        S.append(ConvSpec(f"model.23.cv3.{l}.2", c3, nc, 1, act=False))
        S.append(ConvSpec(f"model.23.cv4.{l}.0.conv", x, c4, 3))
        S.append(ConvSpec(f"model.23.cv4.{l}.1.conv", c4, c4, 3))
        S.append(ConvSpec(f"model.23.cv4.{l}.2", c4, nk, 1, act=False))
    return S


def yolo11_unfused_param_count(scale: str, nc: int, kpt_shape=(5, 3)) -> int:
    """Parameter count of the *unfused* module tree as Ultralytics reports it (conv weight without bias +
    BatchNorm weight/bias for Conv blocks; weight+bias for plain Conv2d heads; + 16 frozen DFL weights).
    KAT: n/s pose-face = 2,662,416 / 9,715,744."""
    total = 16
    for s in yolo11_pose_convs(scale, nc, kpt_shape):
        wsz = s.c2 * (s.c1 // s.g) * s.k * s.k
        plain = not s.name.endswith(".conv")
        total += wsz + (s.c2 if plain else 2 * s.c2)
    return total


def rrdbnet_convs(scale: int = 4, num_block: int = 23, num_feat: int = 64, num_grow_ch: int = 32,
                  num_in_ch: int = 3, num_out_ch: int = 3) -> List[ConvSpec]:
    """Every convolution of basicsr RRDBNet (all 3x3, stride 1, bias, no norm); SURVEY.md Appendix D.1."""
    cin = num_in_ch * (4 if scale == 2 else 16 if scale == 1 else 1)
    S = [ConvSpec("conv_first", cin, num_feat, 3, act=False)]
    for b in range(num_block):
        for r in (1, 2, 3):
            for c in range(1, 6):
                c1 = num_feat + (c - 1) * num_grow_ch
                c2 = num_grow_ch if c < 5 else num_feat
                S.append(ConvSpec(f"body.{b}.rdb{r}.conv{c}", c1, c2, 3, act=(c < 5)))
    S.append(ConvSpec("conv_body", num_feat, num_feat, 3, act=False))
    S.append(ConvSpec("conv_up1", num_feat, num_feat, 3))
    S.append(ConvSpec("conv_up2", num_feat, num_feat, 3))
    S.append(ConvSpec("conv_hr", num_feat, num_feat, 3))
    S.append(ConvSpec("conv_last", num_feat, num_out_ch, 3, act=False))
    return S
