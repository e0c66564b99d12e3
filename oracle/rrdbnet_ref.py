"""ORACLE (test infrastructure): Real-ESRGAN — RRDBNet forward and RealESRGANer.enhance, torch CPU fp32.

The reference builds `RRDBNet(3, 3, 64, num_block, 32, scale)` and `RealESRGANer(scale, model_path, dni_weight=None,
model, tile, tile_pad=10, pre_pad=0, half, gpu_id)` at /root/reference/utils/enhancer.py:99-156 and calls
`self.upsampler.enhance(image, outscale=self.scale)` at :214. Both classes come from `basicsr==1.4.2` /
`realesrgan==0.3.0` (requirements.txt:12,134), not vendored, not installed: their published algorithms are restated
per SURVEY.md Appendix D. Parity unpinned (oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F


class RRDBNetRef:
    def __init__(self, weights: Dict[str, np.ndarray], scale: int = 4, num_block: int = 23):
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in weights.items()}
        self.scale, self.num_block = scale, num_block
        self.pre_hook = None                          # optional fn(name, conv_output) -> conv_output (calibration)

    def _c(self, x, name):
        y = F.conv2d(x, self.w[name + ".weight"], self.w[name + ".bias"], padding=1)
        return y if self.pre_hook is None else self.pre_hook(name, y)

    def _rdb(self, x, p):
        lr = lambda t: F.leaky_relu(t, 0.2)
        x1 = lr(self._c(x, p + ".conv1"))
        x2 = lr(self._c(torch.cat((x, x1), 1), p + ".conv2"))
        x3 = lr(self._c(torch.cat((x, x1, x2), 1), p + ".conv3"))
        x4 = lr(self._c(torch.cat((x, x1, x2, x3), 1), p + ".conv4"))
        x5 = self._c(torch.cat((x, x1, x2, x3, x4), 1), p + ".conv5")
        return x5 * 0.2 + x

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.scale == 2:
            x = F.pixel_unshuffle(x, 2)
        elif self.scale == 1:
            x = F.pixel_unshuffle(x, 4)
        feat = self._c(x, "conv_first")
        body = feat
        for b in range(self.num_block):
            out = body
            for r in (1, 2, 3):
                out = self._rdb(out, f"body.{b}.rdb{r}")
            body = out * 0.2 + body
        feat = feat + self._c(body, "conv_body")
        lr = lambda t: F.leaky_relu(t, 0.2)
        feat = lr(self._c(F.interpolate(feat, scale_factor=2, mode="nearest"), "conv_up1"))
        feat = lr(self._c(F.interpolate(feat, scale_factor=2, mode="nearest"), "conv_up2"))
        return self._c(lr(self._c(feat, "conv_hr")), "conv_last")


def enhance(net: RRDBNetRef, img_bgr: np.ndarray, tile: int = 0, tile_pad: int = 10, pre_pad: int = 0) -> np.ndarray:
    """RealESRGANer.enhance(img, outscale=scale) for a 3-channel uint8 BGR image (Appendix D.2):
    /255 -> BGR2RGB -> CHW -> (pre_pad reflect) -> (mod-pad reflect for scale 2/1) -> tile loop -> crop pads ->
    clamp -> RGB2BGR -> (x*255).round() uint8. Returns uint8 (s*h, s*w, 3) BGR."""
    s = net.scale
    img = img_bgr.astype(np.float32) / 255.0
    x = torch.from_numpy(np.ascontiguousarray(img[..., ::-1].transpose(2, 0, 1))).float().unsqueeze(0)
    if pre_pad:
        x = F.pad(x, (0, pre_pad, 0, pre_pad), "reflect")
    mod = 2 if s == 2 else 4 if s == 1 else None
    mph = mpw = 0
    if mod is not None:
        _, _, h, w = x.shape
        if h % mod:
            mph = mod - h % mod
        if w % mod:
            mpw = mod - w % mod
        x = F.pad(x, (0, mpw, 0, mph), "reflect")
    _, c, H, W = x.shape
    if tile > 0:
        out = x.new_zeros((1, c, H * s, W * s))
        tx, ty = math.ceil(W / tile), math.ceil(H / tile)
        for y in range(ty):
            for xx in range(tx):
                ox, oy = xx * tile, y * tile
                sx, ex = ox, min(ox + tile, W)
                sy, ey = oy, min(oy + tile, H)
                sxp, exp_ = max(sx - tile_pad, 0), min(ex + tile_pad, W)
                syp, eyp = max(sy - tile_pad, 0), min(ey + tile_pad, H)
                o = net.forward(x[:, :, syp:eyp, sxp:exp_])
                tw, th = ex - sx, ey - sy
                oxs, oys = (sx - sxp) * s, (sy - syp) * s
                out[:, :, sy * s:ey * s, sx * s:ex * s] = o[:, :, oys:oys + th * s, oxs:oxs + tw * s]
    else:
        out = net.forward(x)
    if mod is not None:
        _, _, h, w = out.shape
        out = out[:, :, 0:h - mph * s, 0:w - mpw * s]
    if pre_pad:
        _, _, h, w = out.shape
        out = out[:, :, 0:h - pre_pad * s, 0:w - pre_pad * s]
    o = out.squeeze(0).float().clamp_(0, 1).numpy()
    o = np.transpose(o[[2, 1, 0], :, :], (1, 2, 0))
    return (o * 255.0).round().astype(np.uint8)


def flops_per_input_pixel(scale: int = 4, num_block: int = 23) -> int:
    """2*MAC per input pixel (BASELINE.md §2 KAT: 35,853,696 for x4 / 23 blocks)."""
    rdb = 9 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64)
    cin = 3 * (4 if scale == 2 else 16 if scale == 1 else 1)
    px = 1.0 / (4 if scale == 2 else 16 if scale == 1 else 1)
    mac = px * (9 * cin * 64 + num_block * 3 * rdb + 9 * 64 * 64) + px * 4 * 9 * 64 * 64 + px * 16 * (9 * 64 * 64 + 9 * 64 * 64 + 9 * 64 * 3)
    return int(round(2 * mac))
