"""ORACLE (test infrastructure): Ultralytics predict() pre/post-processing around the YOLO11-pose graph.

Restates what `self.model.predict(source=image, conf=..., imgsz=..., verbose=False)` does to one ndarray
(/root/reference/utils/yolo_wrapper.py:74-80): LetterBox -> BGR2RGB/CHW//255 -> forward -> non_max_suppression
-> scale_boxes / scale_coords. Source: upstream `ultralytics` (unpinned; semantics fixed in SURVEY.md Appendix B)
and `opencv-python==4.11.0.86` for the 8-bit INTER_LINEAR resize (requirements.txt:98), `torchvision==0.14.1`
for nms (requirements.txt:170). None of them is installed here: parity unpinned (see oracle/__init__.py).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from .yolo11_ref import Yolo11PoseRef

_COEF_BITS = 11
_COEF_ONE = 1 << _COEF_BITS


def _round_half_even_short(x: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(x), -32768, 32767).astype(np.int32)


def resize_linear_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.resize(src, (dw, dh), interpolation=INTER_LINEAR) for uint8 HWC, restating OpenCV's fixed-point path:
    11-bit coefficients (saturate_cast<short>(c * 2048)), horizontal pass in int32, vertical pass
    ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2  (imgproc/resize.cpp HResizeLinear / VResizeLinear<uchar,int,short>)."""
    sh, sw = src.shape[:2]
    if (sw, sh) == (dw, dh):
        return src.copy()
    scale_x, scale_y = sw / dw, sh / dh          # double, = 1 / inv_scale

    def axis(n_dst, n_src, scale, zero_frac_at_border):
        d = np.arange(n_dst, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if zero_frac_at_border:                    # x axis: fx = 0 when the tap would leave the row
            lo = s < 0
            f[lo] = 0; s[lo] = 0
            hi = s >= n_src - 1
            f[hi] = 0; s[hi] = n_src - 1
        c0 = _round_half_even_short((np.float32(1.0) - f) * np.float32(_COEF_ONE))
        c1 = _round_half_even_short(f * np.float32(_COEF_ONE))
        return s, c0, c1

    sx, ax0, ax1 = axis(dw, sw, scale_x, True)
    sy, by0, by1 = axis(dh, sh, scale_y, False)
    sx1 = np.minimum(sx + 1, sw - 1)
    s32 = src.astype(np.int32)
    rows = s32[:, sx, :] * ax0[None, :, None] + s32[:, sx1, :] * ax1[None, :, None]   # (sh, dw, C)
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    S0, S1 = rows[y0], rows[y1]
    out = (((by0[:, None, None] * (S0 >> 4)) >> 16) + ((by1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(h: int, w: int, imgsz: int, stride: int = 32, auto: bool = True):
    """LetterBox(new_shape=imgsz, auto=True, scaleup=True, center=True) geometry (Appendix B step 2).
    Returns (new_w, new_h, top, bottom, left, right)."""
    r = min(imgsz / h, imgsz / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    dw, dh = imgsz - new_w, imgsz - new_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_w, new_h, top, bottom, left, right


def letterbox(img: np.ndarray, imgsz: int, stride: int = 32, auto: bool = True) -> np.ndarray:
    h, w = img.shape[:2]
    new_w, new_h, top, bottom, left, right = letterbox_geometry(h, w, imgsz, stride, auto)
    if (w, h) != (new_w, new_h):
        img = resize_linear_u8(img, new_w, new_h)
    out = np.full((new_h + top + bottom, new_w + left + right, 3), 114, np.uint8)
    out[top:top + new_h, left:left + new_w] = img
    return out


def preprocess(img: np.ndarray, imgsz: int) -> torch.Tensor:
    """ndarray HWC uint8 (treated as BGR by Ultralytics) -> (1,3,H',W') float32 in [0,1], channel-flipped.
    NB under SAHI the array is really RGB (docs sahi/predict.py:103-106) — the flip is applied regardless."""
    lb = letterbox(img, imgsz)
    x = np.ascontiguousarray(lb[..., ::-1].transpose(2, 0, 1))
    return (torch.from_numpy(x).float() / 255).unsqueeze(0)


def nms_torchvision(boxes: np.ndarray, scores: np.ndarray, iou_thr: float) -> np.ndarray:
    """torchvision.ops.nms on CPU: order by score descending (ties: lower index first — stable), suppress IoU > thr.
    float32 arithmetic throughout."""
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    b = boxes.astype(np.float32)
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores.astype(np.float32), kind="stable")
    suppressed = np.zeros(n, bool)
    keep = []
    thr = np.float32(iou_thr)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1); h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return np.asarray(keep, np.int64)


def non_max_suppression(pred: torch.Tensor, conf_thres: float, iou_thres: float = 0.7, max_det: int = 300,
                        nc: int = 1, max_nms: int = 30000, max_wh: int = 7680) -> torch.Tensor:
    """ultralytics.utils.ops.non_max_suppression for ONE image; pred (4+nc+nm, A). Returns (n, 6+nm) rows
    [x1,y1,x2,y2,conf,cls,extra...] in kept order (score descending). Appendix B step 5."""
    mi = 4 + nc
    xc = pred[4:mi].amax(0) > conf_thres
    x = pred.transpose(0, 1).clone()
    cx, cy, w, h = x[:, 0].clone(), x[:, 1].clone(), x[:, 2].clone(), x[:, 3].clone()
    x[:, 0] = cx - w / 2; x[:, 1] = cy - h / 2; x[:, 2] = cx + w / 2; x[:, 3] = cy + h / 2
    x = x[xc]
    if x.shape[0] == 0:
        return x.new_zeros((0, 6 + pred.shape[0] - mi))
    box, cls, extra = x[:, :4], x[:, 4:mi], x[:, mi:]
    conf, j = cls.max(1, keepdim=True)
    x = torch.cat((box, conf, j.float(), extra), 1)[conf.view(-1) > conf_thres]
    n = x.shape[0]
    if n == 0:
        return x
    if n > max_nms:
        x = x[torch.from_numpy(np.argsort(-x[:, 4].numpy(), kind="stable")[:max_nms])]
    c = x[:, 5:6] * max_wh
    keep = nms_torchvision((x[:, :4] + c).numpy(), x[:, 4].numpy(), iou_thres)[:max_det]
    return x[torch.from_numpy(keep)]


def scale_boxes(img1_shape, boxes: torch.Tensor, img0_shape) -> torch.Tensor:
    """ultralytics.utils.ops.scale_boxes (padding=True): subtract pad, divide by gain, clip to the source image."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad_x = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1)
    pad_y = round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    b = boxes.clone()
    b[:, [0, 2]] -= pad_x
    b[:, [1, 3]] -= pad_y
    b[:, :4] /= gain
    b[:, [0, 2]] = b[:, [0, 2]].clamp(0, img0_shape[1])
    b[:, [1, 3]] = b[:, [1, 3]].clamp(0, img0_shape[0])
    return b


def scale_coords(img1_shape, coords: torch.Tensor, img0_shape) -> torch.Tensor:
    """ultralytics.utils.ops.scale_coords (normalize=False, padding=True) on (n, K, 3) keypoints."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad_x = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1)
    pad_y = round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    c = coords.clone()
    c[..., 0] -= pad_x
    c[..., 1] -= pad_y
    c[..., 0] /= gain
    c[..., 1] /= gain
    c[..., 0] = c[..., 0].clamp(0, img0_shape[1])
    c[..., 1] = c[..., 1].clamp(0, img0_shape[0])
    return c


class PredictResult:
    """The fields of an Ultralytics `Results` the reference reads (utils/yolo_wrapper.py:120-162)."""
    def __init__(self, xyxy, conf, cls, kpts):
        self.xyxy, self.conf, self.cls, self.kpts = xyxy, conf, cls, kpts

    def __len__(self):
        return int(self.xyxy.shape[0])


@torch.no_grad()
def predict(model: Yolo11PoseRef, img: np.ndarray, imgsz: int, conf: float, iou: float = 0.7, max_det: int = 300,
            round_boxes: bool = False) -> PredictResult:
    """One `model.predict(source=ndarray)` call. `round_boxes` selects the older PosePredictor semantic
    (`scale_boxes(...).round()`, ultralytics < ~8.3.40); Appendix B pins False."""
    x = preprocess(img, imgsz)
    out = model.forward(x)[0]
    det = non_max_suppression(out, conf, iou, max_det, nc=model.nc)
    n = det.shape[0]
    boxes = scale_boxes(x.shape[2:], det[:, :4], img.shape[:2])
    if round_boxes:
        boxes = boxes.round()
    kp = det[:, 6:].reshape(n, *model.kpt_shape)
    kp = scale_coords(x.shape[2:], kp, img.shape[:2])
    return PredictResult(boxes.numpy(), det[:, 4].numpy(), det[:, 5].numpy(), kp.numpy())
