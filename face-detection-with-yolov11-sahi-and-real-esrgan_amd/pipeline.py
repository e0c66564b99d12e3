"""Device-resident frame pipeline: SAHI slice -> batched YOLO11-pose -> per-slice NMS -> (RCCL all-gather) -> SAHI merge
-> face crops -> Real-ESRGAN x4. One process per GPU; torch supplies device memory and `torch.distributed`
(backend "nccl" = RCCL on ROCm) — every operator runs in libffp.so.

Multi-GPU: a step processes a batch of B frames stacked into one tall "super-frame" (frames cannot interact in the
merge because their boxes never overlap). The work items — every slice of every frame, each followed by that
frame's full-frame pass, in SAHI's own order (docs sahi/predict.py:270-314) — are split contiguously over the ranks,
each rank writes fixed-cap detections [items_per_rank][max_det][stride] and the ranks exchange them with ONE
all-gather (1.5 MB per 61 items): the only data-path collective. Frame f is then merged and its crops enhanced by rank
f % world.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib


@dataclass
class PipeConfig:
    slice_h: int = 512
    slice_w: int = 512
    overlap: float = 0.2
    imgsz: int = 512
    conf: float = 0.5
    iou: float = 0.7
    max_det: int = 300
    pp_type: str = "GREEDYNMM"
    pp_metric: str = "IOS"
    pp_thr: float = 0.5
    class_agnostic: bool = False
    perform_standard_pred: bool = True
    chan_order: int = _lib.CHAN_AS_BGR
    sr_crops: int = 32            # crops enhanced per frame (0: detection only)
    merge_cap: int = 4096


def frame_items(H: int, W: int, cfg: PipeConfig, n_frames: int = 1) -> np.ndarray:
    """Work items of a super-frame of n_frames stacked (H, W) frames: per frame its slices then its full-frame pass."""
    sl = _lib.slice_bboxes(H, W, cfg.slice_h, cfg.slice_w, cfg.overlap, cfg.overlap)
    items = []
    for f in range(n_frames):
        s = sl.copy()
        s[:, 1] += f * H
        s[:, 3] += f * H
        items.append(s)
        if len(sl) > 1 and cfg.perform_standard_pred:
            items.append(np.asarray([[0, f * H, W, (f + 1) * H]], np.int32))
    return np.concatenate(items, 0).astype(np.int32)


def shard(n_items: int, rank: int, world: int):
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items), per


def crop_boxes_for_sr(rows: np.ndarray, H: int, W: int, n: int, sizes: Sequence[int], seed: int) -> np.ndarray:
    """The fixed SR workload of SURVEY.md §8(d): n square crops with the given sizes, centred on the top-scoring merged
    boxes when there are at least n of them, else at seeded positions; clamped to the frame."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 4), np.int32)
    use_dets = rows.shape[0] >= n
    for i in range(n):
        s = int(sizes[i])
        if use_dets:
            cx = int((rows[i, 0] + rows[i, 2]) / 2)
            cy = int((rows[i, 1] + rows[i, 3]) / 2)
        else:
            cx = int(rng.integers(s, W - s))
            cy = int(rng.integers(s, H - s))
        x0 = min(max(cx - s // 2, 0), W - s)
        y0 = min(max(cy - s // 2, 0), H - s)
        out[i] = (x0, y0, x0 + s, y0 + s)
    return out


def sr_crop_sizes(n: int, seed: int = 0) -> np.ndarray:
    """sizes from {24,32,48,64,96} with p = {.3,.3,.2,.15,.05}."""
    rng = np.random.default_rng(seed + 777)
    return rng.choice(np.asarray([24, 32, 48, 64, 96]), size=n, p=[0.3, 0.3, 0.2, 0.15, 0.05])


def exchange_detections(local_dets, local_counts, world: int):
    """THE data-path collective: all-gather of the fixed-cap per-item detections and their counts (RCCL over xGMI with
    backend "nccl"; the same call runs on gloo for the CPU tests). Payload per rank: items_per_rank x max_det x stride
    fp32 (61 x 300 x 21 x 4 B = 1.5 MB) — latency-bound, so ONE collective per batch rather than one per frame."""
    if world == 1:
        return local_dets, local_counts
    import torch
    import torch.distributed as dist
    dev = local_dets.device
    if dist.get_backend() == "gloo" and dev.type == "cuda":      # rehearsal on a box without RCCL peers: stage through the host
        g, gc = exchange_detections(local_dets.cpu(), local_counts.cpu(), world)
        return g.to(dev), gc.to(dev)
    g = torch.empty((world * local_dets.shape[0],) + tuple(local_dets.shape[1:]), dtype=local_dets.dtype, device=dev)
    gc = torch.empty((world * local_counts.shape[0],), dtype=local_counts.dtype, device=dev)
    dist.all_gather_into_tensor(g, local_dets.contiguous())
    dist.all_gather_into_tensor(gc, local_counts.contiguous())
    return g, gc


class FramePipeline:
    """One rank's share of the pipeline. `torch` is imported lazily: it provides device tensors and the collective."""

    def __init__(self, det_weights, sr_weights, cfg: PipeConfig, arch: str = "s", device: int = 0,
                 det_precision: int = _lib.PREC_F32, sr_half: bool = True, rank: int = 0, world: int = 1):
        import torch
        self.torch = torch
        self.cfg, self.rank, self.world, self.device = cfg, rank, world, device
        self.dev = torch.device("cuda", device)
        self.det = _lib.Detector(det_weights, arch=arch, device=device, precision=det_precision)
        self.sr = _lib.Enhancer(sr_weights, 4, 23, device=device, half=sr_half) if (cfg.sr_crops > 0 and sr_weights is not None) else None
        self.stride = self.det.stride
        self._bufs = {}

    def _buf(self, name, shape, dtype):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = self.torch.zeros(shape, dtype=dtype, device=self.dev)
            self.torch.cuda.synchronize(self.dev)      # libffp runs on its own stream: the fill must have landed
            self._bufs[name] = t
        return t

    def detect(self, d_frame, H: int, W: int, n_frames: int = 1):
        """d_frame: uint8 cuda tensor (n_frames*H, W, 3). Returns (gathered dets [n_items_pad][max_det][stride], counts, items)."""
        torch, cfg = self.torch, self.cfg
        items = frame_items(H, W, cfg, n_frames)
        n_items = items.shape[0]
        lo, hi, per = shard(n_items, self.rank, self.world)
        local = self._buf("local_dets", (per, cfg.max_det, self.stride), torch.float32)
        lcount = self._buf("local_counts", (per,), torch.int32)   # entries past this rank's share stay 0
        if hi > lo:
            t = np.ascontiguousarray(items[lo:hi])
            _lib._check(_lib.lib().ffp_det_infer_tiles_dev(self.det.handle, d_frame.data_ptr(), H * n_frames, W, cfg.chan_order, _lib._ip(t),
                                                           hi - lo, cfg.imgsz, cfg.conf, cfg.iou, cfg.max_det, 0, local.data_ptr(),
                                                           lcount.data_ptr()))
            self._truncate_shift(local, lcount, hi - lo, H * n_frames, W)
        if self.world > 1:
            g, gc = exchange_detections(local, lcount, self.world)
            torch.cuda.synchronize(self.dev)       # libffp's stream must see the gathered boxes
            return g, gc, items
        return local, lcount, items

    def _truncate_shift(self, dets, counts, n, H, W):
        # wrapper + SAHI shift semantics on the device (utils/yolo_wrapper.py:137-162, docs sahi/prediction.py:94-120)
        _lib._check(_lib.lib().ffp_det_truncate_shift_dev(self.det.handle, dets.data_ptr(), counts.data_ptr(), n, self.cfg.max_det, H, W))

    def merge_frame(self, dets, counts, first_item: int, n_items: int):
        """SAHI merge of items [first_item, first_item+n_items) of the gathered buffer -> device rows + count tensor."""
        torch, cfg = self.torch, self.cfg
        out = self._buf("merged", (cfg.merge_cap, self.stride), torch.float32)
        outn = self._buf("merged_n", (1,), torch.int32)
        d = dets[first_item:first_item + n_items]
        c = counts[first_item:first_item + n_items]
        _lib._check(_lib.lib().ffp_merge_dev(self.det.handle, d.data_ptr(), c.data_ptr(), n_items, cfg.max_det, _lib.PP_TYPES[cfg.pp_type],
                                             _lib.METRICS[cfg.pp_metric], float(cfg.pp_thr), int(cfg.class_agnostic), out.data_ptr(),
                                             cfg.merge_cap, outn.data_ptr()))
        return out, outn

    def enhance_crops(self, d_frame_bgr, H: int, W: int, boxes: np.ndarray, wait: bool = True, slot: int = 0):
        """Real-ESRGAN x4 on crops of a resident BGR frame. Returns (uint8 cuda tensor with all outputs, offsets).
        wait=False enqueues on the enhancer's stream and returns; call wait_sr() before reading the tensor."""
        torch = self.torch
        n = boxes.shape[0]
        tot = int(sum(((int(b[3] - b[1]) * 4) * (int(b[2] - b[0]) * 4) * 3 + 15) // 16 * 16 for b in boxes))
        out = self._buf(f"sr_out{slot}", (tot,), torch.uint8)
        offs = np.zeros(n + 1, np.int64)
        b = np.ascontiguousarray(boxes, np.int32)
        fn = _lib.lib().ffp_sr_enhance_crops_dev if wait else _lib.lib().ffp_sr_enhance_crops_dev_async
        _lib._check(fn(self.sr.handle, d_frame_bgr.data_ptr(), H, W, _lib._ip(b), n, out.data_ptr(), tot,
                                                        offs.ctypes.data_as(C.POINTER(C.c_int64))))
        return out, offs

    def enhance_crops_multi(self, d_frames, H: int, W: int, boxes_per_frame, slot: int = 0):
        """Crops of several resident frames as ONE ragged SR batch (async; wait_sr() before reading)."""
        torch = self.torch
        boxes = np.ascontiguousarray(np.concatenate(boxes_per_frame, 0), np.int32)
        fidx = np.ascontiguousarray(np.concatenate([np.full(len(b), i, np.int32) for i, b in enumerate(boxes_per_frame)]))
        n = boxes.shape[0]
        tot = int(sum(((int(b[3] - b[1]) * 4) * (int(b[2] - b[0]) * 4) * 3 + 15) // 16 * 16 for b in boxes))
        out = self._buf(f"sr_out{slot}", (tot,), torch.uint8)
        offs = np.zeros(n + 1, np.int64)
        ptrs = (C.c_void_p * len(d_frames))(*[t.data_ptr() for t in d_frames])
        _lib._check(_lib.lib().ffp_sr_enhance_crops_multi_dev_async(self.sr.handle, len(d_frames), ptrs, _lib._ip(fidx), H, W, _lib._ip(boxes), n,
                                                                    out.data_ptr(), tot, offs.ctypes.data_as(C.POINTER(C.c_int64))))
        return out, offs

    def enhance_frame(self, d_frame_bgr, H: int, W: int, enhancer=None, tile: int = 400, tile_pad: int = 10, pre_pad: int = 0):
        """RealESRGANer.enhance of a whole resident BGR frame (tiled like FaceEnhancer does, utils/enhancer.py:138-156) ->
        resident BGR frame (scale*H, scale*W, 3). `enhancer`: an `_lib.Enhancer` (e.g. the x2plus model) or the pipeline's own."""
        sr = enhancer if enhancer is not None else self.sr
        s = sr.scale
        out = self._buf(f"sr_frame{s}", (H * s, W * s, 3), self.torch.uint8)
        _lib._check(_lib.lib().ffp_sr_enhance_dev(sr.handle, d_frame_bgr.data_ptr(), H, W, tile, tile_pad, pre_pad, out.data_ptr()))
        return out

    def enhance_first(self, d_frame_bgr, H: int, W: int, enhancer=None, tile: int = 400, tile_pad: int = 10):
        """The reference's enhance-first ordering (pipeline_v4_yolo/app_yolo_full.py:87-123): super-resolve the whole picture,
        then sliced detection + merge on the enhanced picture (boxes are in enhanced coordinates, as the reference draws them;
        eval/eval_dual.py:262-265 divides by the scale). Needs cfg.chan_order == CHAN_AS_BGR (the enhancer's output order).
        Returns (enhanced frame, merged rows, count) — all resident."""
        enh = self.enhance_frame(d_frame_bgr, H, W, enhancer, tile, tile_pad)
        He, We = int(enh.shape[0]), int(enh.shape[1])
        dets, counts, items = self.detect(enh, He, We, 1)
        rows, n = self.merge_frame(dets, counts, 0, items.shape[0])
        return enh, rows, n

    def wait_sr(self):
        _lib._check(_lib.lib().ffp_sr_wait(self.sr.handle))
