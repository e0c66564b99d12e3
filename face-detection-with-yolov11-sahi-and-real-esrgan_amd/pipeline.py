"""Device-resident frame pipeline: SAHI slice -> batched YOLO11-pose -> per-slice NMS -> (RCCL all-gather) -> SAHI merge
-> face crops -> Real-ESRGAN x4. One process per GPU; torch supplies device memory and `torch.distributed`
(backend "nccl" = RCCL on ROCm) — every operator runs in libffp.so.

Multi-GPU: a step processes a batch of B frames stacked into one tall "super-frame" (frames cannot interact in the
merge because their boxes never overlap). The work items — every slice of every frame, each followed by that
frame's full-frame pass, in SAHI's own order (docs sahi/predict.py:270-314) — are split into contiguous, cost-balanced
blocks over the ranks (`partition`: cost = pixels entering the network), each rank writes fixed-cap detections
[slots_per_rank][max_det][stride] and, when a frame's items are spread over ranks (the north_star's split: the slices of
ONE image across the GPUs), the ranks exchange them with ONE all-gather (1.5 MB per 61 items): the only data-path
collective; every rank then runs the identical deterministic merge. When every frame lives on one rank (weak scaling with
whole frames per rank) no rank needs another's boxes and the exchange is skipped. SR crops are placed by LPT (`lpt_assign`).
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
    sr_tile: int = 400            # FaceEnhancer's tile / tile_pad (utils/enhancer.py:21,135-142): crops larger than the tile are tiled
    sr_tile_pad: int = 10
    merge_cap: int = 4096         # rows of the merged-detections buffer; merge_frame raises when a frame yields more


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


def item_costs(items: np.ndarray, imgsz: int) -> np.ndarray:
    """Relative cost of each work item = pixels entering the network (LetterBox geometry of the item at `imgsz`;
    imgsz <= 0: native). A 4K full-frame pass at 512 is 288x512 = 0.56 of a 512x512 slice."""
    out = np.zeros(len(items), np.float64)
    cache = {}
    for k, (x0, y0, x1, y1) in enumerate(items):
        h, w = int(y1 - y0), int(x1 - x0)
        if (h, w) not in cache:
            sz = imgsz if imgsz > 0 else (max(h, w) + 31) // 32 * 32
            nw, nh, t, b, l, r = _lib.letterbox_geometry(h, w, sz)
            cache[(h, w)] = float((nh + t + b) * (nw + l + r))
        out[k] = cache[(h, w)]
    return out


def partition(costs: Sequence[float], world: int):
    """Contiguous partition of the item list into `world` blocks minimising the heaviest block (contiguous by
    row-major index so that the gathered buffer is in SAHI's own order). Returns [(lo, hi)] per rank; a rank may be empty
    only when there are fewer items than ranks."""
    c = np.asarray(costs, np.float64)
    n = len(c)
    if world <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (max(world, 1) - 1)

    def blocks(limit):
        out, lo, acc = [], 0, 0.0
        for i in range(n):
            if acc + c[i] > limit and i > lo:
                out.append((lo, i))
                lo, acc = i, 0.0
            acc += c[i]
        out.append((lo, n))
        return out

    lo_t, hi_t = float(c.max()), float(c.sum())
    for _ in range(60):
        mid = 0.5 * (lo_t + hi_t)
        if len(blocks(mid)) <= world:
            hi_t = mid
        else:
            lo_t = mid
    b = blocks(hi_t * (1 + 1e-12))
    while len(b) < min(world, n):                       # fewer blocks than ranks: split the heaviest splittable block
        k = max((i for i in range(len(b)) if b[i][1] - b[i][0] > 1), key=lambda i: c[b[i][0]:b[i][1]].sum())
        lo, hi = b[k]
        cs = np.cumsum(c[lo:hi])
        m = lo + 1 + int(np.argmin(np.abs(cs[:-1] - cs[-1] / 2)))
        b[k:k + 1] = [(lo, m), (m, hi)]
    return b + [(n, n)] * (world - len(b))


def shard(n_items: int, rank: int, world: int):
    """Equal-cost items: (lo, hi, slots per rank) of rank's contiguous share."""
    b = partition(np.ones(n_items), world)
    return b[rank][0], b[rank][1], max(h - l for l, h in b)


class Layout:
    """Where every work item of a super-frame lives: global order `items`, per-rank shares, and the slot of an item in the fixed-cap
    exchange buffer (unused slots keep count 0, so any slot range is still in SAHI's order).

    Two partitions. Default: ONE contiguous cost-balanced split of the whole item list (rank r's k-th item sits at slot r * per + k);
    with one frame per rank every frame lands on one rank and nothing has to be exchanged. `spread=True` (the north_star's split,
    strong scaling): EVERY frame's items are cut into `world` contiguous cost-balanced blocks and rank r takes block r of every
    frame, so a group of frames costs ONE all-gather; the gathered buffer, reordered frame-major ([frame][rank][per]), holds each
    frame's items contiguously and in SAHI's order."""

    def __init__(self, items: np.ndarray, n_frames: int, world: int, costs: np.ndarray, spread: bool = False):
        self.items, self.n_frames, self.world = items, n_frames, world
        self.ipf = len(items) // n_frames
        self.spread = bool(spread) and world > 1
        if self.spread:
            self.frame_bounds = partition(costs[:self.ipf], world)           # frames of a group have the same items: one split serves all
            self.per = max(1, max(h - l for l, h in self.frame_bounds))
            self.bounds = None
        else:
            self.bounds = partition(costs, world)
            self.per = max(1, max(h - l for l, h in self.bounds))

    def local_items(self, rank: int) -> np.ndarray:
        """This rank's items in the order it runs them (spread: block `rank` of frame 0, of frame 1, ...)."""
        if self.spread:
            lo, hi = self.frame_bounds[rank]
            return np.concatenate([self.items[f * self.ipf + lo:f * self.ipf + hi] for f in range(self.n_frames)], 0).reshape(-1, 4)
        lo, hi = self.bounds[rank]
        return self.items[lo:hi]

    def local_slots(self) -> int:
        """Rows of a rank's fixed-cap exchange buffer."""
        return self.n_frames * self.per if self.spread else self.per

    def rank_of(self, i: int) -> int:
        b = self.frame_bounds if self.spread else self.bounds
        j = i % self.ipf if self.spread else i
        for r, (lo, hi) in enumerate(b):
            if lo <= j < hi:
                return r
        raise IndexError(i)

    def slot(self, i: int) -> int:
        """Slot of item i in the gathered buffer (spread: after the frame-major reorder)."""
        r = self.rank_of(i)
        if self.spread:
            f, j = divmod(i, self.ipf)
            return (f * self.world + r) * self.per + (j - self.frame_bounds[r][0])
        return r * self.per + (i - self.bounds[r][0])

    def owner(self, f: int) -> int:
        """The rank holding ALL items of frame f, or -1 when the frame is spread over ranks."""
        if self.spread:
            return -1
        a, b = f * self.ipf, (f + 1) * self.ipf
        r = self.rank_of(a)
        return r if self.bounds[r][1] >= b else -1

    @property
    def aligned(self) -> bool:
        """Every frame lives on one rank: no detection of one rank is needed by another, the exchange can be skipped."""
        return all(self.owner(f) >= 0 for f in range(self.n_frames))

    def frame_slots(self, f: int, gathered: bool, rank: int = 0):
        """(first slot, slot count) of frame f in the gathered buffer, or in rank's local buffer when nothing was exchanged."""
        a, b = f * self.ipf, (f + 1) * self.ipf
        if self.spread:
            assert gathered, "a spread frame needs the exchange"
            return f * self.world * self.per, self.world * self.per
        if gathered:
            s0, s1 = self.slot(a), self.slot(b - 1)
            return s0, s1 - s0 + 1
        lo, hi = self.bounds[rank]
        assert lo <= a and b <= hi, "frame is not local to this rank"
        return a - lo, b - a

    def rows_needed(self, rank: int, H: int):
        """Row range of the super-frame rank must have resident: the rows its items read, the frames it owns (merge + crops) and
        every frame that is spread over ranks (its merge is replicated and its crops are placed on all ranks by LPT)."""
        if self.spread:
            return 0, self.n_frames * H
        lo, hi = self.bounds[rank]
        ys = [(int(self.items[i][1]), int(self.items[i][3])) for i in range(lo, hi)]
        ys += [(f * H, (f + 1) * H) for f in range(self.n_frames) if self.owner(f) in (rank, -1)]
        if not ys:
            return 0, 0
        return min(y[0] for y in ys), max(y[1] for y in ys)


def lpt_assign(weights: Sequence[float], world: int) -> np.ndarray:
    """Longest-processing-time placement (SR crops): heaviest first, each to the least-loaded rank (ties: lowest
    rank, lower index first). Deterministic, so every rank derives the same placement from the same merged boxes."""
    w = np.asarray(weights, np.float64)
    order = sorted(range(len(w)), key=lambda i: (-w[i], i))
    load = np.zeros(max(world, 1))
    out = np.zeros(len(w), np.int32)
    for i in order:
        r = int(np.argmin(load))
        out[i] = r
        load[r] += w[i]
    return out


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


def frame_major(g, gc, world: int, n_frames: int, per: int):
    """Gathered [rank][frame][per] -> [frame][rank][per]: every frame's slots contiguous and in SAHI's order (Layout spread=True)."""
    tail = tuple(g.shape[1:])
    g2 = g.view(world, n_frames, per, *tail).transpose(0, 1).contiguous().view(n_frames * world * per, *tail)
    gc2 = gc.view(world, n_frames, per).transpose(0, 1).contiguous().view(n_frames * world * per)
    return g2, gc2


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
        if self.sr is None:
            self.det.set_lanes(1)       # detection only (BASELINE config 2): nothing runs beside the detector, let its branches overlap
        self.stride = self.det.stride
        self._bufs = {}
        self._layouts = {}

    def _buf(self, name, shape, dtype):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = self.torch.zeros(shape, dtype=dtype, device=self.dev)
            self.torch.cuda.synchronize(self.dev)      # libffp runs on its own stream: the fill must have landed
            self._bufs[name] = t
        return t

    def _bytes(self, name, nbytes: int):
        """Grow-only uint8 scratch (crop outputs change size every frame: no per-frame allocation or fill)."""
        t = self._bufs.get(name)
        if t is None or t.numel() < nbytes:
            t = self.torch.empty((max(int(nbytes * 1.5), 1 << 20),), dtype=self.torch.uint8, device=self.dev)
            self.torch.cuda.synchronize(self.dev)
            self._bufs[name] = t
        return t[:max(nbytes, 1)]

    @staticmethod
    def sr_out_bytes(boxes: np.ndarray, H: int, W: int, scale: int = 4) -> int:
        """Bytes ffp_sr_enhance_crops_* packs for these boxes (int box clamped to the frame, 16-byte aligned entries)."""
        tot = 0
        for b in boxes:
            w = min(W, int(b[2])) - max(0, int(b[0]))
            h = min(H, int(b[3])) - max(0, int(b[1]))
            if w > 0 and h > 0:
                tot += (h * scale * w * scale * 3 + 15) // 16 * 16
        return tot

    def merged_count(self, outn) -> int:
        """Host value of merge_frame's count; raises when the frame produced more merged detections than merge_cap rows."""
        n = int(outn.item())
        if n > self.cfg.merge_cap:
            raise _lib.FfpError(1, f"{n} merged detections exceed merge_cap={self.cfg.merge_cap}: raise PipeConfig.merge_cap")
        return n

    def layout(self, H: int, W: int, n_frames: int = 1, mode: str = "global") -> Layout:
        """mode "global": one contiguous split of all items over the ranks; "spread": every frame's items over the ranks (one
        all-gather per call); "local": this rank runs all n_frames frames by itself (weak scaling: nothing is shared)."""
        key = (H, W, n_frames, mode)
        L = self._layouts.get(key)
        if L is None:
            items = frame_items(H, W, self.cfg, n_frames)
            L = Layout(items, n_frames, 1 if mode == "local" else self.world, item_costs(items, self.cfg.imgsz), spread=(mode == "spread"))
            self._layouts[key] = L
        return L

    def detect(self, d_frame, H: int, W: int, n_frames: int = 1, exchange: str = "auto", mode: str = "global"):
        """d_frame: uint8 cuda tensor (n_frames*H, W, 3); only layout.rows_needed(rank) have to be valid.
        Returns (dets [slots][max_det][stride], counts [slots], layout, gathered). exchange: "auto" skips the all-gather when every
        frame's items live on one rank (weak scaling with whole frames per rank), "always" / "never" force it. mode: see layout()."""
        import time
        torch, cfg = self.torch, self.cfg
        L = self.layout(H, W, n_frames, mode)
        rank = 0 if mode == "local" else self.rank
        world = 1 if mode == "local" else self.world
        t = np.ascontiguousarray(L.local_items(rank))
        n_loc = len(t)
        slots = L.local_slots()
        local = self._buf("local_dets", (slots, cfg.max_det, self.stride), torch.float32)
        lcount = self._buf("local_counts", (slots,), torch.int32)   # entries past this rank's share stay 0
        t0 = time.perf_counter()
        if n_loc:
            dense = L.spread and n_loc != slots                       # blocks smaller than `per`: run dense, then pad per frame
            dst = self._buf("dense_dets", (n_loc, cfg.max_det, self.stride), torch.float32) if dense else local
            dcn = self._buf("dense_counts", (n_loc,), torch.int32) if dense else lcount
            _lib._check(_lib.lib().ffp_det_infer_tiles_dev(self.det.handle, d_frame.data_ptr(), H * n_frames, W, cfg.chan_order, _lib._ip(t),
                                                           n_loc, cfg.imgsz, cfg.conf, cfg.iou, cfg.max_det, 0, dst.data_ptr(), dcn.data_ptr()))
            self._truncate_shift(dst, dcn, n_loc, H * n_frames, W)
            if dense:
                b = n_loc // n_frames
                local.view(n_frames, L.per, cfg.max_det, self.stride)[:, :b].copy_(dst.view(n_frames, b, cfg.max_det, self.stride))
                lcount.view(n_frames, L.per)[:, :b].copy_(dcn.view(n_frames, b))
        if not L.spread and n_loc < slots:
            lcount[n_loc:].zero_()
            torch.cuda.synchronize(self.dev)
        self.t_detect = time.perf_counter() - t0
        self.t_exchange = 0.0
        need = world > 1 and (L.spread or exchange == "always" or (exchange == "auto" and not L.aligned))
        if need:
            t0 = time.perf_counter()
            g, gc = exchange_detections(local, lcount, world)
            if L.spread:
                g, gc = frame_major(g, gc, world, n_frames, L.per)
            # libffp's stream (the merge) must see the gathered boxes: stream-ordered through an event, the host does not wait
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            self.det.stream_wait_event(ev.cuda_event)
            self._keep = (g, gc, ev)                                   # alive until the next call
            self.t_exchange = time.perf_counter() - t0
            return g, gc, L, True
        return local, lcount, L, False

    def merge_frame_of(self, dets, counts, L: Layout, f: int, gathered: bool):
        s0, ns = L.frame_slots(f, gathered, 0 if L.world == 1 else self.rank)
        return self.merge_frame(dets, counts, s0, ns)

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
        tot = self.sr_out_bytes(boxes, H, W, self.sr.scale)
        out = self._bytes(f"sr_out{slot}", tot)
        offs = self.sr.enhance_crops_dev([d_frame_bgr.data_ptr()], H, W, boxes, out.data_ptr(), tot, None, self.cfg.sr_tile, self.cfg.sr_tile_pad, wait)
        return out, offs

    def enhance_crops_multi(self, d_frames, H: int, W: int, boxes_per_frame, slot: int = 0):
        """Crops of several resident frames as ONE ragged SR batch (async; wait_sr() before reading)."""
        boxes = np.ascontiguousarray(np.concatenate(boxes_per_frame, 0), np.int32)
        fidx = np.ascontiguousarray(np.concatenate([np.full(len(b), i, np.int32) for i, b in enumerate(boxes_per_frame)]))
        tot = self.sr_out_bytes(boxes, H, W, self.sr.scale)
        out = self._bytes(f"sr_out{slot}", tot)
        offs = self.sr.enhance_crops_dev([t.data_ptr() for t in d_frames], H, W, boxes, out.data_ptr(), tot, fidx, self.cfg.sr_tile,
                                         self.cfg.sr_tile_pad, wait=False)
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
        dets, counts, L, gathered = self.detect(enh, He, We, 1)
        rows, n = self.merge_frame_of(dets, counts, L, 0, gathered)
        return enh, rows, n

    def wait_sr(self):
        _lib._check(_lib.lib().ffp_sr_wait(self.sr.handle))
