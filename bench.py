#!/usr/bin/env python3
"""bench.py — end-to-end 4K frames/sec of the hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic 4K frames (batch = n_gpus frames, one frame at N=1), timed
over the span SURVEY.md §8(d) defines — from the decoded uint8 frame in (pinned) HOST memory to merged detections and
enhanced crops back in host memory: frame upload (copy stream, overlapped with the previous group's kernels), SAHI
512x512 / 0.2 slicing (60 slices + the full-frame pass), YOLO11s-pose on every item at net input 512, per-item NMS,
int-truncate + shift, [RCCL all-gather of the fixed-cap boxes when a frame's items are spread over ranks], SAHI
GREEDYNMM/IOS/0.5 merge, merged detections to host, Real-ESRGAN x4 on 32 crops per frame whose sizes change every frame,
enhanced crops to host. Rank 0 prints ONE JSON line (contract in the task statement) with `roofline`, `secondary`
(the reference-default image_size=1024, the eval merge NMS/IOS/agnostic, the exact-fp32 detector, fixed crop sizes, frames
already resident in HBM; at N>1 the forced all-gather and ONE frame's slices across the ranks) and, at N=1, `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense MFMA peaks (/opt/skills/guides/MI355X_MICROARCH.md); f32x3 spends three fp16 MFMAs per algorithmic product
PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3, "f32x3": 2500.0 / 3}
# what a bare MFMA loop on random operands sustains on this chip (tools/probes/mfma_peak.hip, profiles/r03_mfma_sustained_peak_probe.txt):
# the clock drops under matrix load (DVFS), so the 2.4 GHz figures above are not reachable by any kernel; context only, `frac` uses PEAK_TFLOPS
SUSTAINED_TFLOPS = {"f16": 1908.0, "f32x3": 1677.0 / 3}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--slice", type=int, default=512)
    ap.add_argument("--overlap", type=float, default=0.2)
    ap.add_argument("--imgsz", type=int, default=512, help="network input size (512 = native tile; 1024 = reference wrapper default)")
    ap.add_argument("--arch", default="s")
    ap.add_argument("--det-precision", default="f32x3", choices=["f32", "f32x3", "f16"],
                    help="f32: exact-fp32 MFMA; f32x3: fp32 storage, scaled fp16 hi/lo split products (fp32-grade); f16: speed mode")
    ap.add_argument("--sr-crops", type=int, default=32, help="crops enhanced per frame (0: config 2, detection only)")
    ap.add_argument("--sr-sizes", default="random-per-frame", choices=["random-per-frame", "fixed", "from-detections"],
                    help="crop sizes: SURVEY §8(d) law re-drawn for every frame (what a real stream looks like to the enhancer), "
                         "the same multiset every frame, or the merged boxes themselves")
    ap.add_argument("--pp-type", default="GREEDYNMM", choices=["GREEDYNMM", "NMS"])
    ap.add_argument("--class-agnostic", action="store_true", help="merge across classes (the reference's eval setting with NMS)")
    ap.add_argument("--conf", type=float, default=0.5)
    ap.add_argument("--distinct-frames", type=int, default=2)
    ap.add_argument("--frames-per-step", type=int, default=0, help="frames per step over ALL ranks (0: one per rank = weak scaling; "
                                                                    "1 with --gpus N: ONE frame's slices across the ranks = strong scaling)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "always", "never"],
                    help="all-gather of the per-item detections: auto = only when a frame's items are spread over ranks")
    ap.add_argument("--resident", action="store_true", help="frames already in HBM when the timed region starts (no upload in the span)")
    ap.add_argument("--sr-batch-frames", type=int, default=10,
                    help="frames whose crops are enhanced together as one ragged Real-ESRGAN batch (1: per frame)")
    ap.add_argument("--det-batch-frames", type=int, default=5,
                    help="consecutive steps whose frames are detected together as one ragged batch of slices (1: per step)")
    ap.add_argument("--lanes", type=int, default=2,
                    help="pipeline lanes per process: L detector + enhancer handle pairs driven by L host threads, detection groups dealt round-robin "
                         "(their kernels interleave on the card; secondary.one_lane is the same loop on one lane). The kernel events behind `roofline` are "
                         "taken on ONE lane in an extra profiled pass after the timed region, with the card to itself. Only where the ranks run independent "
                         "frames (no collective in the loop): other layouts run one lane")
    ap.add_argument("--conv-totals", action="store_true",
                    help="count launches, algorithmic FLOPs and bytes of every conv launch of the process (ffp_conv_totals_*) and print them as \"conv_totals\": "
                         "what tools/pmc_traffic.py needs beside a rocprofv3 --pmc pass of this same command")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", action="store_true", help="CPU baseline on a bounded sample (3 slices + 3 crops) instead of one whole frame")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--secondary-steps", type=int, default=20)
    ap.add_argument("--secondary-only", default="", help="comma-separated names: measure only these secondary rows (experiments)")
    ap.add_argument("--sr-exclusive", action="store_true", help="experiment: wait for every SR batch before the next detection group (no overlap of the two streams)")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed run: the pipelined loop's outputs (merged rows, crop boxes, enhanced crops; JPEG files with jpeg_io) over 25 steps against "
                         "synchronous calls on fresh buffers, byte for byte, with 1 lane, 2 lanes and once with JPEG decode/encode in the span -> \"verify\" in the JSON line")
    return ap.parse_args(argv)


def _cpu_budget():
    """Host cores this process may actually use (cgroup quota of the GPU box, else the affinity mask)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(args, det_w, sr_w, frame, crop_boxes, premerge_rows, budget_s=45.0):
    """The oracle (CPU restatement, torch fp32) timed on the GPU box's host cores over ONE WHOLE FRAME of the same workload, in the
    reference's own order: one slice at a time (docs sahi/predict.py:270-298), the full-frame pass, the merge, one crop at a time
    (utils/enhancer.py:344-391). Bounded: each of the two loops (slices, crops) stops after `budget_s` seconds once it has done at
    least 3 units and is scaled to the frame — on a box where a frame costs ~70 s the sample IS the whole frame; the sample string
    says what was run. --cpu-sample times 3 slices + the full-frame pass + 3 crops and scales."""
    import torch
    from oracle import rrdbnet_ref, sahi_ref, ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    cores = _cpu_budget()
    torch.set_num_threads(cores)                         # more threads than the cgroup grants only adds contention
    H, W = frame.shape[:2]
    ref = Yolo11PoseRef(det_w, args.arch)
    boxes = sahi_ref.get_slice_bboxes(H, W, args.slice, args.slice, args.overlap, args.overlap)
    idx = np.linspace(0, len(boxes) - 1, 3).astype(int) if args.cpu_sample else np.arange(len(boxes))
    ultra_post.predict(ref, frame[:args.slice, :args.slice], args.imgsz, args.conf)   # warm-up (scripts/inference_time.py:46-52)
    t0 = time.perf_counter()
    done = 0
    for i in idx:
        x0, y0, x1, y1 = boxes[i]
        ultra_post.predict(ref, frame[y0:y1, x0:x1], args.imgsz, args.conf)
        done += 1
        if done >= 3 and time.perf_counter() - t0 > budget_s:
            break
    t_slices = (time.perf_counter() - t0) / done * len(boxes)
    t0 = time.perf_counter()
    ultra_post.predict(ref, frame, args.imgsz, args.conf)
    t_full = time.perf_counter() - t0
    t0 = time.perf_counter()
    dets = [sahi_ref.Det(r[:4].tolist(), r[4], int(r[5])) for r in premerge_rows]
    if len(dets) > 1:
        sahi_ref.postprocess(dets, args.pp_type, "IOS", 0.5, False)
    t_merge = time.perf_counter() - t0
    t_sr_frame, sr_sample = 0.0, "none"
    if args.sr_crops > 0 and len(crop_boxes):
        net = rrdbnet_ref.RRDBNetRef(sr_w, 4, 23)
        sample = ([b for b in crop_boxes if (b[2] - b[0]) <= 48][:3] or [crop_boxes[0]]) if args.cpu_sample else list(crop_boxes)
        px = n_done = 0
        t0 = time.perf_counter()
        for b in sample:
            rrdbnet_ref.enhance(net, frame[b[1]:b[3], b[0]:b[2]][..., ::-1].copy())
            px += int(b[2] - b[0]) * int(b[3] - b[1])
            n_done += 1
            if n_done >= 3 and time.perf_counter() - t0 > budget_s:
                break
        t_sr = time.perf_counter() - t0
        tot_px = sum(int(b[2] - b[0]) * int(b[3] - b[1]) for b in crop_boxes)
        t_sr_frame = t_sr / px * tot_px
        sr_sample = f"{n_done} of {len(crop_boxes)} crops ({px} px of {tot_px})"
    t_frame = t_slices + t_full + t_merge + t_sr_frame
    whole = done == len(boxes) and (args.sr_crops == 0 or not len(crop_boxes) or n_done == len(crop_boxes))
    what = "one whole frame" if whole else f"time-bounded sample ({budget_s:.0f} s per loop) scaled to one frame"
    return {"value": 1.0 / t_frame, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{what} (frame 0 of the timed loop, the crop boxes the GPU enhanced for it; one frame, not three): "
                      f"{done} of {len(boxes)} slices + full-frame pass + merge of {len(dets)} boxes + SR on {sr_sample} "
                      f"({t_frame:.1f} s/frame: det {t_slices + t_full:.1f}, merge {t_merge:.3f}, sr {t_sr_frame:.1f})"}


class Runner:
    """One configuration of the hot path: its pipeline, its resident plans, its timed loop."""

    def __init__(self, args, ctx, *, det_precision=None, imgsz=None, pp_type=None, class_agnostic=None, sr_sizes=None, frames_per_step=None,
                 exchange=None, resident=None, pipe=None, pipes=None, det_batch=None, sr_batch=None, jpeg_io=False, lanes=None, workload=None,
                 verify=False, _lane=0, _n_lanes=None):
        import torch
        kw = dict(det_precision=det_precision, imgsz=imgsz, pp_type=pp_type, class_agnostic=class_agnostic, sr_sizes=sr_sizes, frames_per_step=frames_per_step,
                  exchange=exchange, resident=resident, det_batch=det_batch, sr_batch=sr_batch, jpeg_io=jpeg_io, workload=workload, verify=verify)
        if workload:                                  # another BASELINE config: its own frame size / tiling / crop count (and its own synthetic frames, in ctx)
            args = argparse.Namespace(**{**vars(args), **workload})
        self.verify = verify
        self.rec = {}                                 # verify: per frame id -> what the pipelined loop produced for it
        self.own_pipe = False
        if pipes is None and pipe is not None:
            pipes = [pipe]
        pipe = pipes[_lane] if pipes is not None and _lane < len(pipes) else None
        from ffp_amd import _lib, pipeline, synth
        self.torch, self.pipeline, self.ctx, self.args = torch, pipeline, ctx, args
        self.rank, self.world, self.dev = ctx["rank"], ctx["world"], ctx["dev"]
        self.det_precision = det_precision or args.det_precision
        self.imgsz = imgsz or args.imgsz
        self.pp_type = pp_type or args.pp_type
        self.class_agnostic = args.class_agnostic if class_agnostic is None else class_agnostic
        self.sr_sizes = sr_sizes or args.sr_sizes
        self.B = frames_per_step or args.frames_per_step or self.world
        self.exchange = exchange or args.exchange
        # how the frames of a step map to ranks (FramePipeline.layout): "local" = weak scaling, every rank runs its own frame by itself
        # and holds nothing of the others'; "spread" = the north_star split, EVERY frame's items over all ranks, one all-gather per
        # detection group (strong scaling); "global" = one contiguous split of all items (forced-exchange comparison row)
        if self.world == 1:
            self.mode = "local"
        elif self.B == 1:
            self.mode = "spread"
        elif self.B == self.world and self.exchange == "auto":
            self.mode = "local"
        else:
            self.mode = "global"
        self.Bl = self.B if self.mode == "global" else 1          # frames of one step in THIS rank's super-frame
        # lanes: this object is lane `lane` of `n_lanes`; lane 0 owns the others (self.sub) and runs them on threads. Detection groups are dealt
        # round-robin, so the lanes together process exactly the single-lane sequence of frames, crops and seeds. Never with a collective in
        # the loop (two threads of a rank would have to issue collectives in one order on every rank).
        self.lane = _lane
        self.n_lanes = _n_lanes if _n_lanes is not None else (max(1, lanes if lanes is not None else args.lanes) if self.mode == "local" else 1)
        self.sub = []
        self.resident = args.resident if resident is None else resident
        self.H, self.W = args.height, args.width
        self.DB = max(1, det_batch or args.det_batch_frames)
        self.SB = max(1, sr_batch or args.sr_batch_frames)
        self.cfg = pipeline.PipeConfig(slice_h=args.slice, slice_w=args.slice, overlap=args.overlap, imgsz=self.imgsz, conf=args.conf,
                                       pp_type=self.pp_type, class_agnostic=self.class_agnostic, sr_crops=args.sr_crops)
        if pipe is not None and pipe.det.precision == {"f16": _lib.PREC_F16, "f32x3": _lib.PREC_F32X3, "f32": _lib.PREC_F32}[self.det_precision]:
            self.pipe = pipe                      # same weights, same arithmetic: share the handles (plans are keyed by shape)
            self.pipe.cfg = self.cfg
            self.pipe._layouts = {}
        else:
            self.pipe = pipeline.FramePipeline(ctx["det_w"], ctx["sr_w"], self.cfg, arch=args.arch, device=ctx["local_rank"],
                                               det_precision={"f16": _lib.PREC_F16, "f32x3": _lib.PREC_F32X3, "f32": _lib.PREC_F32}[self.det_precision],
                                               sr_half=True, rank=self.rank, world=self.world)
            self.own_pipe = True
        self.items_per_frame = pipeline.frame_items(self.H, self.W, self.cfg, 1).shape[0]
        self.fixed_sizes = pipeline.sr_crop_sizes(max(args.sr_crops, 1), seed=0)
        self.host_rows = torch.empty((self.cfg.merge_cap, self.pipe.stride), dtype=torch.float32).pin_memory()
        self.host_sr = None
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        # jpeg_io: the reference's file boundaries inside the span (reported separately, never the headline) — every frame
        # arrives as a JPEG stream and is decoded straight into its device slot (host Huffman decoding on a thread pool of 8, the rest
        # on the device: no 24.9 MB upload), every SR batch leaves as JPEG files encoded from the device buffer (quality 95, like cv2.imwrite)
        self.jpeg_io = jpeg_io
        if jpeg_io:
            from concurrent.futures import ThreadPoolExecutor
            self._lib = _lib
            self.jpegs = [_lib.jpeg_encode(f, 95, bgr=False) for f in ctx["host_frames"]]
            self.pool = ThreadPoolExecutor(max_workers=8)
            self.jpeg_bytes_in = self.jpeg_bytes_out = 0
        # frame slots: a ring deep enough that a slot is never refilled while queued crops (held until SB frames are together) or the
        # SR batch in flight (whose crop gather reads the frames asynchronously on the enhancer's stream, include/ffp.h: d_frame stays
        # untouched until ffp_sr_wait) still read it; upload() also waits for the batch in flight if it ever meets its slot
        self.nslots = -(-self.SB // self.DB) + 3
        self.host_supers, self.slots = {}, {}
        self.state, self.pending, self.queue, self.group = {}, {}, [], {}
        self.sr_px = 0
        self.tm = {}                                   # host-side wall time per stage over the timed loop (seconds)
        self.lat_t0, self.lat = {}, []                 # per-frame latency: handed to the pipeline -> enhanced crops on the host
        if _n_lanes is None and self.n_lanes > 1:
            self.sub = [Runner(args, ctx, pipes=pipes, _lane=k, _n_lanes=self.n_lanes, **kw) for k in range(1, self.n_lanes)]
        self.pipes = [self.pipe] + [r.pipe for r in self.sub]
        self.prof_pipe = self.pipe                     # the lane whose last group / SR batch carries the kernel events (loop())

    def close(self):
        """Release what this configuration holds on the device and in pinned memory (a secondary row must not leave its plans behind:
        round 3's four-ranks-on-one-GPU rehearsal ran out of memory in the later rows)."""
        for r in self.sub:
            r.close()
        self.torch.cuda.synchronize(self.dev)
        if self.own_pipe:
            self.pipe.det.close()
            if self.pipe.sr is not None:
                self.pipe.sr.close()
            self.pipe._bufs.clear()
        self.slots.clear(); self.host_supers.clear(); self.pending.clear(); self.queue.clear(); self.group.clear(); self.rec.clear()
        self.host_sr = None
        self.torch.cuda.empty_cache()

    def mem(self):
        """Device bytes the lanes' handles hold: packed weights + resident plans (ffp_det_mem_bytes / ffp_sr_mem_bytes)."""
        out = {"det_weights": 0, "det_plans": 0, "det_plans_resident": 0, "sr_weights": 0, "sr_plans": 0, "sr_plans_resident": 0}
        for p in {id(q): q for q in self.pipes}.values():
            d = p.det.mem_bytes()
            out["det_weights"] += d["weights"]; out["det_plans"] += d["plans"]; out["det_plans_resident"] += d["plans_resident"]
            if p.sr is not None:
                e = p.sr.mem_bytes()
                out["sr_weights"] += e["weights"]; out["sr_plans"] += e["plans"]; out["sr_plans_resident"] += e["plans_resident"]
        return out

    # ---- frames: pinned host super-frames, three device slots per super-frame size, uploads on a copy stream ----------------
    def frame_of(self, variant, f):
        """Which synthetic frame sits at position f of a super-frame (ranks of a weak-scaling run take different ones)."""
        hf = self.ctx["host_frames"]
        return (variant + f + (self.rank if self.mode == "local" else 0)) % len(hf)

    def host_super(self, nf, variant):
        """Pinned host super-frame of THIS rank's frames only (weak scaling: a rank never holds another rank's frames)."""
        key = (nf, variant)
        if key not in self.host_supers:
            hf = self.ctx["host_frames"]
            sf = np.concatenate([hf[self.frame_of(variant, f)] for f in range(nf)], 0)
            self.host_supers[key] = self.torch.from_numpy(sf).pin_memory()
        return self.host_supers[key]

    def tick(self, name, t0):
        t1 = time.perf_counter()
        self.tm[name] = self.tm.get(name, 0.0) + (t1 - t0)
        return t1

    def upload(self, gi, nf):
        """Start the host->device copy of group gi's super-frame (only the rows this rank reads) on the copy stream."""
        torch = self.torch
        variant = gi % len(self.ctx["host_frames"])
        if self.resident:
            key = ("res", nf, variant)
            self.state["slot_key"] = key
            if key not in self.slots:
                self.slots[key] = self.host_super(nf, variant).to(self.dev)
                torch.cuda.synchronize(self.dev)
            return self.slots[key], None
        self.up_seq = getattr(self, "up_seq", -1) + 1          # this lane's own upload count (its groups are every n_lanes-th one): the ring position
        key = (nf, self.up_seq % self.nslots)
        if any(x[0] == key for x in self.queue):
            raise RuntimeError("frame slot ring too small: a slot with queued crops would be overwritten")
        if key in self.pending.get("slots", ()):
            self.drain_sr()                            # the SR batch in flight still gathers crops from this slot
        self.state["slot_key"] = key
        if key not in self.slots:
            self.slots[key] = torch.empty((nf * self.H, self.W, 3), dtype=torch.uint8, device=self.dev)
            torch.cuda.synchronize(self.dev)
        if self.jpeg_io:
            slot, hf = self.slots[key], self.ctx["host_frames"]
            fb = self.H * self.W * 3
            futs = [self.pool.submit(self._lib.jpeg_decode_dev, self.jpegs[self.frame_of(variant, f)], slot.data_ptr() + f * fb, self.W * 3, fb, False, self.ctx["local_rank"])
                    for f in range(nf)]
            self.jpeg_bytes_in += sum(len(self.jpegs[self.frame_of(variant, f)]) for f in range(nf))
            self.frames_in = getattr(self, "frames_in", 0) + nf

            class _Wait:
                def synchronize(_self):
                    for fu in futs:
                        fu.result()
            self.state["upload_bytes"] = 0
            return slot, _Wait()
        slot, src = self.slots[key], self.host_super(nf, variant)
        r0, r1 = self.pipe.layout(self.H, self.W, nf, self.mode).rows_needed(0 if self.mode == "local" else self.rank, self.H)
        ev = torch.cuda.Event()
        if r1 > r0:
            with torch.cuda.stream(self.copy_stream):
                slot[r0:r1].copy_(src[r0:r1], non_blocking=True)
                ev.record(self.copy_stream)
        else:
            ev.record(self.copy_stream)
        self.state["upload_bytes"] = (r1 - r0) * self.W * 3
        return slot, ev

    # ---- super-resolution: crops of SB frames as one ragged batch on the enhancer's stream -------------------------------------
    def crop_boxes(self, rows, frame_seed):
        p, a = self.pipeline, self.args
        if self.sr_sizes == "from-detections" and rows.shape[0] > 0:
            b = rows[:a.sr_crops, :4].astype(np.int32)
            b[:, 2] = np.maximum(b[:, 2], b[:, 0] + 4); b[:, 3] = np.maximum(b[:, 3], b[:, 1] + 4)     # enhance_image rejects < 4 px
            return b
        sizes = self.fixed_sizes if self.sr_sizes == "fixed" else p.sr_crop_sizes(a.sr_crops, seed=1000 + frame_seed)
        return p.crop_boxes_for_sr(rows, self.H, self.W, a.sr_crops, sizes, seed=frame_seed)

    def drain_sr(self):
        if self.pending:
            t0 = time.perf_counter()
            self.pipe.wait_sr()
            out = self.pending.pop("out")
            self.pending.pop("slots", None)
            ids = self.pending.pop("frame_ids", ())
            v = self.pending.pop("v", None)
            if v is not None:                          # verify: keep every enhanced crop of the batch, by frame
                offs_v = self.pending.pop("offs")
                host = out[:int(offs_v[-1])].cpu().numpy()
                k = 0
                for fid, bx in v:
                    self.rec[fid]["crops"] = [host[int(offs_v[k + j]):int(offs_v[k + j + 1])].copy() for j in range(len(bx))]
                    k += len(bx)
            if self.jpeg_io:
                offs, hs, ws = self.pending.pop("meta")
                keep = self.pending.pop("keep")
                files = self._lib.jpeg_encode_batch_dev(out.data_ptr(), offs, hs, ws, 95, bgr=True, device=self.ctx["local_rank"])     # enhanced crops -> .jpg bytes on the host
                self.jpeg_bytes_out += sum(len(f) for f in files)
                if v is not None:
                    it, k = iter(files), 0
                    for fid, bx in v:
                        self.rec[fid]["files"] = [bytes(next(it)) if keep[k + j] else None for j in range(len(bx))]
                        k += len(bx)
                return
            if self.host_sr is None or self.host_sr.numel() < out.numel():
                self.host_sr = self.torch.empty((int(out.numel() * 1.5),), dtype=self.torch.uint8).pin_memory()
            self.host_sr[:out.numel()].copy_(out)                  # enhanced crops -> host
            t1 = self.tick("sr_wait_and_d2h", t0)
            for fid in ids:
                if fid in self.lat_t0:
                    self.lat.append(t1 - self.lat_t0.pop(fid))

    def flush_sr(self, slot):
        if not self.queue:
            return
        self.drain_sr()
        q = [x[1:3] for x in self.queue if len(x[2])]
        keys = {x[0] for x in self.queue if len(x[2])}
        ids = [x[3] for x in self.queue]
        vmeta = [(x[3], x[2]) for x in self.queue if len(x[2])] if self.verify else None
        self.queue.clear()
        if not q:
            t1 = time.perf_counter()
            for fid in ids:                            # frames without crops are done once their detections are on the host
                if fid in self.lat_t0:
                    self.lat.append(t1 - self.lat_t0.pop(fid))
            return
        t0 = time.perf_counter()
        out, offs = self.pipe.enhance_crops_multi([x[0] for x in q], self.H, self.W, [x[1] for x in q], slot=slot)
        self.tick("sr_submit", t0)
        self.pending["out"] = out
        self.pending["slots"] = keys
        self.pending["frame_ids"] = ids
        if vmeta is not None:
            self.pending["v"], self.pending["offs"] = vmeta, np.array(offs, copy=True)
        if self.jpeg_io:
            bx = np.concatenate([x[1] for x in q], 0).astype(np.int64)
            bx = np.stack([np.clip(bx[:, 0], 0, self.W), np.clip(bx[:, 1], 0, self.H), np.clip(bx[:, 2], 0, self.W), np.clip(bx[:, 3], 0, self.H)], 1)   # as the library clamps
            hs, ws = (bx[:, 3] - bx[:, 1]) * self.pipe.sr.scale, (bx[:, 2] - bx[:, 0]) * self.pipe.sr.scale
            keep = (hs > 0) & (ws > 0)
            self.pending["meta"] = (np.asarray(offs[:len(bx)], np.int64)[keep], hs[keep], ws[keep])
            self.pending["keep"] = keep
        self.sr_px += int(sum(int((b[:, 2] - b[:, 0]) @ (b[:, 3] - b[:, 1])) for _, b in q))

    # ---- one group of DB steps ---------------------------------------------------------------------------------------------------
    def groups(self, n_total):
        return [(g0, min(self.DB, n_total - g0)) for g0 in range(0, n_total, self.DB)]

    def run_group(self, gi, g0, gsz, n_total, nxt, profile):
        """Detect the group's frames as one ragged batch (detector stream) while the next group's frames upload (copy stream) and the
        previous frames' crops are enhanced (enhancer stream); then merge every frame and queue its crops."""
        torch, pipe, a = self.torch, self.pipe, self.args
        B, Bl, H, W = self.B, self.Bl, self.H, self.W
        sf, ev = self.group.pop("next")
        slot_key = self.group.pop("next_key")
        if nxt is not None:
            self.group["next"] = self.upload(nxt[0], Bl * nxt[2])         # this lane's next group: overlaps this group's detection
            self.group["next_key"] = self.state["slot_key"]
        t0 = time.perf_counter()
        if ev is not None:
            ev.synchronize()
        t0 = self.tick("upload_wait", t0)
        for i in range(g0, g0 + gsz):
            for fb in range(Bl):
                self.lat_t0[(i, fb)] = t0
        if profile:
            pipe.det.set_profile(True)
        dets, counts, L, gathered = pipe.detect(sf, H, W, Bl * gsz, exchange=self.exchange, mode=self.mode)
        if profile:
            pipe.det.set_profile(False)
        self.tm["detect"] = self.tm.get("detect", 0.0) + pipe.t_detect
        self.tm["exchange"] = self.tm.get("exchange", 0.0) + pipe.t_exchange
        self.state["gathered"] = gathered
        my = 0 if self.mode == "local" else self.rank
        for i in range(g0, g0 + gsz):
            last = profile and i == g0 + gsz - 1
            for fb in range(Bl):
                f = (i - g0) * Bl + fb
                spread = L.owner(f) < 0                                                   # this frame's items live on several ranks
                if not spread and L.owner(f) != my:
                    self.lat_t0.pop((i, fb), None)
                    continue
                if spread and not gathered:
                    raise RuntimeError("frame items are spread over ranks but the exchange was disabled")
                t0 = time.perf_counter()
                rows_d, n_d = pipe.merge_frame_of(dets, counts, L, f, gathered)      # replicated on every rank when spread (§8e)
                n = pipe.merged_count(n_d)
                self.host_rows[:n].copy_(rows_d[:n])                                   # merged detections -> host
                rows = self.host_rows[:n].numpy().copy()
                self.tick("merge_and_d2h", t0)
                rows[:, [1, 3]] -= f * H
                self.state["rows"] = rows
                if a.sr_crops > 0:
                    seed = i * B + (self.rank if self.mode == "local" else fb)
                    boxes = self.crop_boxes(rows, seed)
                    self.state["boxes"] = boxes
                    if seed == 0:
                        self.state["boxes_frame0"] = boxes                              # what the CPU baseline enhances
                    if spread:                                                          # crops of ONE frame over the ranks: LPT by area
                        own = self.pipeline.lpt_assign((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]), self.world) == self.rank
                        boxes = boxes[own]
                    self.queue.append((slot_key, sf[f * H:(f + 1) * H], boxes, (i, fb)))
                    if self.verify:
                        self.rec[(i, fb)] = {"rows": rows.copy(), "boxes": np.array(boxes, copy=True)}
                else:
                    if self.verify:
                        self.rec[(i, fb)] = {"rows": rows.copy(), "boxes": None}
                    self.lat.append(time.perf_counter() - self.lat_t0.pop((i, fb)))
            if a.sr_crops > 0 and (len(self.queue) >= self.SB or last):
                prof = self.prof_flush is not None and self.flush_no == self.prof_flush     # the kernel times of ONE whole SR batch
                if prof:
                    self.drain_sr()
                    for ev_done in getattr(self, "_wait_for", ()):       # the profiled SR batch runs with the card to itself: the other lanes (which own no
                        ev_done.wait()                                   # later group) finish their last batches first
                    pipe.sr.set_profile(True)
                self.flush_sr(slot=self.flush_no & 1)
                if a.sr_exclusive:
                    self.drain_sr()                                                      # experiment: the enhancer never runs beside the detector
                if prof:
                    self.drain_sr()
                    torch.cuda.synchronize(self.dev)
                    pipe.sr.set_profile(False)
                self.flush_no += 1

    def my_groups(self, n_steps):
        """This lane's share of the loop: (global group index, first global step, steps) of every n_lanes-th group."""
        return [(gi, g0, gsz) for gi, (g0, gsz) in enumerate(self.groups(n_steps)) if gi % self.n_lanes == self.lane]

    def loop_own(self, n_steps, profile_last=False):
        gs = self.my_groups(n_steps)
        if not gs:                                     # --warmup 0, or fewer groups than lanes
            return
        # SR kernels are profiled on the last FULL batch of the loop (a trailing partial batch would understate the launch sizes)
        own = sum(g[2] for g in gs)
        n_flush = -(-(own * self.Bl) // self.SB) if self.args.sr_crops > 0 else 0
        n_full = (own * self.Bl) // self.SB
        self.prof_flush = (max(n_full, 1) - 1 if n_flush else None) if profile_last else None
        self.flush_no = 0
        self.group["next"] = self.upload(gs[0][0], self.Bl * gs[0][2])
        self.group["next_key"] = self.state["slot_key"]
        for k, (gi, g0, gsz) in enumerate(gs):
            self.run_group(gi, g0, gsz, n_steps, gs[k + 1] if k + 1 < len(gs) else None, profile_last and k == len(gs) - 1)
        self.flush_sr(slot=self.flush_no & 1)
        self.drain_sr()                                # the last frames' crops are part of the timed work

    def loop(self, n_steps, profile_last=False):
        """All lanes over n_steps steps. The kernel events (profile_last) are taken on the lane that owns the LAST group: its final SR batch is
        the drain of the pipeline, so its launches are timed without another lane's kernels sharing the CUs."""
        lanes = [self] + self.sub
        n_groups = len(self.groups(n_steps))
        prof_lane = (n_groups - 1) % self.n_lanes if n_groups else 0
        self.prof_pipe = lanes[prof_lane].pipe
        if not self.sub:
            return self.loop_own(n_steps, profile_last)
        import threading
        errs = []
        done = [threading.Event() for _ in lanes]
        for k, r in enumerate(lanes):
            r._wait_for = [d for j, d in enumerate(done) if j != k] if (profile_last and k == prof_lane) else []

        def work(k, r, prof):
            try:
                self.torch.cuda.set_device(self.dev)
                r.loop_own(n_steps, prof)
            except BaseException as e:      # noqa: BLE001
                errs.append(e)
            finally:
                done[k].set()

        th = [threading.Thread(target=work, args=(k, r, profile_last and k == prof_lane)) for k, r in enumerate(lanes)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]

    def verify_against_synchronous(self, n_steps):
        """What the pipelined loop (frame-slot ring, copy / detector / enhancer streams, lanes, an SR batch in flight while the next groups are
        detected) produced for every frame of loop(n_steps), against the same calls made one at a time on fresh buffers with a device
        synchronisation after each: the group's frames in one new tensor, detect, merge per frame, the crops of ONE frame per enhancer call (the
        pipelined loop batches SB frames: a crop's bytes must not depend on its batch), JPEG files encoded per frame. Byte for byte."""
        torch, pipe = self.torch, self.pipe
        assert self.mode == "local" and self.world == 1, "the verification loop is a single-process check"
        rec = {}
        for r in [self] + self.sub:
            rec.update(r.rec)
        H, W, Bl, hf = self.H, self.W, self.Bl, self.ctx["host_frames"]
        fbytes = H * W * 3
        out = {"steps": n_steps, "lanes": self.n_lanes, "det_batch_frames": self.DB, "sr_batch_frames": self.SB, "jpeg_io": bool(self.jpeg_io), "frames": 0, "crops": 0,
               "rows_equal": True, "boxes_equal": True, "crops_equal": True, "files_equal": True if self.jpeg_io else None, "mismatches": []}

        def bad(key, what, fid):
            out[key] = False
            if len(out["mismatches"]) < 8:
                out["mismatches"].append(f"{what} of frame {fid}")

        for gi, (g0, gsz) in enumerate(self.groups(n_steps)):
            variant, nf = gi % len(hf), Bl * gsz
            if self.jpeg_io:
                sf = torch.empty((nf * H, W, 3), dtype=torch.uint8, device=self.dev)
                for f in range(nf):
                    self._lib.jpeg_decode_dev(self.jpegs[self.frame_of(variant, f)], sf.data_ptr() + f * fbytes, W * 3, fbytes, False, self.ctx["local_rank"])
            else:
                sf = torch.from_numpy(np.concatenate([hf[self.frame_of(variant, f)] for f in range(nf)], 0)).to(self.dev)
            torch.cuda.synchronize(self.dev)
            dets, counts, L, gathered = pipe.detect(sf, H, W, nf, exchange=self.exchange, mode=self.mode)
            torch.cuda.synchronize(self.dev)
            for i in range(g0, g0 + gsz):
                for fb in range(Bl):
                    f, fid = (i - g0) * Bl + fb, (i, fb)
                    got = rec.get(fid)
                    if got is None:
                        bad("rows_equal", "no record", fid)
                        continue
                    out["frames"] += 1
                    rows_d, n_d = pipe.merge_frame_of(dets, counts, L, f, gathered)
                    n = pipe.merged_count(n_d)
                    rows = rows_d[:n].cpu().numpy().copy()
                    rows[:, [1, 3]] -= f * H
                    if not np.array_equal(rows, got["rows"]):
                        bad("rows_equal", "merged rows", fid)
                    if self.args.sr_crops <= 0:
                        continue
                    boxes = self.crop_boxes(rows, i * self.B + self.rank)
                    if not np.array_equal(boxes, got["boxes"]):
                        bad("boxes_equal", "crop boxes", fid)
                        continue
                    if not len(boxes):
                        continue
                    o, offs = pipe.enhance_crops(sf[f * H:(f + 1) * H], H, W, boxes, wait=True, slot=0)
                    torch.cuda.synchronize(self.dev)
                    host = o[:int(offs[-1])].cpu().numpy()
                    crops = got.get("crops")
                    if crops is None or len(crops) != len(boxes):
                        bad("crops_equal", "crop list", fid)
                        continue
                    for j in range(len(boxes)):
                        out["crops"] += 1
                        if not np.array_equal(host[int(offs[j]):int(offs[j + 1])], crops[j]):
                            bad("crops_equal", f"crop {j}", fid)
                    if self.jpeg_io:
                        bx = boxes.astype(np.int64)
                        bx = np.stack([np.clip(bx[:, 0], 0, W), np.clip(bx[:, 1], 0, H), np.clip(bx[:, 2], 0, W), np.clip(bx[:, 3], 0, H)], 1)
                        hs, ws = (bx[:, 3] - bx[:, 1]) * pipe.sr.scale, (bx[:, 2] - bx[:, 0]) * pipe.sr.scale
                        keep = (hs > 0) & (ws > 0)
                        files = self._lib.jpeg_encode_batch_dev(o.data_ptr(), np.asarray(offs[:len(bx)], np.int64)[keep], hs[keep], ws[keep], 95, bgr=True, device=self.ctx["local_rank"])
                        mine = [x for x in got.get("files", []) if x is not None]
                        if len(mine) != len(files) or any(bytes(a) != b for a, b in zip(files, mine)):
                            bad("files_equal", "JPEG files", fid)
        out["ok"] = bool(out["frames"] == n_steps * self.Bl and out["rows_equal"] and out["boxes_equal"] and out["crops_equal"] and out["files_equal"] in (True, None)
                         and (self.args.sr_crops <= 0 or out["crops"] > 0))
        return out

    def run_verify(self, steps):
        """Plans first (untimed set-up), then the pipelined loop with recording, then the synchronous pass."""
        self.setup(2, steps)
        for r in [self] + self.sub:
            r.rec.clear()
        self.loop(steps)
        self.torch.cuda.synchronize(self.dev)
        return self.verify_against_synchronous(steps)

    def barrier(self):
        import torch.distributed as dist
        self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def setup(self, warmup, steps):
        """Untimed, like loading weights: lay out, tune and graph-capture the detector plan of every group size, and the enhancer plan of
        every capacity bucket the loops will meet (a plan is keyed by capacity, not by crop sizes: a stream builds each bucket once)."""
        for r in [self] + self.sub:
            for gsz in sorted({g[2] for n in (warmup, steps) for g in r.my_groups(n)}):
                sf, ev = r.upload(0, r.Bl * gsz)
                if ev is not None:
                    ev.synchronize()
                for _ in range(3):
                    r.pipe.detect(sf, r.H, r.W, r.Bl * gsz, exchange=r.exchange, mode=r.mode)
        self.loop(max(warmup, 2 * self.DB * self.n_lanes))
        if self.args.sr_crops > 0:
            self.loop(steps)                           # rehearsal with the timed loop's own crop sizes: every bucket it meets exists afterwards
        for r in [self] + self.sub:
            r.sr_px = 0

    def timed(self, warmup, steps, profile_last=True):
        import torch.distributed as dist
        self.setup(warmup, steps)
        self.loop(warmup)
        self.barrier()
        lanes = [self] + self.sub
        for r in lanes:
            r.sr_px = 0
            r.tm, r.lat, r.lat_t0 = {}, [], {}
            if r.jpeg_io:
                r.jpeg_bytes_in = r.jpeg_bytes_out = 0
                r.frames_in = 0
        t0 = time.perf_counter()
        self.loop(steps)                               # nothing is profiled inside the timed region
        self.barrier()
        dt = time.perf_counter() - t0
        if self.world > 1:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev if self.ctx["backend"] == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        self.dt_own = time.perf_counter() - t0
        for r in self.sub:                             # one view of the run: stage times summed over the lanes' host threads, every frame's latency
            for k, v in r.tm.items():
                self.tm[k] = self.tm.get(k, 0.0) + v
            self.lat += r.lat
            self.sr_px += r.sr_px
            if r.jpeg_io:
                self.jpeg_bytes_in += r.jpeg_bytes_in
                self.jpeg_bytes_out += r.jpeg_bytes_out
                self.frames_in = getattr(self, "frames_in", 0) + getattr(r, "frames_in", 0)
            for key in ("boxes_frame0",):
                if key in r.state and key not in self.state:
                    self.state[key] = r.state[key]
        if profile_last:
            self.profile_pass()
        return dt

    def profile_pass(self):
        """The kernel events behind `roofline` / `conv_profile_last_step`: ONE extra pass after the timed region, on lane 0 alone (no other lane's
        kernels beside the launches that are timed), over the first max(SB, 2 DB) steps of the loop — the same frames, crops and seeds, one full
        SR batch. Its last detection group and its last full SR batch run eagerly with an event pair around every conv launch."""
        keep = (self.sub, self.n_lanes, self.tm, self.lat, self.lat_t0, self.sr_px, dict(self.state))
        jp = (self.jpeg_bytes_in, self.jpeg_bytes_out, getattr(self, "frames_in", 0)) if self.jpeg_io else None
        self.sub, self.n_lanes, self.tm, self.lat, self.lat_t0 = [], 1, {}, [], {}
        self._wait_for = []
        try:
            n = -(-max(self.SB, 2 * self.DB) // self.Bl)
            self.loop_own(n, profile_last=True)
            self.torch.cuda.synchronize(self.dev)
        finally:
            self.sub, self.n_lanes, self.tm, self.lat, self.lat_t0, self.sr_px, self.state = keep
            if jp:
                self.jpeg_bytes_in, self.jpeg_bytes_out, self.frames_in = jp
        self.prof_pipe = self.pipe
        self.profile_steps = n

    def report(self, steps):
        """Per-rank view of the timed loop: host wall time per stage (ms per step), per-frame latency percentiles."""
        lat = np.sort(np.asarray(self.lat, np.float64)) * 1e3
        mine = {"rank": self.rank, "stage_ms_per_step": {k: round(v / max(steps, 1) * 1e3, 3) for k, v in sorted(self.tm.items())},
                "frames": int(len(lat)), "latency_ms": ({"p50": round(float(lat[len(lat) // 2]), 2), "p99": round(float(lat[min(len(lat) - 1, int(len(lat) * 0.99))]), 2),
                                                        "max": round(float(lat[-1]), 2)} if len(lat) else None)}
        if self.world == 1:
            return [mine]
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, mine)
        return out

    def describe(self):
        a = self.args
        return (f"{self.W}x{self.H} frame, YOLO11{a.arch}-pose (random-init), SAHI {a.slice}x{a.slice}/{a.overlap} ({self.items_per_frame - 1} slices + full frame), "
                f"net input {self.imgsz}, conf {a.conf}, NMS 0.7, {self.pp_type}/IOS/0.5{'/agnostic' if self.class_agnostic else ''} merge"
                + (f", Real-ESRGAN x4 on {a.sr_crops} crops/frame (sizes {self.sr_sizes})" if a.sr_crops > 0 else ", no SR"))


def make_ctx(args, rank=0, world=1, local_rank=0, dev=None, backend="nccl"):
    """What every Runner of a process shares: the random-init weights of BASELINE.json's architectures and the synthetic frames."""
    import torch
    from ffp_amd import synth
    return {"rank": rank, "world": world, "local_rank": local_rank, "dev": dev if dev is not None else torch.device("cuda", local_rank), "backend": backend,
            "det_w": synth.yolo11_pose_weights(args.arch), "sr_w": synth.rrdbnet_weights(4, 23) if args.sr_crops > 0 else None,
            "host_frames": [synth.synthetic_frame(args.height, args.width, seed=i) for i in range(max(1, args.distinct_frames))]}


COMPAT = os.path.join(ROOT, "face-detection-with-yolov11-sahi-and-real-esrgan_amd", "compat")


def compat_api_row(args, ctx, steps):
    """secondary.through_compat_api: the reference's own loop, one frame and one crop at a time, through the drop-in modules under compat/ —
    `get_sliced_prediction(image, detection_model, slice_height, slice_width, overlap…)` as pipeline_v4_yolo/app_yolo_sahi.py:49-56 calls it, then
    `FaceEnhancer.enhance_image(crop)` per crop as pipeline_v1_detection_first/app_v1.py:91-104 -> utils/enhancer.py:344-391 does. Frames come
    from memory (an ndarray) and, second figure, from .jpg files on disk read the way the reference reads them (PIL inside SAHI, cv2.imread
    for the crops). Same frames, crop law and seeds as the headline; nothing batched across frames or crops."""
    import contextlib
    import io
    import tempfile
    from ffp_amd import _lib, pipeline
    sys.path.insert(0, COMPAT)
    quiet = io.StringIO()
    try:
        import cv2                                                              # compat/cv2: imread / imwrite on this build's JPEG codec
        from sahi.predict import get_sliced_prediction
        from utils.enhancer import FaceEnhancer
        from utils.yolo_wrapper import YOLOv11PoseDetectionModel
        with contextlib.redirect_stdout(quiet):
            model = YOLOv11PoseDetectionModel(model_path=ctx["det_w"], confidence_threshold=args.conf, device=f"cuda:{ctx['local_rank']}", image_size=args.imgsz)
            enh = FaceEnhancer("RealESRGAN_x4plus", model_path=ctx["sr_w"], scale=4, tile=400, half=True) if args.sr_crops > 0 else None
        hf, H, W = ctx["host_frames"], args.height, args.width
        tm = {"sliced_prediction": 0.0, "enhance_crops": 0.0, "read_crop_source": 0.0}
        state = {"batched": False}

        def one(k, src):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(quiet):
                res = get_sliced_prediction(src, model, slice_height=args.slice, slice_width=args.slice, overlap_height_ratio=args.overlap,
                                            overlap_width_ratio=args.overlap, postprocess_type=args.pp_type, postprocess_class_agnostic=args.class_agnostic, verbose=0)
            t1 = time.perf_counter()
            tm["sliced_prediction"] += t1 - t0
            preds = sorted(res.object_prediction_list, key=lambda p: -p.score.value)
            state["detections"] = len(preds)
            if enh is None:
                return
            bgr = cv2.imread(src) if isinstance(src, str) else src[..., ::-1]      # app_v1 reads the picture once more with cv2 for the crops (BGR)
            t2 = time.perf_counter()
            tm["read_crop_source"] += t2 - t1
            rows = np.asarray([list(p.bbox.to_xyxy()) + [p.score.value] for p in preds], np.float32).reshape(-1, 5)
            boxes = pipeline.crop_boxes_for_sr(rows, H, W, args.sr_crops, pipeline.sr_crop_sizes(args.sr_crops, seed=1000 + k), seed=k)
            crops = [np.ascontiguousarray(bgr[y0:y1, x0:x1]) for x0, y0, x1, y1 in boxes]
            with contextlib.redirect_stdout(quiet):
                outs = enh.enhance_images(crops) if state["batched"] else [enh.enhance_image(c) for c in crops]
            for c, (out, ok) in zip(crops, outs):
                assert ok and out.shape[0] == 4 * c.shape[0]
            tm["enhance_crops"] += time.perf_counter() - t2

        def timed(srcs):
            for k in range(2):
                one(k, srcs[k % len(srcs)])
            for key in tm:
                tm[key] = 0.0
            t0 = time.perf_counter()
            for k in range(steps):
                one(k, srcs[k % len(srcs)])
            dt = time.perf_counter() - t0
            return steps / dt, {key: round(v / steps * 1e3, 3) for key, v in tm.items()}

        def file_flow(srcs, n):
            """pipeline_v1_detection_first/app_v1.py:52-104 on files, literally: sliced prediction on the .jpg, keypoints attached, every detected face saved as a crop
            file (save_face_crops), the crop directory enhanced (enhance_face_crops_batch: one ragged GPU batch inside) and written back as files."""
            import shutil
            from utils.visualization import save_face_crops
            from utils.enhancer import enhance_face_crops_batch
            td2 = tempfile.mkdtemp()
            ncrops = 0
            try:
                def once(k):
                    nonlocal ncrops
                    src = srcs[k % len(srcs)]
                    d = os.path.join(td2, f"crops{k}")
                    with contextlib.redirect_stdout(quiet):
                        res = get_sliced_prediction(src, model, slice_height=args.slice, slice_width=args.slice, overlap_height_ratio=args.overlap, overlap_width_ratio=args.overlap, verbose=0)
                        model.attach_keypoints_to_predictions(res.object_prediction_list)
                        saved = save_face_crops(src, res, d, prefix="f") if res.object_prediction_list else []
                        if saved and enh is not None:
                            out = enhance_face_crops_batch(crops_dir=d, enhancer=enh, prefix="f")
                            assert out["statistics"]["successful"] + out["statistics"]["failed"] == len(saved)
                    ncrops += len(saved)
                    shutil.rmtree(d, ignore_errors=True)
                    shutil.rmtree(os.path.join(td2, "f_enhanced"), ignore_errors=True)
                once(0); once(1)
                ncrops = 0
                t0 = time.perf_counter()
                for k in range(n):
                    once(k)
                return n / (time.perf_counter() - t0), ncrops / n
            finally:
                shutil.rmtree(td2, ignore_errors=True)

        fps_mem, st_mem = timed(hf)
        state["batched"] = True
        fps_bat, st_bat = timed(hf) if enh is not None else (None, None)
        state["batched"] = False
        with tempfile.TemporaryDirectory() as td:
            paths = []
            for i, f in enumerate(hf):
                paths.append(os.path.join(td, f"frame{i}.jpg"))
                with open(paths[-1], "wb") as fh:
                    fh.write(_lib.jpeg_encode(f, 95, bgr=False))
            fps_jpg, st_jpg = timed(paths)
            fps_files, crops_files = file_flow(paths, steps)
        return {"value": round(fps_mem, 3), "unit": "frames/s", "steps": steps, "frames": "ndarray in host memory", "host_stage_ms_per_frame": st_mem,
                "from_jpg_files": {"value": round(fps_jpg, 3), "unit": "frames/s", "host_stage_ms_per_frame": st_jpg},
                "crops_as_one_batch": ({"value": round(fps_bat, 3), "unit": "frames/s", "host_stage_ms_per_frame": st_bat,
                                        "note": "FaceEnhancer.enhance_images (extension; what compat's enhance_face_crops_batch does inside): the frame's crops as one ragged GPU "
                                                "batch instead of one synchronous enhance_image per crop"} if fps_bat else None),
                "note": "enhance_image is synchronous per crop by signature (it returns the pixels): 32 crops = 32 dependent passes through 349 conv launches, "
                        "about 3 ms each however small the crop; the detection half runs at the frame_by_frame rate",
                "app_v1_file_flow": {"value": round(fps_files, 3), "unit": "frames/s", "crops_per_frame": round(crops_files, 1),
                                     "calls": "get_sliced_prediction(.jpg) + attach_keypoints_to_predictions + save_face_crops + enhance_face_crops_batch (pipeline_v1_detection_first/app_v1.py:52-104): "
                                              "every DETECTED face is cropped, written, enhanced and written again — the crop count is what the random-init detector finds, not the 32-crop law"},
                "detections_last_frame": state.get("detections"),
                "calls": "sahi.predict.get_sliced_prediction + utils.enhancer.FaceEnhancer.enhance_image per crop (compat/), one frame and one crop at a time"}
    finally:
        sys.path.remove(COMPAT)
        for m in [k for k in sys.modules if k.split(".")[0] in ("sahi", "utils", "cv2", "eval")]:
            if getattr(sys.modules[m], "__file__", "") and COMPAT in (sys.modules[m].__file__ or ""):
                del sys.modules[m]


def config1_row(args, ctx):
    """secondary.config1_640_yolo11n — BASELINE configs[0] as scripts/inference_time.py:43-56 measures it: one 640 x 640 image, YOLO11n, no slicing,
    no SR; `model.predict(img, imgsz=640)` once as warm-up, then ONE timed call (fps = 1 / that call). Through compat's YOLO (utils/yolo_wrapper.py:55's
    `YOLO(model_path)`), image handed over as PIL like the script does. The median of 20 further calls is given beside it."""
    import contextlib
    import io
    from PIL import Image
    from ffp_amd import synth
    sys.path.insert(0, COMPAT)
    try:
        from utils.yolo_wrapper import YOLO
        with contextlib.redirect_stdout(io.StringIO()):
            model = YOLO(synth.yolo11_pose_weights("n"), device=f"cuda:{ctx['local_rank']}")
        img = Image.fromarray(synth.synthetic_frame(640, 640, seed=3, n_blobs=12))
        model.predict(img, imgsz=640, device=f"cuda:{ctx['local_rank']}", verbose=False)
        t0 = time.time()
        r = model.predict(img, imgsz=640, device=f"cuda:{ctx['local_rank']}", verbose=False)
        first = time.time() - t0
        more = []
        for _ in range(20):
            t0 = time.perf_counter()
            model.predict(img, imgsz=640, device=f"cuda:{ctx['local_rank']}", verbose=False)
            more.append(time.perf_counter() - t0)
        del model
        return {"value": round(1.0 / first, 2), "unit": "images/s", "ms": round(first * 1e3, 3), "median_ms_of_20_more": round(float(np.median(more)) * 1e3, 3),
                "workload": "one 640x640 image, YOLO11n-pose (random-init), no SAHI, no SR; model.predict after one warm-up call (scripts/inference_time.py:43-56)",
                "detections": int(len(r[0].boxes))}
    finally:
        sys.path.remove(COMPAT)
        for m in [k for k in sys.modules if k.split(".")[0] in ("sahi", "utils", "cv2", "eval")]:
            if getattr(sys.modules[m], "__file__", "") and COMPAT in (sys.modules[m].__file__ or ""):
                del sys.modules[m]


def verify_rows(args, ctx, pipes, steps=25):
    """--verify: pipelined against synchronous, DB / SB as configured (5 / 10 by default), with 1 lane, 2 lanes, and 1 lane with JPEG in and out."""
    rows = {}
    for name, kw in (("one_lane", dict(lanes=1)), ("two_lanes", dict(lanes=2)), ("one_lane_jpeg_io", dict(lanes=1, jpeg_io=True))):
        if kw.get("jpeg_io") and args.sr_crops <= 0:
            continue
        r = Runner(args, ctx, pipes=pipes, verify=True, **kw)
        try:
            rows[name] = r.run_verify(steps)
        finally:
            r.close()
    return rows


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import ffp_amd  # noqa: F401
    from ffp_amd import _lib, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    # FFP_BENCH_BACKEND=gloo FFP_BENCH_ONE_DEVICE=1 rehearses the N>1 code path with several ranks on ONE GPU (no RCCL peers)
    backend = os.environ.get("FFP_BENCH_BACKEND", "nccl")
    if os.environ.get("FFP_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    H, W = args.height, args.width
    ctx = make_ctx(args, rank, world, local_rank, dev, backend)
    if args.conv_totals:
        _lib.conv_totals_enable(True)
    hbm_peak = [0]

    def sample_hbm():
        """Bytes in use on the card right now (every process, every allocator: hipMemGetInfo), kept as the maximum seen at the sampling points:
        after the headline's timed loop and at the end of every secondary row, before the row's handles are released."""
        free, total = torch.cuda.mem_get_info(dev)
        hbm_peak[0] = max(hbm_peak[0], int(total - free))
        return int(total - free)

    main_r = Runner(args, ctx)
    dt = main_r.timed(args.warmup, args.steps)
    hbm_headline = sample_hbm()
    main_mem = main_r.mem()
    B = main_r.B
    fps = B * args.steps / dt
    main_report = main_r.report(args.steps)
    pipe = main_r.prof_pipe                        # the lane that carried the kernel events of the timed loop (lane 0 when there is one lane)
    pipes = main_r.pipes
    prof = [dict(p, stage="det") for p in pipe.det.profile()]
    if pipe.sr is not None:
        prof += [dict(p, stage="sr") for p in pipe.sr.profile()]
    prof.sort(key=lambda p: -p["ms"])
    stage_ms = pipe.det.last_ms()
    sr_ms = pipe.sr.last_ms() if pipe.sr is not None else None
    sr_state = pipe.sr.plan_state() if pipe.sr is not None else None
    det_graph = pipe.det.graph_status()
    main_state = dict(main_r.state)
    main_sr_px = main_r.sr_px

    # ---- secondary rows, measured in this same run (--secondary-steps, default 20 like the headline): what the headline configuration is NOT ------------------------
    secondary = {}
    if not args.no_secondary:
        ss, sw = max(2, args.secondary_steps), max(2, min(args.warmup, main_r.DB))     # whole detection groups: a warm-up of 4 would lay out a 4-frame plan beside the 5-frame one

        only = set(x for x in args.secondary_only.split(",") if x)

        def sec(name, **kw):
            if only and name not in only:
                secondary[name] = {"value": None, "skipped": "--secondary-only"}
                return None
            # a secondary row must never cost the headline line: a failure (e.g. out of memory in the 4x-larger imgsz-1024 plan) is
            # reported in place of the row; the rows that follow still run
            r = None
            try:
                steps_r = kw.pop("steps", ss)
                prof_sr = kw.pop("profile_sr", False)
                ctx_r, share = ctx, pipes
                wl = kw.get("workload")
                if wl:                                   # another BASELINE config: its own weights / frames, nothing shared with the headline's handles
                    ctx_r, share = dict(ctx), None
                    if wl.get("arch", args.arch) != args.arch:
                        ctx_r["det_w"] = synth.yolo11_pose_weights(wl["arch"])
                    if (wl.get("height", H), wl.get("width", W)) != (H, W):
                        assert (wl["height"], wl["width"]) == (2 * H, 2 * W), "workload frames are 2 x 2 mosaics of the headline's synthetic frames"
                        hf = ctx["host_frames"]
                        ctx_r["host_frames"] = [np.ascontiguousarray(np.concatenate([np.concatenate([hf[i], hf[(i + 1) % len(hf)]], 1),
                                                                                     np.concatenate([hf[(i + 1) % len(hf)], hf[i]], 1)], 0)) for i in range(len(hf))]
                    if wl.get("sr_crops", args.sr_crops) > 0 and ctx_r["sr_w"] is None:
                        ctx_r["sr_w"] = synth.rrdbnet_weights(4, 23)
                if share is None or kw.get("det_precision") or kw.get("imgsz"):
                    # a row with handles or plans of its own: the headline handles give their plans back first (rebuilt, untimed, by the next row
                    # that shares them) — four ranks on one card, or the 4x-larger plans of image_size 1024, fit beside nothing else
                    for p_ in {id(q): q for q in pipes}.values():
                        p_.det.drop_plans()
                        if p_.sr is not None:
                            p_.sr.drop_plans()
                r = Runner(args, ctx_r, pipes=share, **kw)
                d = r.timed(sw, steps_r, profile_last=prof_sr)
                rep = r.report(steps_r)
                secondary[name] = {"value": round(r.B * steps_r / d, 3), "unit": "frames/s", "ms_per_step": round(d / steps_r * 1e3, 3), "steps": steps_r, "workload": r.describe(),
                                   "frames_per_step": r.B, "det_batch_frames": r.DB, "sr_batch_frames": r.SB, "lanes": r.n_lanes, "mode": r.mode, "gathered": bool(r.state.get("gathered", False)),
                                   "latency_ms_rank0": rep[0]["latency_ms"], "host_stage_ms_per_step_rank0": rep[0]["stage_ms_per_step"], "hbm_bytes": r.mem(),
                                   "hbm_bytes_in_use_on_card": sample_hbm()}
                if world > 1:
                    secondary[name]["per_rank"] = rep
                if prof_sr and r.prof_pipe.sr is not None:         # the dominant kernel at THIS row's launch sizes
                    top = max(r.prof_pipe.sr.profile(), key=lambda q: q["ms"], default=None)
                    if top and top["ms"] > 0:
                        ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
                        secondary[name]["roofline_dominant"] = {"kernel": top["variant"], "achieved": round(ach, 2), "peak": PEAK_TFLOPS["f16"], "unit": "TFLOP/s",
                                                                "frac": round(ach / PEAK_TFLOPS["f16"], 4), "launches": top["launches"],
                                                                "avg_launch_us": round(top["ms"] * 1e3 / max(top["launches"], 1), 2)}
            except Exception as e:      # noqa: BLE001
                secondary[name] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
                print(f"[bench] secondary row {name} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            finally:
                if r is not None:
                    r.close()                          # a row's own handles, slots and pinned buffers go before the next row is built
                if rank == 0:
                    print(f"[bench] row {name}: {secondary.get(name, {}).get('value')}", file=sys.stderr, flush=True)
            return r

        # the reference's own order: strictly one frame, one SR pass at a time (docs sahi/predict.py:226,270) — nothing batched across frames
        sec("frame_by_frame", det_batch=1, sr_batch=1, profile_sr=True, lanes=1)      # one lane: never two frames on the card at once
        sec("steps_200", steps=200)                       # the headline configuration over a 10x longer timed region
        if main_r.mode == "local" and world == 1:
            # pipeline lanes: L detector + enhancer handle pairs, L host threads, detection groups dealt round-robin — the same frames, crops and
            # seeds whatever L is. Their kernels interleave on the card (an HBM-bound 1x1 layer beside an MFMA-bound 3x3 or SR launch, a 16^2-level
            # layer that fills half the CUs beside anything). The row is the headline's loop on the OTHER lane count
            if main_r.n_lanes == 1:
                sec("two_lanes", lanes=2)
                sec("two_lanes_steps_200", lanes=2, steps=200)
            else:
                sec("one_lane", lanes=1)
                sec("one_lane_steps_200", lanes=1, steps=200)
        sec("frames_resident_in_hbm", resident=True)
        def sec_fn(name, fn):
            if only and name not in only:
                secondary[name] = {"value": None, "skipped": "--secondary-only"}
                return
            try:
                secondary[name] = fn()
            except Exception as e:      # noqa: BLE001
                secondary[name] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
                print(f"[bench] secondary row {name} failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            torch.cuda.empty_cache()

        if world == 1:
            # the drop-in API itself (the pipeline objects above are this build's own host side): what a caller of the reference's functions gets
            sec_fn("through_compat_api", lambda: compat_api_row(args, ctx, min(ss, 20)))
            fb, ca = secondary.get("frame_by_frame", {}).get("value"), secondary.get("through_compat_api", {}).get("value")
            if fb and ca:
                secondary["through_compat_api"]["vs_frame_by_frame"] = round(ca / fb, 3)
                cb = (secondary["through_compat_api"].get("crops_as_one_batch") or {}).get("value")
                if cb:
                    secondary["through_compat_api"]["crops_as_one_batch"]["vs_frame_by_frame"] = round(cb / fb, 3)
            sec_fn("config1_640_yolo11n", lambda: config1_row(args, ctx))
        if world == 1:
            # the other BASELINE configs on the final tree (config 3 is the headline; 4 and the 8-GPU part of 5 need the driver's node)
            sec("config2_detect_only", workload={"sr_crops": 0})           # single 4K image, YOLO11s, SAHI 512 / 0.2, no SR
            sec("config5_8k_640", workload={"height": 2 * H, "width": 2 * W, "slice": 640, "overlap": 0.25, "imgsz": 640, "sr_crops": 128},
                imgsz=640, det_batch=2, sr_batch=2, steps=min(ss, 8), lanes=1)   # 8K frame, 640 / 0.25 (144 slices + full frame), x4 SR on 128 crops: the one-GPU shape of config 5
        if args.sr_crops > 0 and args.sr_sizes != "fixed":
            sec("sr_sizes_fixed", sr_sizes="fixed")
        sec("merge_nms_ios_agnostic", pp_type="NMS", class_agnostic=True)
        if world > 1:
            if not main_state.get("gathered", False):
                sec("with_allgather", exchange="always", det_batch=1)       # "global" layout: a rank's super-frame spans every rank's frame
            if main_r.B != 1:
                # the north_star split: EVERY frame's 61 items over the ranks, det_batch_frames frames in flight, one all-gather per group
                r = sec("one_frame_across_ranks", frames_per_step=1, exchange="auto")
                secondary["one_frame_across_ranks"]["scaling"] = "strong"        # (key exists whether or not the row failed)
        if args.sr_crops > 0 and world == 1:
            r = sec("with_jpeg_decode_and_encode", jpeg_io=True, lanes=1)     # one lane: two lanes' decode pools (8 host threads each) and codec calls get in each other's way (110 vs 125-128 frames/s)
            if r is not None and secondary["with_jpeg_decode_and_encode"].get("value") is not None:
              secondary["with_jpeg_decode_and_encode"].update({"note": "frames arrive as JPEG (quality 95, 4:2:0) and are decoded into device memory; enhanced crops leave as JPEG files "
                                                                     "(quality 95, byte-identical to cv2.imwrite); reported separately as SURVEY §8(d) prescribes",
                                                             "jpeg_bytes_in_per_frame": int(r.jpeg_bytes_in / max(1, r.frames_in)), "jpeg_bytes_out_per_frame": int(r.jpeg_bytes_out / max(1, r.frames_in))})
        if args.imgsz != 1024:
            sec("image_size_1024_reference_default", imgsz=1024, det_batch=min(main_r.DB, 2))      # 4x the activations per item: keep the plan in memory
        if args.det_precision != "f32":
            sec("detector_exact_f32", det_precision="f32")

    if rank == 0:
        roof = None
        pmc = {}
        pmc_src = None
        for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):   # HBM bytes per launch from the committed PMC passes, keyed by kernel variant
            try:
                with open(os.path.join(ROOT, "profiles", name)) as fh:
                    pmc = json.load(fh).get("kernels", {})
                pmc_src = name
                break
            except Exception:
                pmc = {}
        if prof:
            d = prof[0]
            dtp = d["variant"].split("_")[0]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
            kname = "conv_trunk_kernel" if d["variant"].endswith("_trunk") else "conv_rows16pc_kernel" if d["variant"].endswith("_rows16pc") else "conv_rows16_kernel" if d["variant"].endswith("_rows16") else "conv_rows_kernel" if "_rows" in d["variant"] else "conv_mfma_kernel"
            roof = {"bound": "mfma", "kernel": f"{kname}<{d['variant']}> ({d['stage']})", "achieved": round(ach, 2),
                    "peak": PEAK_TFLOPS[dtp], "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS[dtp], 4),
                    # algorithmic bytes of the SAME launches the events timed: inputs, residuals and outputs once + weights once (ffp_*_profile_bytes)
                    "algorithmic_bytes": round(d.get("bytes", 0.0) / max(d["launches"], 1)),
                    # HBM bytes per launch from the committed PMC passes; the same command printed its launches' algorithmic figures (ffp_conv_totals_*),
                    # so `traffic` and `traffic_population.algorithmic_bytes` are per-launch means over ONE set of launches
                    "traffic": (round(pmc[d["variant"]]["hbm_bytes_per_launch"]) if d["variant"] in pmc else None),
                    "traffic_population": ({k: pmc[d["variant"]].get(k) for k in ("launches", "algorithmic_bytes_per_launch", "flops_per_launch", "traffic_over_algorithmic",
                                                                                 "launches_counted_by_the_library")} if d["variant"] in pmc else None),
                    "traffic_source": (f"profiles/{pmc_src} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)" if d["variant"] in pmc else None),
                    "events": f"extra profiled pass after the timed region: lane 0 alone, {getattr(main_r, 'profile_steps', 0)} steps, last detection group + last full SR batch",
                    "launches": d["launches"], "avg_launch_us": round(d["ms"] * 1e3 / max(d["launches"], 1), 2),
                    "flops_per_launch": d["flops"] / max(d["launches"], 1),
                    "sustained_mfma_loop": ({"tflops": SUSTAINED_TFLOPS[dtp], "frac_of_it": round(ach / SUSTAINED_TFLOPS[dtp], 4),
                                             "source": "profiles/r03_mfma_sustained_peak_probe.txt (bare MFMA loop, random operands, this chip)"} if dtp in SUSTAINED_TFLOPS else None)}
        res = {
            "metric": "end-to-end 4K frames/sec (SAHI+YOLOv11s+ESRGAN×4)", "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if (world > 1 and B == 1) else "weak", "vs_baseline": None,
            "dtype": f"{args.det_precision}(detect)+f16(sr)" if args.sr_crops > 0 else args.det_precision, "data": "synthetic",
            "config": {"workload": main_r.describe(),
                       "span": ("frames resident in HBM -> results in host memory" if main_r.resident else
                                "frame in pinned host memory -> upload (copy stream) -> detect -> merge -> SR -> detections + enhanced crops in host memory"),
                       "frames_per_step": B, "det_batch_frames": main_r.DB, "sr_batch_frames": main_r.SB, "mode": main_r.mode,
                       "lanes": main_r.n_lanes,
                       "lanes_note": ("detection groups are dealt round-robin to %d lanes (one detector + enhancer handle pair and one host thread each) whose kernels "
                                      "interleave on the card; the kernel events behind roofline / conv_profile_last_step / stage_ms_last_call come from an extra pass "
                                      "AFTER the timed region, on lane 0 alone; per_rank stage times are summed over the lanes' host threads"
                                      % main_r.n_lanes) if main_r.n_lanes > 1 else None,
                       "collective": {"backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None, "ranks": world,
                                      "communicator_ranks": (dist.get_world_size() if world > 1 else 1),
                                      "all_gathers_per_group": int(bool(main_state.get("gathered"))),
                                      "exchange_ms_per_step_by_rank": [r_["stage_ms_per_step"].get("exchange", 0.0) for r_ in main_report]},
                       "hbm_bytes_peak": hbm_peak[0], "hbm_bytes_in_use_after_headline": hbm_headline, "hbm_bytes_headline_handles": main_mem,
                       "hbm_note": "hbm_bytes_peak: the most bytes in use on the card (hipMemGetInfo: all allocators) at the sampling points — after the headline's timed loop "
                                   "and at the end of every secondary row; *_handles: packed weights + resident plans of the rows' own handles (ffp_det_mem_bytes / ffp_sr_mem_bytes)",
                       "parallelism": (f"items of {B} frame(s) in {world} contiguous cost-balanced blocks; "
                                       + ("one all-gather per group" if main_state.get("gathered") else "whole frames per rank: no exchange needed")) if world > 1 else "single GPU",
                       "upload_bytes_per_group_per_rank": int(main_state.get("upload_bytes", 0)),
                       "sr_px_per_frame_mean": round(main_sr_px * world / max(args.steps * B, 1), 1),
                       "detections_last_frame": int(main_state.get("rows", np.zeros((0, 1))).shape[0]),
                       "det_graph_status": det_graph, "sr_plan_state": sr_state},
            "latency_ms_rank0": main_report[0]["latency_ms"],
            "per_rank": main_report,
            "stage_ms_last_call": {k: round(v, 3) for k, v in stage_ms.items()},
            "sr_ms_last_call": round(sr_ms, 3) if sr_ms is not None else None,
            "conv_profile_last_step": [{"kernel": p["variant"], "stage": p["stage"], "ms": round(p["ms"], 3), "launches": p["launches"],
                                        "tflops": round(p["flops"] / max(p["ms"], 1e-9) / 1e9, 2)} for p in prof],
            "roofline": roof,
            "secondary": secondary,
        }
        if world == 1 and args.verify:
            res["verify"] = verify_rows(args, ctx, pipes, steps=max(25, args.steps))
            if not all(v["ok"] for v in res["verify"].values()):
                print("[bench] --verify: the pipelined loop's outputs differ from the synchronous calls", file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            base_r = Runner(args, ctx, pipes=pipes, resident=True, lanes=1)
            sf, _ = base_r.upload(0, 1)
            d, c, _, _ = base_r.pipe.detect(sf, H, W, 1)
            pre = torch.cat([d[k, :int(c[k])] for k in range(d.shape[0])], 0).cpu().numpy()
            boxes = main_state.get("boxes_frame0", main_state.get("boxes", np.zeros((0, 4), np.int32)))      # the crops the GPU enhanced for THIS frame (frame 0 of the timed loop)
            res["cpu_baseline"] = cpu_baseline(args, ctx["det_w"], ctx["sr_w"], ctx["host_frames"][0], boxes, pre)
        if args.conv_totals:
            res["conv_totals"] = _lib.conv_totals()        # last thing before the line is printed: every launch of the process is in it
        print(json.dumps(res), flush=True)
    if world > 1:
        try:                                        # a secondary row that failed on some ranks only (out of memory on a shared card) leaves the
            dist.barrier()                          # ranks out of step; the line is printed by then, and the run must still end
            dist.destroy_process_group()
        except Exception as e:      # noqa: BLE001
            print(f"[bench] rank {rank}: final barrier: {type(e).__name__}", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
