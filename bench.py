#!/usr/bin/env python3
"""bench.py — end-to-end 4K frames/sec of the hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic 4K frames resident in HBM (batch = n_gpus frames, one
frame at N=1): SAHI 512x512 / 0.2 slicing (60 slices + the full-frame pass), YOLO11s-pose on every item at net input
512, per-item NMS, int-truncate + shift, [N>1: RCCL all-gather of the fixed-cap boxes], SAHI GREEDYNMM/IOS/0.5 merge,
merged detections to host, Real-ESRGAN x4 on 32 crops per frame, enhanced crops to host.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and, at N=1, `cpu_baseline`.
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


def parse():
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
                    help="f32: exact-fp32 MFMA; f32x3: fp32 storage, fp16 hi/lo split products (fp32-grade); f16: speed mode")
    ap.add_argument("--sr-crops", type=int, default=32, help="crops enhanced per frame (0: config 2, detection only)")
    ap.add_argument("--pp-type", default="GREEDYNMM", choices=["GREEDYNMM", "NMS"])
    ap.add_argument("--class-agnostic", action="store_true", help="merge across classes (the reference's eval setting with NMS)")
    ap.add_argument("--conf", type=float, default=0.5)
    ap.add_argument("--distinct-frames", type=int, default=2)
    ap.add_argument("--sr-batch-frames", type=int, default=2,
                    help="frames whose crops are enhanced together as one ragged Real-ESRGAN batch (1: per frame)")
    ap.add_argument("--det-batch-frames", type=int, default=2,
                    help="consecutive steps whose frames are detected together as one ragged batch of slices (1: per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(args, det_w, sr_w, frame, crop_boxes, premerge_rows):
    """The oracle (CPU restatement, torch fp32) timed on a bounded sample of the same workload and scaled to one frame."""
    import torch
    from oracle import rrdbnet_ref, sahi_ref, ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    H, W = frame.shape[:2]
    ref = Yolo11PoseRef(det_w, args.arch)
    boxes = sahi_ref.get_slice_bboxes(H, W, args.slice, args.slice, args.overlap, args.overlap)
    n_sample = 3
    idx = np.linspace(0, len(boxes) - 1, n_sample).astype(int)
    ultra_post.predict(ref, frame[:args.slice, :args.slice], args.imgsz, args.conf)   # warm-up (scripts/inference_time.py:46-52)
    t0 = time.perf_counter()
    for i in idx:
        x0, y0, x1, y1 = boxes[i]
        ultra_post.predict(ref, frame[y0:y1, x0:x1], args.imgsz, args.conf)
    t_slice = (time.perf_counter() - t0) / n_sample
    t0 = time.perf_counter()
    ultra_post.predict(ref, frame, args.imgsz, args.conf)
    t_full = time.perf_counter() - t0
    t0 = time.perf_counter()
    dets = [sahi_ref.Det(r[:4].tolist(), r[4], int(r[5])) for r in premerge_rows]
    if len(dets) > 1:
        sahi_ref.postprocess(dets, args.pp_type, "IOS", 0.5, False)
    t_merge = time.perf_counter() - t0
    t_sr_frame, sr_sample = 0.0, "none"
    if args.sr_crops > 0:
        net = rrdbnet_ref.RRDBNetRef(sr_w, 4, 23)
        sample = [b for b in crop_boxes if (b[2] - b[0]) <= 48][:3] or [crop_boxes[0]]
        px = 0
        t0 = time.perf_counter()
        for b in sample:
            rrdbnet_ref.enhance(net, frame[b[1]:b[3], b[0]:b[2]][..., ::-1].copy())
            px += int(b[2] - b[0]) * int(b[3] - b[1])
        t_sr = time.perf_counter() - t0
        tot_px = sum(int(b[2] - b[0]) * int(b[3] - b[1]) for b in crop_boxes)
        t_sr_frame = t_sr / px * tot_px
        sr_sample = f"{len(sample)} crops ({px} px of {tot_px})"
    t_frame = t_slice * len(boxes) + t_full + t_merge + t_sr_frame
    return {"value": 1.0 / t_frame, "unit": "frames/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{n_sample} of {len(boxes)} slices + full-frame pass + merge of {len(dets)} boxes + SR on {sr_sample}, "
                      f"scaled to one frame ({t_frame:.1f} s/frame: det {t_slice * len(boxes) + t_full:.1f}, merge {t_merge:.3f}, sr {t_sr_frame:.1f})"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import ffp_amd  # noqa: F401
    from ffp_amd import _lib, pipeline, synth

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

    H, W, B = args.height, args.width, world
    cfg = pipeline.PipeConfig(slice_h=args.slice, slice_w=args.slice, overlap=args.overlap, imgsz=args.imgsz, conf=args.conf,
                              pp_type=args.pp_type, class_agnostic=args.class_agnostic, sr_crops=args.sr_crops)
    det_w = synth.yolo11_pose_weights(args.arch)
    sr_w = synth.rrdbnet_weights(4, 23) if args.sr_crops > 0 else None
    pipe = pipeline.FramePipeline(det_w, sr_w, cfg, arch=args.arch, device=local_rank,
                                  det_precision={"f16": _lib.PREC_F16, "f32x3": _lib.PREC_F32X3, "f32": _lib.PREC_F32}[args.det_precision], sr_half=True,
                                  rank=rank, world=world)

    # synthetic frames, resident in HBM before the timed region. Detection runs on groups of DB consecutive steps: the
    # B frames of each step of a group are stacked into one super-frame whose slices form ONE ragged batch (the small
    # stride-16/32 layers of a single frame's 61 items do not fill 256 CUs: 6.8 -> 6.25 ms per frame at DB = 2).
    DB = max(1, args.det_batch_frames)
    host_frames = [synth.synthetic_frame(H, W, seed=i) for i in range(max(1, args.distinct_frames))]
    supers = {}      # frames per super-frame -> list of resident stacks

    def stacks(nf):
        if nf not in supers:
            lst = []
            for s_ in range(len(host_frames)):
                sf_ = np.concatenate([host_frames[(s_ + f) % len(host_frames)] for f in range(nf)], 0)
                lst.append(torch.from_numpy(sf_).to(dev))
            supers[nf] = lst
        return supers[nf]

    items_per_frame = pipeline.frame_items(H, W, cfg, 1).shape[0]
    sizes = pipeline.sr_crop_sizes(max(args.sr_crops, 1), seed=0)
    host_rows = torch.empty((cfg.merge_cap, pipe.stride), dtype=torch.float32).pin_memory()
    sr_bytes = int(sum(((int(s) * 4) ** 2 * 3 + 15) // 16 * 16 for s in sizes[:args.sr_crops]))
    host_sr = torch.empty((max(sr_bytes, 16) * max(args.sr_batch_frames, 1),), dtype=torch.uint8).pin_memory()
    state = {}

    pending = {}     # super-resolution batch still running on the enhancer's stream
    queue = []       # (frame tensor, crop boxes) of frames waiting for their SR batch

    def drain_sr():
        if pending:
            pipe.wait_sr()
            out = pending.pop("out")
            host_sr[:out.numel()].copy_(out)                       # enhanced crops -> host

    def flush_sr(slot):
        """Enhance the queued frames' crops as ONE ragged batch (asynchronously, on the enhancer's stream)."""
        if not queue:
            return
        drain_sr()
        out, offs = pipe.enhance_crops_multi([q[0] for q in queue], H, W, [q[1] for q in queue], slot=slot)
        pending["out"] = out
        queue.clear()

    group = {}

    def step(i, n_total, profile=False):
        """Frame(s) of step i. The first step of a group detects the whole group's frames (detector stream); every step
        merges its own frames and queues their crops, which are enhanced on the enhancer's stream while later frames are
        detected. The last group of the timed loop is the profiled one (per-launch events, eager launches)."""
        g0 = (i // DB) * DB
        gsz = min(DB, n_total - g0)                               # steps in this group
        if i == g0:
            sf = stacks(B * gsz)[(i // DB) % len(host_frames)]
            if profile:
                pipe.det.set_profile(True)
            dets, counts, _ = pipe.detect(sf, H, W, B * gsz)
            if profile:
                pipe.det.set_profile(False)
            group.update(sf=sf, dets=dets, counts=counts)
        sf, dets, counts = group["sf"], group["dets"], group["counts"]
        last = profile and i == n_total - 1
        for fb in range(B):
            f = (i - g0) * B + fb                                 # frame index inside the group's super-frame
            if fb % world != rank:
                continue
            rows_d, n_d = pipe.merge_frame(dets, counts, f * items_per_frame, items_per_frame)
            n = int(n_d.item())
            host_rows[:n].copy_(rows_d[:n])                       # merged detections -> host
            rows = host_rows[:n].numpy().copy()
            rows[:, [1, 3]] -= f * H
            state["rows"] = rows
            if args.sr_crops > 0:
                boxes = pipeline.crop_boxes_for_sr(rows, H, W, args.sr_crops, sizes, seed=i)
                state["boxes"] = boxes
                queue.append((sf[f * H:(f + 1) * H], boxes))
                if len(queue) >= args.sr_batch_frames or last:
                    if last:
                        drain_sr()
                        pipe.sr.set_profile(True)
                    flush_sr(slot=(i // max(args.sr_batch_frames, 1)) & 1)
        if last:
            drain_sr()
            torch.cuda.synchronize(dev)
            if pipe.sr is not None:
                pipe.sr.set_profile(False)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # untimed setup: build (and graph-capture) the detector plan of every group size the loops below will meet
    for gsz in sorted({min(DB, n - g0) for n in (args.warmup, args.steps) for g0 in range(0, n, DB)}):
        for _ in range(2):
            pipe.detect(stacks(B * gsz)[0], H, W, B * gsz)
    for i in range(args.warmup):
        step(i, args.warmup)
    flush_sr(slot=0)
    drain_sr()
    barrier()
    last_g0 = ((args.steps - 1) // DB) * DB
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, args.steps, profile=(i >= last_g0))   # the last group also brackets every conv launch with HIP events
    flush_sr(slot=0)
    drain_sr()                                      # the last frames' crops are part of the timed work
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        fps = B * args.steps / dt
        prof = [dict(p, stage="det") for p in pipe.det.profile()]
        if pipe.sr is not None:
            prof += [dict(p, stage="sr") for p in pipe.sr.profile()]
        prof.sort(key=lambda p: -p["ms"])
        roof = None
        pmc = {}
        try:    # HBM bytes per launch from the committed PMC passes (profiles/r01_pmc_traffic.json), keyed by kernel variant
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
                pmc = json.load(fh).get("kernels", {})
        except Exception:
            pmc = {}
        if prof:
            d = prof[0]
            dtp = d["variant"].split("_")[0]
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
            roof = {"bound": "mfma", "kernel": f"{'conv_rows_kernel' if d['variant'].endswith('_rows') else 'conv_mfma_kernel'}<{d['variant']}> ({d['stage']})", "achieved": round(ach, 2),
                    "peak": PEAK_TFLOPS[dtp], "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS[dtp], 4),
                    "traffic": (round(pmc[d["variant"]]["hbm_bytes_per_launch"]) if d["variant"] in pmc else None),
                    "traffic_source": ("profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)" if d["variant"] in pmc else None),
                    "launches": d["launches"], "avg_launch_us": round(d["ms"] * 1e3 / max(d["launches"], 1), 2),
                    "flops_per_launch": d["flops"] / max(d["launches"], 1)}
        stage_ms = pipe.det.last_ms()
        res = {
            "metric": "end-to-end 4K frames/sec (SAHI+YOLOv11s+ESRGAN×4)", "value": round(fps, 3), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": f"{args.det_precision}(detect)+f16(sr)" if args.sr_crops > 0 else args.det_precision, "data": "synthetic",
            "config": {"workload": f"{W}x{H} frame, YOLO11{args.arch}-pose (random-init), SAHI {args.slice}x{args.slice}/{args.overlap} "
                                   f"({items_per_frame - 1} slices + full frame), net input {args.imgsz}, conf {args.conf}, NMS 0.7, "
                                   f"{args.pp_type}/IOS/0.5{'/agnostic' if args.class_agnostic else ''} merge" + (f", Real-ESRGAN x4 on {args.sr_crops} crops/frame "
                                   f"({int((sizes[:args.sr_crops] ** 2).sum())} px)" if args.sr_crops > 0 else ", no SR"),
                       "frames_per_step": B, "det_batch_frames": DB, "sr_batch_frames": args.sr_batch_frames, "parallelism": f"items sharded over {world} rank(s), 1 all-gather" if world > 1 else "single GPU",
                       "detections_last_frame": int(state.get("rows", np.zeros((0, 1))).shape[0])},
            "stage_ms_last_call": {k: round(v, 3) for k, v in stage_ms.items()},
            "sr_ms_last_call": round(pipe.sr.last_ms(), 3) if pipe.sr is not None else None,
            "conv_profile_last_step": [{"kernel": p["variant"], "stage": p["stage"], "ms": round(p["ms"], 3), "launches": p["launches"],
                                        "tflops": round(p["flops"] / max(p["ms"], 1e-9) / 1e9, 2)} for p in prof],
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            d, c, _ = pipe.detect(stacks(1)[0][:H], H, W, 1)
            pre = torch.cat([d[k, :int(c[k])] for k in range(d.shape[0])], 0).cpu().numpy()
            res["cpu_baseline"] = cpu_baseline(args, det_w, sr_w, host_frames[0], state.get("boxes", np.zeros((0, 4), np.int32)), pre)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
