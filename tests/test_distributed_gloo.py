"""world_size-2 (and 3) CPU test of the multi-GPU orchestration over gloo: contiguous item sharding + the single
all-gather reproduce the single-process detection list exactly, every rank ends with identical data and the replicated
merge (oracle as stand-in for the GPU kernel) gives the same result on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ffp_amd  # noqa: F401
from ffp_amd import pipeline

MAX_DET, STRIDE = 12, 21


def fake_item_dets(item_idx: int, tile):
    """Deterministic stand-in for the detector: a few boxes per item, already truncated + shifted to frame coords."""
    rng = np.random.default_rng(1000 + item_idx)
    n = int(rng.integers(0, MAX_DET))
    d = np.zeros((MAX_DET, STRIDE), np.float32)
    x0, y0, x1, y1 = tile
    for k in range(n):
        w, h = rng.integers(8, 60, 2)
        cx, cy = rng.integers(x0, max(x0 + 1, x1 - w)), rng.integers(y0, max(y0 + 1, y1 - h))
        d[k, :4] = (cx, cy, cx + w, cy + h)
        d[k, 4] = rng.uniform(0.5, 1.0)
        d[k, 6:] = rng.standard_normal(STRIDE - 6)
    return d, n


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = pipeline.PipeConfig(slice_h=256, slice_w=256, imgsz=256)
        items = pipeline.frame_items(540, 960, cfg, n_frames)
        L = pipeline.Layout(items, n_frames, world, pipeline.item_costs(items, cfg.imgsz))
        lo, hi = L.bounds[rank]
        local = torch.zeros((L.per, MAX_DET, STRIDE))
        counts = torch.zeros((L.per,), dtype=torch.int32)
        for j, i in enumerate(range(lo, hi)):
            d, n = fake_item_dets(i, items[i])
            local[j] = torch.from_numpy(d)
            counts[j] = n
        g, gc = pipeline.exchange_detections(local, counts, world)
        q.put((rank, g.numpy().copy(), gc.numpy().copy(), len(items), L.per))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_frames", [(2, 1), (2, 2), (3, 2)])
def test_sharded_exchange_reproduces_single_process(world, n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    n_items, per = got[0][3], got[0][4]
    cfg = pipeline.PipeConfig(slice_h=256, slice_w=256, imgsz=256)
    items = pipeline.frame_items(540, 960, cfg, n_frames)
    L = pipeline.Layout(items, n_frames, world, pipeline.item_costs(items, cfg.imgsz))
    assert n_items == len(items) and per == L.per
    # blocks are contiguous, cover every item once, and no block is heavier than the optimum for a contiguous split allows
    assert [b[0] for b in L.bounds] + [n_items] == [0] + [b[1] for b in L.bounds]
    costs = pipeline.item_costs(items, cfg.imgsz)
    heaviest = max(costs[lo:hi].sum() for lo, hi in L.bounds)
    assert heaviest <= costs.sum() / world + costs.max()
    exp = np.zeros((world * per, MAX_DET, STRIDE), np.float32)
    expc = np.zeros((world * per,), np.int32)
    for i in range(n_items):
        exp[L.slot(i)], expc[L.slot(i)] = fake_item_dets(i, items[i])
    for _, g, gc, _, _ in got:                      # every rank holds the same buffer, in SAHI's order (holes have count 0)
        assert np.array_equal(g, exp) and np.array_equal(gc, expc)
    slots = [L.slot(i) for i in range(n_items)]
    assert slots == sorted(slots)
    # replicated merge: same input on every rank -> same output; a frame is a contiguous slot range
    from oracle import sahi_ref
    for f in range(n_frames):
        s0, ns = L.frame_slots(f, True)
        rows = np.concatenate([exp[k, :expc[k]] for k in range(s0, s0 + ns)], 0)
        ref_rows = np.concatenate([fake_item_dets(i, items[i])[0][:fake_item_dets(i, items[i])[1]] for i in range(f * L.ipf, (f + 1) * L.ipf)], 0)
        assert np.array_equal(rows, ref_rows)
        dets = [sahi_ref.Det(r[:4].tolist(), r[4], 0) for r in rows]
        out = sahi_ref.postprocess(dets, "GREEDYNMM", "IOS", 0.5) if len(dets) > 1 else dets
        ys = np.asarray([d.bbox[1] for d in out])
        assert ((ys >= f * 540) & (ys < (f + 1) * 540)).all()        # frames of a super-frame never interact
        if L.owner(f) >= 0:                                            # whole frame on one rank: its local slots hold the same rows
            a0, an = L.frame_slots(f, False, L.owner(f))
            lo = L.bounds[L.owner(f)][0]
            assert [lo + a0 + k for k in range(an)] == list(range(f * L.ipf, (f + 1) * L.ipf))


def test_layout_alignment_and_lpt():
    """Weak scaling with whole frames per rank needs no exchange; ONE frame over the ranks does; LPT placement of SR crops is
    deterministic and balanced."""
    cfg = pipeline.PipeConfig()
    for world in (2, 4, 8):
        items = pipeline.frame_items(2160, 3840, cfg, world)
        L = pipeline.Layout(items, world, world, pipeline.item_costs(items, 512))
        assert L.aligned and [L.owner(f) for f in range(world)] == list(range(world))
        assert L.rows_needed(1, 2160) == (2160, 4320)                  # a rank uploads its own frame only
        one = pipeline.frame_items(2160, 3840, cfg, 1)
        L1 = pipeline.Layout(one, 1, world, pipeline.item_costs(one, 512))
        assert not L1.aligned and L1.owner(0) == -1
        assert all(L1.rows_needed(r, 2160) == (0, 2160) for r in range(world))     # crops may come from anywhere in a spread frame
        c1 = pipeline.item_costs(one, 512) / (512 * 512)               # 60 unit slices + one 0.5625 full-frame pass
        assert max(c1[l:h].sum() for l, h in L1.bounds) <= -(-60 // world) + 0.5625
    sizes = pipeline.sr_crop_sizes(32, 5).astype(np.int64)
    a = pipeline.lpt_assign(sizes ** 2, 8)
    assert np.array_equal(a, pipeline.lpt_assign(sizes ** 2, 8))
    loads = np.asarray([(sizes[a == r] ** 2).sum() for r in range(8)])
    assert loads.max() <= (sizes ** 2).sum() / 8 + (sizes ** 2).max()
    assert set(a.tolist()) <= set(range(8))


def _worker_spread(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = pipeline.PipeConfig(slice_h=256, slice_w=256, imgsz=256)
        items = pipeline.frame_items(540, 960, cfg, n_frames)
        L = pipeline.Layout(items, n_frames, world, pipeline.item_costs(items, cfg.imgsz), spread=True)
        mine = L.local_items(rank)
        lo, hi = L.frame_bounds[rank]
        b = hi - lo
        assert len(mine) == n_frames * b and L.local_slots() == n_frames * L.per
        # what FramePipeline.detect does: dense run over the rank's items, then padding to `per` slots per frame
        dense = torch.zeros((n_frames * b, MAX_DET, STRIDE))
        dcount = torch.zeros((n_frames * b,), dtype=torch.int32)
        for k in range(len(mine)):
            f, j = divmod(k, b)
            i = f * L.ipf + lo + j
            assert np.array_equal(mine[k], items[i])
            d, n = fake_item_dets(i, items[i])
            dense[k] = torch.from_numpy(d)
            dcount[k] = n
        local = torch.zeros((L.local_slots(), MAX_DET, STRIDE))
        counts = torch.zeros((L.local_slots(),), dtype=torch.int32)
        local.view(n_frames, L.per, MAX_DET, STRIDE)[:, :b].copy_(dense.view(n_frames, b, MAX_DET, STRIDE))
        counts.view(n_frames, L.per)[:, :b].copy_(dcount.view(n_frames, b))
        g, gc = pipeline.exchange_detections(local, counts, world)           # ONE all-gather for the whole group of frames
        g, gc = pipeline.frame_major(g, gc, world, n_frames, L.per)
        q.put((rank, g.numpy().copy(), gc.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 3), (3, 2)])
def test_spread_group_one_gather_per_group(world, n_frames):
    """The north_star split with several frames in flight (strong scaling): every frame's items over all ranks, ONE all-gather per
    group of frames; after the frame-major reorder every rank holds, per frame, a contiguous slot range with that frame's detections
    in SAHI's order — the input of the replicated merge."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_spread, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = pipeline.PipeConfig(slice_h=256, slice_w=256, imgsz=256)
    items = pipeline.frame_items(540, 960, cfg, n_frames)
    costs = pipeline.item_costs(items, cfg.imgsz)
    L = pipeline.Layout(items, n_frames, world, costs, spread=True)
    assert not L.aligned and all(L.owner(f) == -1 for f in range(n_frames))
    assert all(L.rows_needed(r, 540) == (0, n_frames * 540) for r in range(world))
    # every rank's share of a frame is a contiguous cost-balanced block of THAT frame
    fb = L.frame_bounds
    assert [b[0] for b in fb] + [L.ipf] == [0] + [b[1] for b in fb]
    assert max(costs[lo:hi].sum() for lo, hi in fb) <= costs[:L.ipf].sum() / world + costs[:L.ipf].max()
    exp = np.zeros((n_frames * world * L.per, MAX_DET, STRIDE), np.float32)
    expc = np.zeros((n_frames * world * L.per,), np.int32)
    for i in range(len(items)):
        exp[L.slot(i)], expc[L.slot(i)] = fake_item_dets(i, items[i])
    slots = [L.slot(i) for i in range(len(items))]
    assert slots == sorted(slots) and len(set(slots)) == len(slots)
    for _, g, gc in got:
        assert np.array_equal(g, exp) and np.array_equal(gc, expc)
    for f in range(n_frames):
        s0, ns = L.frame_slots(f, True)
        assert (s0, ns) == (f * world * L.per, world * L.per)
        rows = np.concatenate([exp[k, :expc[k]] for k in range(s0, s0 + ns)], 0)
        ref_rows = np.concatenate([fake_item_dets(i, items[i])[0][:fake_item_dets(i, items[i])[1]] for i in range(f * L.ipf, (f + 1) * L.ipf)], 0)
        assert np.array_equal(rows, ref_rows)


def _worker_spread8(rank, world, port, n_frames, q):
    """The BASELINE config-4 shape on CPU: 4K frames, 512 / 0.2 slices (61 items per frame), 5-frame detection groups over 8 ranks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = pipeline.PipeConfig()
        items = pipeline.frame_items(2160, 3840, cfg, n_frames)
        L = pipeline.Layout(items, n_frames, world, pipeline.item_costs(items, cfg.imgsz), spread=True)
        mine = L.local_items(rank)
        lo, hi = L.frame_bounds[rank]
        b = hi - lo
        local = torch.zeros((L.local_slots(), MAX_DET, STRIDE))
        counts = torch.zeros((L.local_slots(),), dtype=torch.int32)
        for k in range(len(mine)):
            f, j = divmod(k, b)
            i = f * L.ipf + lo + j
            d, n = fake_item_dets(i, items[i])
            local[f * L.per + j] = torch.from_numpy(d)
            counts[f * L.per + j] = n
        g, gc = pipeline.exchange_detections(local, counts, world)
        g, gc = pipeline.frame_major(g, gc, world, n_frames, L.per)
        sizes = pipeline.sr_crop_sizes(32, seed=1000).astype(np.int64)
        own = pipeline.lpt_assign(sizes ** 2, world) == rank                # this rank's crops of frame 0
        q.put((rank, int(gc.sum()), float(g.double().sum()), int(own.sum()), int((sizes[own] ** 2).sum())))
    finally:
        dist.destroy_process_group()


def test_spread_world8_five_frame_groups():
    """VERDICT r3 item 9 — insurance for the day an 8-GPU node runs `--frames-per-step 1`: world 8, 5-frame groups, 61 items per 4K frame.
    Slot order, padding, one gather per group, cost balance of the contiguous blocks and of the LPT crop placement."""
    world, n_frames = 8, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_spread8, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = pipeline.PipeConfig()
    items = pipeline.frame_items(2160, 3840, cfg, n_frames)
    costs = pipeline.item_costs(items, cfg.imgsz)
    L = pipeline.Layout(items, n_frames, world, costs, spread=True)
    assert L.ipf == 61 and len(items) == 305
    fb = L.frame_bounds
    assert [b[0] for b in fb] + [61] == [0] + [b[1] for b in fb]          # contiguous, complete
    assert L.per == 8 and L.local_slots() == 40                            # 8 slots per frame and rank: 8 x 5 x 8 x 25 KB = 8 MB per gather at max_det 300
    blk = np.asarray([costs[lo:hi].sum() for lo, hi in fb]) / (512 * 512)
    assert blk.max() <= 8.0 and blk.max() / blk.mean() <= 1.10, blk       # heaviest block within 10 % of the mean (60.56 slice units / 8 = 7.57)
    assert sorted(hi - lo for lo, hi in fb) == [5] + [8] * 7              # the last rank: 4 slices + the full-frame pass
    slots = [L.slot(i) for i in range(len(items))]
    assert slots == sorted(slots) and len(set(slots)) == len(slots) and max(slots) < n_frames * world * L.per
    tot_n = sum(fake_item_dets(i, items[i])[1] for i in range(len(items)))
    tot_v = sum(float(fake_item_dets(i, items[i])[0].astype(np.float64).sum()) for i in range(len(items)))
    for r, n, v, _, _ in got:                                              # every rank holds every frame's detections after ONE gather
        assert n == tot_n and abs(v - tot_v) < 1e-3 * max(1.0, abs(tot_v)), (r, n, tot_n)
    sizes = pipeline.sr_crop_sizes(32, seed=1000).astype(np.int64)
    assert sum(g[3] for g in got) == 32 and sum(g[4] for g in got) == int((sizes ** 2).sum())       # every crop placed exactly once
    loads = np.asarray([g[4] for g in got], np.float64)
    assert loads.max() <= loads.mean() + (sizes ** 2).max()               # LPT bound
