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
        cfg = pipeline.PipeConfig()
        items = pipeline.frame_items(540, 960, pipeline.PipeConfig(slice_h=256, slice_w=256), n_frames)
        lo, hi, per = pipeline.shard(len(items), rank, world)
        local = torch.zeros((per, MAX_DET, STRIDE))
        counts = torch.zeros((per,), dtype=torch.int32)
        for j, i in enumerate(range(lo, hi)):
            d, n = fake_item_dets(i, items[i])
            local[j] = torch.from_numpy(d)
            counts[j] = n
        g, gc = pipeline.exchange_detections(local, counts, world)
        q.put((rank, g.numpy().copy(), gc.numpy().copy(), len(items), per))
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
    items = pipeline.frame_items(540, 960, pipeline.PipeConfig(slice_h=256, slice_w=256), n_frames)
    assert n_items == len(items)
    exp = np.zeros((world * per, MAX_DET, STRIDE), np.float32)
    expc = np.zeros((world * per,), np.int32)
    for i in range(n_items):
        exp[i], expc[i] = fake_item_dets(i, items[i])
    for _, g, gc, _, _ in got:                      # every rank holds the same, globally ordered buffer
        assert np.array_equal(g, exp) and np.array_equal(gc, expc)
    # replicated merge: same input on every rank -> same output; per-frame ranges are contiguous item ranges
    from oracle import sahi_ref
    ipf = n_items // n_frames
    for f in range(n_frames):
        rows = np.concatenate([exp[i, :expc[i]] for i in range(f * ipf, (f + 1) * ipf)], 0)
        dets = [sahi_ref.Det(r[:4].tolist(), r[4], 0) for r in rows]
        out = sahi_ref.postprocess(dets, "GREEDYNMM", "IOS", 0.5) if len(dets) > 1 else dets
        ys = np.asarray([d.bbox[1] for d in out])
        assert ((ys >= f * 540) & (ys < (f + 1) * 540)).all()        # frames of a super-frame never interact
