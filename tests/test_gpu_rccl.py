"""RCCL next to libffp.so in one process (backend "nccl" is RCCL on ROCm): the exchange of fixed-cap detections that the
multi-GPU path performs (pipeline.exchange_detections -> all_gather_into_tensor) on device buffers libffp.so wrote, with ONE
HIP runtime shared by torch, RCCL and libffp (ffp_amd/_lib.py). Runs with world_size 1 on a single-GPU box (communicator
creation, collective launch and stream interop are exercised; the ring is trivial) and with world_size 2 wherever two GPUs are
visible — the 8-GPU scaling run itself is the driver's."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
import ffp_amd
from ffp_amd import _lib, pipeline, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", device_id=dev)
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
assert len(libs) == 1, libs                                    # one HIP runtime under torch, RCCL and libffp
H, W = 540, 960
cfg = pipeline.PipeConfig(slice_h=256, slice_w=256, imgsz=256, conf=0.25, sr_crops=0)
pipe = pipeline.FramePipeline(synth.yolo11_pose_weights("n"), None, cfg, arch="n", device=rank, det_precision=_lib.PREC_F32, rank=rank, world=world)
frame = torch.from_numpy(synth.synthetic_frame(H, W, seed=3)).to(dev)
torch.cuda.synchronize()
dets, counts, L, gathered = pipe.detect(frame, H, W, 1, exchange="always" if world > 1 else "auto")
if world == 1:                                                 # force the collective through RCCL on one rank too
    g, gc = torch.empty_like(dets), torch.empty_like(counts)
    dist.all_gather_into_tensor(g, dets.contiguous()); dist.all_gather_into_tensor(gc, counts.contiguous())
    torch.cuda.synchronize()
    assert torch.equal(g, dets) and torch.equal(gc, counts)
rows, n = pipe.merge_frame_of(dets, counts, L, 0, gathered)
n = pipe.merged_count(n)
ref = _lib.Detector(synth.yolo11_pose_weights("n"), arch="n", device=rank, precision=_lib.PREC_F32).sliced_predict(
    synth.synthetic_frame(H, W, seed=3), 256, 256, 0.2, 0.2, True, 256, 0.25, 0.7, 300, "GREEDYNMM", "IOS", 0.5, False)
got = rows[:n].cpu().numpy()
assert got.shape == ref.shape and np.array_equal(got, ref), (got.shape, ref.shape)   # sharded + all-gather + merge == single-process fused call
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", n)
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [1, 2])
def test_rccl_all_gather_of_libffp_detections(gpu_lib, world):
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"{world} GPUs needed, {torch.cuda.device_count()} visible")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
        assert "ok" in out


SPREAD_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
import ffp_amd
from ffp_amd import _lib, pipeline, synth
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)                                       # every rank on the ONE visible GPU: the code path of N ranks, no RCCL peers needed
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
H, W, NF = 540, 960, 3
cfg = pipeline.PipeConfig(slice_h=256, slice_w=256, imgsz=256, conf=0.25, sr_crops=0)
Wd = synth.yolo11_pose_weights("n")
pipe = pipeline.FramePipeline(Wd, None, cfg, arch="n", device=0, det_precision=_lib.PREC_F32, rank=rank, world=world)
frames = [synth.synthetic_frame(H, W, seed=3 + f) for f in range(NF)]
sf = torch.from_numpy(np.concatenate(frames, 0)).to(dev)
torch.cuda.synchronize()
L = pipe.layout(H, W, NF, "spread")
assert L.spread and L.local_slots() == NF * L.per and len(L.local_items(rank)) == NF * (L.frame_bounds[rank][1] - L.frame_bounds[rank][0])
dets, counts, L2, gathered = pipe.detect(sf, H, W, NF, mode="spread")          # ONE all-gather for the three frames
assert gathered and L2 is L and dets.shape[0] == NF * world * L.per
ref_det = _lib.Detector(Wd, arch="n", device=0, precision=_lib.PREC_F32)
for f in range(NF):
    rows, n = pipe.merge_frame_of(dets, counts, L, f, gathered)
    n = pipe.merged_count(n)
    got = rows[:n].cpu().numpy().copy()
    got[:, [1, 3]] -= f * H                                    # boxes of frame f live at rows [f * H, (f + 1) * H) of the super-frame
    if got.shape[1] > 6:
        got[:, 7::3] -= f * H                                  # keypoint y
    ref = ref_det.sliced_predict(frames[f], 256, 256, 0.2, 0.2, True, 256, 0.25, 0.7, 300, "GREEDYNMM", "IOS", 0.5, False)
    # boxes, scores and classes bit for bit; keypoints were shifted by (tile origin + f * H) in fp32, so their y differs from the
    # single-frame call by the rounding of that sum (<= 1 ulp of ~1000: 6e-5)
    assert got.shape == ref.shape and np.array_equal(got[:, :6], ref[:, :6]), (f, got.shape, ref.shape)
    assert np.allclose(got[:, 6:], ref[:, 6:], rtol=0, atol=2e-4), (f, np.abs(got[:, 6:] - ref[:, 6:]).max())
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "spread ok")
"""


def test_spread_mode_two_ranks_on_one_gpu_matches_single_process(gpu_lib):
    """The north_star split with several frames in flight, run by the REAL kernels: two ranks (gloo, both on the one visible GPU) each
    detect their contiguous share of EVERY frame of a 3-frame group, exchange once, reorder frame-major, and every rank's replicated
    merge of every frame equals the single-process fused `ffp_sliced_predict` of that frame bit for bit (exact fp32: the arithmetic
    whose results do not depend on batch mates)."""
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", SPREAD_WORKER % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
        assert "spread ok" in out
