"""bench.py's pipelined loop against synchronous calls, byte for byte (`bench.py --verify` runs the same check and prints it in the JSON line).

The timed loop keeps three streams busy at once — the next detection group's frames upload into a ring of device slots while this group is
detected and the previous ten frames' crops are enhanced — and with `--lanes 2` two host threads drive two handle pairs. Nothing in the frames/s
figure says that the bytes coming out are the ones the reference's one-call-at-a-time order (docs sahi/predict.py:226,270; utils/enhancer.py:
158-176) would produce. Here every frame of a 25-step loop (DB = 5, SB = 10, the headline's batching) is recomputed with one call at a time on
fresh buffers, a device synchronisation after each call, and ONE frame's crops per enhancer call: merged rows, crop boxes, every enhanced crop
and (with JPEG in the span) every output file must be identical."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 25


@pytest.fixture(scope="module")
def bench_mod():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def shared(gpu_lib, bench_mod):
    """BASELINE config 3 (the headline's defaults) with two lanes' handle pairs, built once for the module."""
    import torch
    torch.cuda.set_device(0)
    args = bench_mod.parse([])
    assert args.det_batch_frames == 5 and args.sr_batch_frames == 10
    ctx = bench_mod.make_ctx(args)
    owner = bench_mod.Runner(args, ctx, lanes=2)
    yield args, ctx, owner.pipes
    owner.close()


@pytest.mark.parametrize("lanes,jpeg_io", [(1, False), (2, False), (1, True)], ids=["one_lane", "two_lanes", "one_lane_jpeg_io"])
def test_pipelined_loop_equals_synchronous_calls(bench_mod, shared, lanes, jpeg_io):
    args, ctx, pipes = shared
    r = bench_mod.Runner(args, ctx, pipes=pipes, lanes=lanes, jpeg_io=jpeg_io, verify=True)
    try:
        v = r.run_verify(STEPS)
    finally:
        r.close()
    assert v["lanes"] == lanes and v["det_batch_frames"] == 5 and v["sr_batch_frames"] == 10 and v["steps"] == STEPS
    assert v["frames"] == STEPS and v["crops"] == STEPS * args.sr_crops, v
    assert v["rows_equal"] and v["boxes_equal"] and v["crops_equal"], v["mismatches"]
    if jpeg_io:
        assert v["files_equal"] is True, v["mismatches"]
    assert v["ok"]


def test_verify_notices_a_wrong_byte(bench_mod, shared):
    """The checker itself: one flipped byte in one recorded crop must fail the comparison."""
    args, ctx, pipes = shared
    r = bench_mod.Runner(args, ctx, pipes=pipes, lanes=1, verify=True)
    try:
        r.setup(2, 10)
        r.rec.clear()
        r.loop(10)
        r.rec[(7, 0)]["crops"][3][5] ^= 1
        v = r.verify_against_synchronous(10)
    finally:
        r.close()
    assert not v["crops_equal"] and not v["ok"] and v["mismatches"] == ["crop 3 of frame (7, 0)"]
