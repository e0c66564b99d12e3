"""Full-size (BASELINE.json configs 2/3: 3840x2160, YOLO11s, SAHI 512/0.2, 32 crops) checks through size-independent
properties — the oracle needs minutes per 4K frame on a CPU, so here the HIP path is checked against itself and
against cheap invariants: determinism, fused == tile-wise composition (bit exact), sharded == unsharded, merge
idempotence / order invariants, SR batch == single, and an oracle spot check on two of the 61 items."""
import numpy as np
import pytest

from util import match_by_iou, psnr_u8

pytestmark = pytest.mark.gpu
H, W = 2160, 3840


@pytest.fixture(scope="module")
def ctx(gpu_lib):
    from ffp_amd import synth
    Wd = synth.yolo11_pose_weights("s")
    frame = synth.synthetic_frame(H, W, seed=0)
    det = gpu_lib.Detector(Wd, arch="s", precision=gpu_lib.PREC_F32)
    return gpu_lib, Wd, frame, det


def test_4k_item_count_and_determinism(ctx):
    lib, _, frame, det = ctx
    sl = lib.slice_bboxes(H, W, 512, 512, 0.2, 0.2)
    assert len(sl) == 60
    a = det.sliced_predict(frame, 512, 512, 0.2, 0.2, True, 512, 0.25, 0.7, 300, "GREEDYNMM", "IOS", 0.5, False)
    b = det.sliced_predict(frame, 512, 512, 0.2, 0.2, True, 512, 0.25, 0.7, 300, "GREEDYNMM", "IOS", 0.5, False)
    assert a.shape[0] > 0 and np.array_equal(a, b)                           # bitwise repeatable
    assert np.all(a[:, :4] == np.trunc(a[:, :4])) and np.all(a[:, 0] >= 0) and np.all(a[:, 2] <= W) and np.all(a[:, 3] <= H)
    assert np.all(np.diff(a[:, 4]) <= 0)                                     # output order = score descending


def test_4k_fused_equals_tilewise_and_sharded(ctx):
    lib, _, frame, det = ctx
    sl = lib.slice_bboxes(H, W, 512, 512, 0.2, 0.2)
    tiles = [tuple(t) for t in sl] + [(0, 0, W, H)]
    per = det.infer_tiles(frame, tiles, 512, 0.25, 0.7, 300)
    # the same items in two shards (what two ranks would compute) must give the same per-item detections
    half = len(tiles) // 2
    per_a = det.infer_tiles(frame, tiles[:half], 512, 0.25, 0.7, 300)
    per_b = det.infer_tiles(frame, tiles[half:], 512, 0.25, 0.7, 300)
    for x, y in zip(per, per_a + per_b):
        assert np.array_equal(x, y)          # exact-fp32 mode: every item's result is independent of what else is in the batch, bit for bit
    # The default arithmetic (FFP_PREC_F32X3, scaled fp16 hi/lo split) takes its power-of-two activation scale per (tensor, IMAGE) since
    # round 3 (a max-|value| slot per image, Plan::per_image_amax): an item's bits no longer depend on its batch mates, so sharded ==
    # unsharded bit for bit in this arithmetic too — whatever the split: halves, thirds, one item at a time
    lib3 = lib.Detector(ctx[1], arch="s", precision=lib.PREC_F32X3)
    q = lib3.infer_tiles(frame, tiles, 512, 0.25, 0.7, 300)
    qa = lib3.infer_tiles(frame, tiles[:half], 512, 0.25, 0.7, 300) + lib3.infer_tiles(frame, tiles[half:], 512, 0.25, 0.7, 300)
    third = len(tiles) // 3
    qb = lib3.infer_tiles(frame, tiles[:third], 512, 0.25, 0.7, 300) + lib3.infer_tiles(frame, tiles[third:2 * third + 5], 512, 0.25, 0.7, 300) + \
        lib3.infer_tiles(frame, tiles[2 * third + 5:], 512, 0.25, 0.7, 300)
    for k, (x, y, y3, z) in enumerate(zip(q, qa, qb, per)):
        assert np.array_equal(x, y) and np.array_equal(x, y3), k
        if x.shape[0]:
            assert x.shape == z.shape
            jz = [m[1] for m in match_by_iou(x[:, :4], z[:, :4], x[:, 4], z[:, 4])]
            np.testing.assert_allclose(x[:, :4], z[jz, :4], atol=5e-3, rtol=1e-5)      # and the split agrees with exact fp32
    for k in (0, 31, 60):                                                              # single items, incl. the full-frame pass (padded deepest level)
        one = lib3.infer_tiles(frame, [tiles[k]], 512, 0.25, 0.7, 300)[0]
        assert np.array_equal(one, q[k]), k
    rows = []
    for t, d in zip(tiles, per):
        d = d.copy()
        b = np.trunc(d[:, :4])
        b[:, 2] = np.minimum(b[:, 2], W); b[:, 3] = np.minimum(b[:, 3], H)
        d[:, 0] = b[:, 0] + t[0]; d[:, 1] = b[:, 1] + t[1]; d[:, 2] = b[:, 2] + t[0]; d[:, 3] = b[:, 3] + t[1]
        d[:, 6::3] += t[0]; d[:, 7::3] += t[1]
        rows.append(d)
    rows = np.concatenate(rows, 0)
    for pt, m, ag in (("GREEDYNMM", "IOS", False), ("NMS", "IOS", True)):
        merged, src = lib.merge(rows, pt, m, 0.5, ag)
        fused = det.sliced_predict(frame, 512, 512, 0.2, 0.2, True, 512, 0.25, 0.7, 300, pt, m, 0.5, ag)
        assert np.array_equal(merged, fused)
        if pt == "NMS":
            # idempotence: the survivors of NMS do not suppress each other; and every survivor is an input row
            again, _ = lib.merge(merged, pt, m, 0.5, ag)
            assert np.array_equal(again, merged)
            assert np.array_equal(merged, rows[src])
        else:
            # a merged box contains its source box, scores never decrease
            assert np.all(merged[:, 0] <= rows[src, 0]) and np.all(merged[:, 2] >= rows[src, 2])
            assert np.all(merged[:, 4] >= rows[src, 4])


def test_4k_oracle_spot_check(ctx):
    """Two of the 61 items (one native slice, the resized full frame) against the CPU oracle."""
    lib, Wd, frame, det = ctx
    from oracle import ultra_post
    from oracle.yolo11_ref import Yolo11PoseRef
    ref = Yolo11PoseRef(Wd, "s")
    tiles = [(1640, 820, 2152, 1332), (0, 0, W, H)]
    res = det.infer_tiles(frame, tiles, 512, 0.25, 0.7, 300)
    for t, d in zip(tiles, res):
        r = ultra_post.predict(ref, frame[t[1]:t[3], t[0]:t[2]], 512, 0.25, 0.7, 300)
        assert d.shape[0] == len(r)
        if len(r):
            ious = np.array([x[2] for x in match_by_iou(r.xyxy, d[:, :4])])
            assert ious.min() >= 0.999


def test_sr_config3_batch_properties(ctx):
    lib, _, frame, _ = ctx
    from ffp_amd import pipeline, synth
    sizes = pipeline.sr_crop_sizes(32, 0)
    boxes = pipeline.crop_boxes_for_sr(np.zeros((0, 21), np.float32), H, W, 32, sizes, seed=0)
    crops = [frame[b[1]:b[3], b[0]:b[2]].copy() for b in boxes]
    e = lib.Enhancer(synth.rrdbnet_weights(4, 23), 4, 23, half=True)
    outs = e.enhance_batch(crops)
    assert all(o.shape == (c.shape[0] * 4, c.shape[1] * 4, 3) for o, c in zip(outs, crops))
    again = e.enhance_batch(crops)
    assert all(np.array_equal(a, b) for a, b in zip(outs, again))              # deterministic
    for i in (0, 7, 31):                                                       # ragged batch == one at a time
        assert np.array_equal(outs[i], e.enhance(crops[i]))
    # permuting the batch permutes the outputs
    perm = np.random.default_rng(0).permutation(32)
    outs_p = e.enhance_batch([crops[i] for i in perm])
    assert all(np.array_equal(outs_p[k], outs[perm[k]]) for k in range(32))
    # fp32 mode agrees with fp16 mode to the stated PSNR floor
    e32 = lib.Enhancer(synth.rrdbnet_weights(4, 23), 4, 23, half=False)
    assert psnr_u8(e32.enhance(crops[3]), outs[3]) >= 50.0
