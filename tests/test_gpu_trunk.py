"""The fused body launch (csrc/conv_trunk.hip: every conv of RRDBNet's residual dense blocks — /root/reference/utils/enhancer.py:121-128 — as
ONE persistent launch with per-tile dependency counters) against its two oracles: the per-layer kernel it replaces (BIT-IDENTICAL: same MFMA,
same accumulation order) and, through that, the CPU oracle every SR test holds the per-layer path to. Ragged batches, partial tiles at every
border, batches far smaller than the grid (every dependency is late) and far larger (none is), repeated calls (the counters are re-zeroed)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rng_conv(cin, cout, seed):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((cout, cin, 3, 3)) * (2.0 / (cin * 9)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    return w, b


@pytest.mark.parametrize("cin,cout", [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64)])
@pytest.mark.parametrize("n,h,w", [(3, 32, 16), (2, 50, 41), (5, 17, 33), (1, 96, 96), (7, 8, 8)])
def test_single_layer_matches_rows16_bit_for_bit(gpu_lib, cin, cout, n, h, w):
    """force_shape 25 = conv_trunk_kernel on one layer; 9 = conv_rows16_kernel. Same bits, including the zero padding at every image border
    (out-of-range DMA lanes), partial tiles (h, w not multiples of 32 / 16) and the residual + LeakyReLU epilogue."""
    rng = np.random.default_rng(cin * 1000 + cout + h)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt, b = _rng_conv(cin, cout, cin + cout)
    res = rng.standard_normal((n, h, w, cout)).astype(np.float32)
    outs = []
    for shape in (9, 25):
        gpu_lib.op_conv2d_shape(shape)
        try:
            outs.append((gpu_lib.op_conv2d(x, wt, b, act=2, precision=gpu_lib.PREC_F16),
                         gpu_lib.op_conv2d(x, wt, b, act=0, res=res, res_scale=0.2, precision=gpu_lib.PREC_F16)))
        finally:
            gpu_lib.op_conv2d_shape(-1)
    assert np.isfinite(outs[1][0]).all() and np.abs(outs[1][0]).max() > 0
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])


def _crops(sizes, seed=0):
    from ffp_amd import synth
    out = []
    for k, (h, w) in enumerate(sizes):
        f = synth.synthetic_frame(max(h, 64), max(w, 64), seed=seed + k, n_blobs=5)
        out.append(f[:h, :w][..., ::-1].copy())
    return out


@pytest.fixture(scope="module")
def enh(gpu_lib):
    from ffp_amd import synth
    return gpu_lib.Enhancer(synth.rrdbnet_weights(4, 23), 4, 23, half=True)


@pytest.mark.parametrize("sizes", [
    [(24, 24)],                                             # one tile: every layer waits for the layer before it
    [(32, 32), (24, 37), (50, 41)],                         # a few tiles, ragged: far fewer items per layer than workgroups
    [(48, 48)] * 3 + [(96, 96), (64, 64), (33, 17), (16, 16), (4, 4), (5, 70)],
], ids=["one_tile", "ragged_small", "ragged_mixed"])
def test_fused_body_equals_per_layer_launches(enh, sizes):
    imgs = _crops(sizes, seed=11)
    enh.set_fused_body(False)
    ref = enh.enhance_batch(imgs)
    enh.set_fused_body(True)
    for rep in range(3):                                    # eager run, graph capture, graph replay: the queue and the counters start from zero each time
        out = enh.enhance_batch(imgs)
        for a, b in zip(out, ref):
            assert a.shape == b.shape and np.array_equal(a, b), (rep, a.shape)


def test_fused_body_large_batch_and_changing_batches(enh, gpu_lib):
    """A 10-frame-sized batch (about 2,000 16 x 16 tiles: several items per workgroup and layer, dependencies met ahead of time) and then
    batches of other sizes through the same capacity plan."""
    from ffp_amd import pipeline
    rng = np.random.default_rng(3)
    for n_crops, seed in ((320, 1), (64, 2), (200, 3)):
        sizes = pipeline.sr_crop_sizes(n_crops, seed=seed)
        imgs = [rng.integers(0, 256, (int(s), int(s), 3), dtype=np.uint8) for s in sizes]
        enh.set_fused_body(True)
        out = enh.enhance_batch(imgs)
        enh.set_fused_body(False)
        ref = enh.enhance_batch(imgs)
        bad = [k for k, (a, b) in enumerate(zip(out, ref)) if not np.array_equal(a, b)]
        assert not bad, (n_crops, bad[:8])
    enh.set_fused_body(True)


def test_fused_body_matches_the_cpu_oracle(enh):
    """The usual bar on the shipped path itself: fp16 >= 50 dB against the torch-CPU restatement (tests/test_gpu_sr.py holds the per-layer path to it)."""
    from ffp_amd import synth
    from oracle import rrdbnet_ref
    from util import psnr_u8
    ref_net = rrdbnet_ref.RRDBNetRef(synth.rrdbnet_weights(4, 23), 4, 23)
    enh.set_fused_body(True)
    for img in _crops([(40, 28), (33, 49)], seed=5):
        out = enh.enhance(img)
        assert psnr_u8(out, rrdbnet_ref.enhance(ref_net, img)) >= 50.0


def test_memory_report(enh, gpu_lib):
    m = enh.mem_bytes()
    assert m["weights"] > 30e6 and m["plans"] > 0 and 1 <= m["plans_resident"] <= 4
