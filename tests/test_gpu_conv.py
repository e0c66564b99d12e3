"""GPU parity of the MFMA convolution kernels (through the C-ABI ffp_op_conv2d) against a plain PyTorch fp32
reference of the same op. fp32 mode: exact-f32 MFMA, tolerance = summation-order noise. fp16 mode: operands rounded
to fp16 on both sides, fp32 accumulate, output rounded to fp16 (tolerance stated per assert)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {0: lambda t: t, 1: F.silu, 2: lambda t: F.leaky_relu(t, 0.2)}


def ref_conv(x, w, b, stride, groups, act, up, res, res_scale, half):
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    wt = torch.from_numpy(w)
    if half:
        xt = xt.half().float()
        wt = wt.half().float()
    if up:
        xt = F.interpolate(xt, scale_factor=2, mode="nearest")
    y = F.conv2d(xt, wt, torch.from_numpy(b), stride=stride, padding=w.shape[2] // 2, groups=groups)
    y = ACTS[act](y)
    if res is not None:
        r = torch.from_numpy(res).permute(0, 3, 1, 2)
        if half:
            r = r.half().float()
        y = y * res_scale + r
    return y.permute(0, 2, 3, 1).contiguous().numpy()


CASES = [
    # (n, h, w, cin, cout, k, stride, act, up, res)
    (2, 16, 16, 64, 128, 1, 1, 1, 0, False),
    (1, 20, 24, 96, 32, 1, 1, 0, 0, True),
    (3, 9, 16, 512, 512, 1, 1, 1, 0, False),
    (1, 16, 16, 48, 64, 1, 1, 1, 0, False),
    (2, 32, 32, 32, 64, 3, 1, 1, 0, True),
    (1, 18, 37, 64, 32, 3, 1, 2, 0, False),
    (1, 13, 11, 192, 64, 3, 1, 0, 0, True),
    (1, 24, 24, 160, 32, 3, 1, 2, 0, False),
    (2, 32, 32, 4, 32, 3, 2, 1, 0, False),
    (1, 64, 48, 64, 128, 3, 2, 1, 0, False),
    (1, 16, 16, 256, 256, 3, 2, 1, 0, False),
    (1, 12, 10, 64, 64, 3, 1, 2, 1, False),
    (1, 16, 16, 64, 3, 3, 1, 0, 0, False),
    (1, 16, 16, 128, 1, 1, 1, 0, 0, False),
    (1, 16, 16, 32, 15, 1, 1, 0, 0, False),
    (2, 8, 8, 16, 16, 3, 1, 1, 0, True),
    (1, 16, 16, 8, 16, 3, 1, 1, 0, False),
    (1, 8, 8, 1024, 512, 1, 1, 1, 0, False),
    (1, 16, 16, 768, 256, 1, 1, 1, 0, False),
    (1, 33, 47, 3, 32, 3, 2, 1, 0, False),      # image-input convs: direct VALU kernel
    (2, 16, 16, 3, 64, 3, 1, 2, 0, False),
    (1, 20, 20, 3, 16, 3, 2, 1, 0, False),
    (3, 41, 42, 128, 32, 3, 1, 2, 0, False),    # ESRGAN dense-block shapes: fp16 takes the row-reuse kernel (conv_rows.hip)
    (2, 52, 33, 96, 32, 3, 1, 2, 0, True),
    (1, 96, 96, 192, 64, 3, 1, 0, 0, True),
    (2, 17, 16, 64, 64, 3, 1, 2, 1, False),
    (1, 1, 1, 64, 32, 3, 1, 2, 0, False),
    (1, 20, 20, 64, 128, 3, 1, 1, 0, False),    # row-reuse kernel with 4 / 3 channel blocks per pixel tile
    (1, 16, 24, 128, 96, 3, 1, 0, 0, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_k%ds%d_a%d_u%d_r%d" % c)
def test_conv_dense_f32x3(gpu_lib, case):
    """fp32 storage, fp16 hi/lo split products (3 MFMAs): fp32-grade. Tolerance 1e-5 relative to the output scale
    (vs 2e-5 for the exact-fp32 kernel, whose only noise is summation order)."""
    n, h, w, cin, cout, k, stride, act, up, has_res = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, k, k), dtype=np.float32) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32) * 0.1
    hi, wi = (h * 2, w * 2) if up else (h, w)
    ho, wo = (hi + 2 * (k // 2) - k) // stride + 1, (wi + 2 * (k // 2) - k) // stride + 1
    res = rng.standard_normal((n, ho, wo, cout), dtype=np.float32) if has_res else None
    y = gpu_lib.op_conv2d(x, wt, b, stride=stride, act=act, up=bool(up), res=res, res_scale=0.2 if has_res else 1.0, precision=gpu_lib.PREC_F32X3)
    ref = ref_conv(x, wt, b, stride, 1, act, up, res, 0.2, False)
    np.testing.assert_allclose(y, ref, rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_k%ds%d_a%d_u%d_r%d" % c)
def test_conv_dense(gpu_lib, case, half):
    n, h, w, cin, cout, k, stride, act, up, has_res = case
    if half and cin % 8 and cin != 3:
        pytest.skip("fp16 activations are addressed in 8-channel vectors")
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, k, k), dtype=np.float32) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32) * 0.1
    hi, wi = (h * 2, w * 2) if up else (h, w)
    ho, wo = (hi + 2 * (k // 2) - k) // stride + 1, (wi + 2 * (k // 2) - k) // stride + 1
    res = rng.standard_normal((n, ho, wo, cout), dtype=np.float32) if has_res else None
    y = gpu_lib.op_conv2d(x, wt, b, stride=stride, act=act, up=bool(up), res=res, res_scale=0.2 if has_res else 1.0,
                          precision=gpu_lib.PREC_F16 if half else gpu_lib.PREC_F32)
    ref = ref_conv(x, wt, b, stride, 1, act, up, res, 0.2, half)
    assert y.shape == ref.shape
    if half:
        # operands identical (fp16-rounded), fp32 accumulate; the only difference is the final rounding to fp16
        np.testing.assert_allclose(y, ref, rtol=2e-3, atol=2e-3)
    else:
        np.testing.assert_allclose(y, ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("half", [False, True], ids=["f32", "f16"])
@pytest.mark.parametrize("c,h,w,act,has_res", [(128, 16, 16, 1, False), (256, 9, 16, 0, True), (64, 33, 17, 1, False)])
def test_conv_depthwise(gpu_lib, c, h, w, act, has_res, half):
    rng = np.random.default_rng(c + h)
    x = rng.standard_normal((2, h, w, c), dtype=np.float32)
    wt = rng.standard_normal((c, 1, 3, 3), dtype=np.float32) / 3
    b = rng.standard_normal(c, dtype=np.float32) * 0.1
    res = rng.standard_normal((2, h, w, c), dtype=np.float32) if has_res else None
    y = gpu_lib.op_conv2d(x, wt, b, groups=c, act=act, res=res, precision=gpu_lib.PREC_F16 if half else gpu_lib.PREC_F32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)
    if half:
        xt = xt.half().float()
    r = ACTS[act](F.conv2d(xt, torch.from_numpy(wt), torch.from_numpy(b), padding=1, groups=c))
    if has_res:
        rr = torch.from_numpy(res).permute(0, 3, 1, 2)
        r = r + (rr.half().float() if half else rr)
    ref = r.permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y, ref, rtol=2e-3 if half else 2e-5, atol=2e-3 if half else 2e-5)


def test_conv_rows_kernel_random_shapes(gpu_lib):
    """Seeded sweep of the row-reuse fp16 3x3 kernel (conv_rows.hip) over ragged sizes, channel blocks, activations, the
    folded x2 upsample and the residual epilogue; every case is also run through the generic kernel (force via cin % 32 != 0
    is impossible here, so the comparison is against the PyTorch reference only)."""
    rng = np.random.default_rng(20260101)
    for it in range(24):
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        cin = int(rng.choice([64, 96, 128, 160, 192]))
        cout = int(rng.choice([32, 64, 96, 128]))
        act = int(rng.choice([0, 1, 2]))
        up = int(rng.integers(0, 2)) if h * w <= 900 else 0
        has_res = bool(rng.integers(0, 2))
        x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
        wt = (rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) / np.sqrt(cin * 9)).astype(np.float32)
        b = rng.standard_normal(cout, dtype=np.float32) * 0.1
        ho, wo = (h * 2, w * 2) if up else (h, w)
        res = rng.standard_normal((n, ho, wo, cout), dtype=np.float32) if has_res else None
        y = gpu_lib.op_conv2d(x, wt, b, stride=1, act=act, up=bool(up), res=res, res_scale=0.2 if has_res else 1.0, precision=gpu_lib.PREC_F16)
        ref = ref_conv(x, wt, b, 1, 1, act, up, res, 0.2, True)
        assert y.shape == ref.shape, (it, y.shape, ref.shape)
        np.testing.assert_allclose(y, ref, rtol=2e-3, atol=2e-3, err_msg=f"case {it}: n{n} {h}x{w} {cin}->{cout} act{act} up{up} res{has_res}")


RANGE_CASES = {
    # name: (weight out-channel scales 10^U(lo,hi), activation multiplier, per-input-channel activation scales 10^U(lo,hi))
    "bn_folded_weights_1e-4_to_1e3": ((-4, 3), 1.0, None),
    "saturating_activations_x3e5": (None, 3e5, None),             # |x| up to ~1.5e6: an unscaled fp16 split gives inf / NaN
    "tiny_activations_x1e-6": (None, 1e-6, None),                 # an unscaled split keeps ~fp16 precision here
    "activation_channels_1e-4_to_1e3": (None, 1.0, (-4, 3)),
    "everything_at_once": ((-4, 3), 37.0, (-3, 2)),
}


@pytest.mark.parametrize("name", list(RANGE_CASES))
@pytest.mark.parametrize("shape", [(2, 24, 20, 64, 64, 3, 1), (1, 16, 16, 128, 96, 1, 1), (1, 33, 31, 32, 64, 3, 2)], ids=["k3s1", "k1", "k3s2"])
def test_f32x3_scaled_split_is_fp32_grade_over_ranges(gpu_lib, name, shape):
    """FFP_PREC_F32X3 outside the unit-variance comfort zone: per-channel weight scales as BatchNorm folding of a trained
    checkpoint yields them, activations beyond fp16's range and far below it. The split is SCALED — activations by a tensor-wide
    power of two taken from the producer's max |value|, weights by a per-output-channel power of two — so the error, relative to
    each output channel's own magnitude, has to stay at the exact-fp32 kernel's level. Reference: float64 convolution."""
    wsc, amul, asc = RANGE_CASES[name]
    n, h, w, cin, cout, k, stride = shape
    rng = np.random.default_rng(abs(hash((name, shape))) % (2 ** 31))
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32) * np.float32(amul)
    if asc is not None:
        x *= (10.0 ** rng.uniform(asc[0], asc[1], cin)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    if wsc is not None:
        s = (10.0 ** rng.uniform(wsc[0], wsc[1], cout)).astype(np.float32)
        wt *= s[:, None, None, None]
        b *= s
    ref = F.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), stride=stride,
                   padding=k // 2).permute(0, 2, 3, 1).numpy()
    scale = np.abs(ref).reshape(-1, cout).max(0) + 1e-30                 # per output channel
    errs = {}
    for mode, prec in (("f32x3", gpu_lib.PREC_F32X3), ("f32", gpu_lib.PREC_F32)):
        y = gpu_lib.op_conv2d(x, wt, b, stride=stride, act=0, precision=prec)
        assert np.isfinite(y).all(), (name, mode)
        errs[mode] = float((np.abs(y - ref).reshape(-1, cout).max(0) / scale).max())
    # exact-fp32 MFMA is the yardstick: its only noise is the fp32 summation order (~1e-6 at these K)
    assert errs["f32x3"] <= max(3.0 * errs["f32"], 2e-6), (name, errs)
    assert errs["f32x3"] <= 1e-5, (name, errs)


# force_shape -> (32-channel tiles per workgroup, LDS-resident weight fragments; a negative number -m: weights streamed by a producer wave, cin a multiple of m)
PW_SHAPES = {10: (4, 40), 11: (2, 40), 12: (1, 40), 13: (4, 80), 14: (2, 80), 15: (1, 80), 16: (4, -128), 22: (2, -64)}


def pw_can_run(shape, cin, cout):
    nt, frags = PW_SHAPES[shape]
    if (-(-cout // 32)) % nt:
        return False
    return cin % (-frags) == 0 if frags < 0 else (cin % 32 == 0 and nt * (cin // 16) <= frags)


PW_CASES = [
    # (n, h, w, cin, cout, act)
    (2, 16, 16, 64, 128, 1),
    (1, 33, 47, 96, 128, 1),        # 1551 pixels: ragged last pixel block
    (1, 16, 16, 128, 1, 0),         # one out channel (cls head)
    (1, 20, 24, 32, 15, 0),         # cout not a multiple of 4 (keypoint head): dword stores, channel tail by the range check
    (1, 64, 64, 192, 256, 1),
    (2, 40, 40, 64, 64, 1),
    (3, 9, 16, 512, 512, 1),
    (1, 16, 16, 256, 256, 0),
    (5, 128, 128, 96, 128, 1),      # 81920 pixels: every workgroup walks several pixel blocks with resident weights
    (9, 96, 96, 64, 64, 1),
    (2, 32, 32, 1024, 512, 1),      # K = 1024: one 32-channel tile per 8-wave workgroup is all that stays resident; streamed weights take 128
    (9, 64, 64, 384, 128, 1),
    (11, 64, 64, 768, 256, 1),      # streamed weights, several pixel blocks per workgroup (stage parity carries over blocks)
    (1, 15, 15, 128, 128, 0),       # one stage per block
]


@pytest.mark.parametrize("case", PW_CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_a%d" % c)
def test_conv_pointwise_kernels(gpu_lib, case):
    """conv_pw.hip (1x1, fp32-grade split, activations loaded straight into MFMA operand registers): every workgroup shape
    that can run the case, against the fp32 reference at the generic split kernel's tolerance, and against the generic kernel
    itself (same products, same accumulation order inside a k-group: 1e-6 of the output scale)."""
    n, h, w, cin, cout, act = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, 1, 1), dtype=np.float32) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32) * 0.1
    ref = ref_conv(x, wt, b, 1, 1, act, 0, None, 1.0, False)
    gen = gpu_lib.op_conv2d(x, wt, b, act=act, precision=gpu_lib.PREC_F32X3)
    ran = 0
    try:
        for shape in sorted(PW_SHAPES):
            if not pw_can_run(shape, cin, cout):
                continue
            gpu_lib.op_conv2d_shape(shape)
            y = gpu_lib.op_conv2d(x, wt, b, act=act, precision=gpu_lib.PREC_F32X3)
            np.testing.assert_allclose(y, ref, rtol=3e-5, atol=3e-5, err_msg=f"shape {shape}")
            np.testing.assert_allclose(y, gen, rtol=0, atol=2e-6 * max(1.0, float(np.abs(ref).max())), err_msg=f"shape {shape} vs generic")
            ran += 1
    finally:
        gpu_lib.op_conv2d_shape(-1)
    assert ran >= 1


def test_conv_pointwise_input_and_output_views_ranges(gpu_lib):
    """Scaled split through the pointwise kernels: per-channel weight scales over 1e-4..1e3 and activations near the fp16
    limits stay fp32-grade (same bar as the generic kernel's range test)."""
    rng = np.random.default_rng(5)
    cin, cout = 96, 128
    x = (rng.standard_normal((2, 24, 24, cin)) * 3000.0).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32) * (10.0 ** rng.uniform(-4, 3, (cout, 1, 1, 1))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = ref_conv(x.astype(np.float64).astype(np.float32), wt, b, 1, 1, 0, 0, None, 1.0, False)
    ref64 = np.einsum("nhwc,oc->nhwo", x.astype(np.float64), wt[:, :, 0, 0].astype(np.float64)) + b
    scale = np.abs(ref64).max(axis=(0, 1, 2), keepdims=True)
    try:
        for shape in (10, 11, 13, 14):
            gpu_lib.op_conv2d_shape(shape)
            y = gpu_lib.op_conv2d(x, wt, b, act=0, precision=gpu_lib.PREC_F32X3)
            assert np.abs((y - ref64) / scale).max() <= 2.0 * max(np.abs((ref - ref64) / scale).max(), 1e-6)
    finally:
        gpu_lib.op_conv2d_shape(-1)


@pytest.mark.parametrize("case", [(2, 16, 16, 256, 256, 128, 1), (1, 24, 40, 512, 256, 256, 1), (3, 34, 18, 64, 32, 64, 0), (2, 64, 64, 128, 64, 128, 1)],
                         ids=lambda c: "n%d_%dx%d_up%d+%d-%d_a%d" % c)
def test_conv1x1_over_virtual_upsample_concat(gpu_lib, case):
    """1x1 conv over [nearest_x2(coarse) | fine] (YOLO neck, layers 11-13 / 14-16 of yolo11-pose.yaml) without the upsampled
    tensor: generic kernel and every pointwise shape that can run the case, against torch (interpolate + cat + conv2d)."""
    n, h, w, c_up, c_fine, cout, act = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    coarse = rng.standard_normal((n, h // 2, w // 2, c_up), dtype=np.float32) * 3.0
    fine = rng.standard_normal((n, h, w, c_fine), dtype=np.float32)
    cin = c_up + c_fine
    wt = (rng.standard_normal((cout, cin), dtype=np.float32) / np.sqrt(cin)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32) * 0.1
    up = F.interpolate(torch.from_numpy(coarse).permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    xt = torch.cat([up, torch.from_numpy(fine).permute(0, 3, 1, 2)], 1)
    ref = ACTS[act](F.conv2d(xt, torch.from_numpy(wt)[:, :, None, None], torch.from_numpy(b))).permute(0, 2, 3, 1).numpy()
    tol = 3e-5 * max(1.0, float(np.abs(ref).max()))
    ran = []
    try:
        for shape in [-1] + sorted(PW_SHAPES):
            if shape >= 0 and not pw_can_run(shape, cin, cout):
                continue
            gpu_lib.op_conv2d_shape(shape)
            y = gpu_lib.op_conv1x1_up2(coarse, fine, wt, b, act=act, precision=gpu_lib.PREC_F32X3)
            np.testing.assert_allclose(y, ref, rtol=3e-5, atol=tol, err_msg=f"shape {shape}")
            ran.append(shape)
    finally:
        gpu_lib.op_conv2d_shape(-1)
    assert len(ran) >= 2, ran


# conv_k3d.hip: force_shape -> minimum number of 32-channel output tiles
K3D_SHAPES = {17: 3, 18: 2, 19: 1, 20: 3, 21: 2}
K3D_CASES = [
    # (n, h, w, cin, cout, stride, act, res)
    (2, 32, 32, 32, 64, 1, 1, True),
    (1, 18, 37, 64, 32, 1, 1, False),      # ragged tile edges in both directions
    (1, 13, 11, 192, 64, 1, 0, True),
    (2, 8, 8, 16, 16, 1, 1, True),         # one 16-channel chunk, cout below one tile
    (1, 1, 1, 64, 32, 1, 1, False),        # a one-pixel image: everything but the centre tap reads padding
    (1, 64, 48, 64, 128, 2, 1, False),
    (1, 16, 16, 256, 256, 2, 1, False),
    (3, 33, 31, 32, 64, 2, 1, False),      # odd sizes at stride 2
    (1, 20, 20, 64, 128, 1, 1, False),
    (1, 16, 24, 128, 96, 1, 0, True),      # 3 tiles: the 128-channel workgroup has an idle tile slot
    (5, 64, 64, 128, 128, 2, 1, False),    # model.3-like: many tiles per persistent workgroup, 8 chunks per item
    (7, 40, 24, 16, 32, 1, 1, True),       # model.2.m.0.cv2-like: one chunk per item (the stream changes item at every chunk)
    (4, 16, 16, 128, 128, 1, 1, True),     # C3k bottleneck at 16 x 16
    (2, 17, 16, 512, 64, 1, 1, False),     # 32 chunks
    (61, 16, 16, 48, 160, 2, 1, False),    # more images than workgroup slots want; 5 tiles (two channel blocks of 128)
]


@pytest.mark.parametrize("case", K3D_CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_s%d_a%d_r%d" % c)
def test_conv_k3_direct_weight_kernels(gpu_lib, case):
    """conv_k3d.hip (3x3 split-arithmetic convs, weights straight from L2 into MFMA operand registers, persistent tile walk): every
    workgroup shape that can run the case, against the fp32 reference at the generic split kernel's tolerance AND bit for bit
    against the generic kernel (same instruction, same fragments, same k order)."""
    n, h, w, cin, cout, stride, act, has_res = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
    wt = (rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout, dtype=np.float32) * 0.1
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = rng.standard_normal((n, ho, wo, cout), dtype=np.float32) if has_res else None
    kw = dict(stride=stride, act=act, res=res, res_scale=0.2 if has_res else 1.0, precision=gpu_lib.PREC_F32X3)
    ref = ref_conv(x, wt, b, stride, 1, act, 0, res, 0.2, False)
    gen = gpu_lib.op_conv2d(x, wt, b, **kw)
    ran = 0
    try:
        for shape, min_tiles in sorted(K3D_SHAPES.items()):
            if -(-cout // 32) < min_tiles:
                continue
            gpu_lib.op_conv2d_shape(shape)
            y = gpu_lib.op_conv2d(x, wt, b, **kw)
            np.testing.assert_allclose(y, ref, rtol=3e-5, atol=3e-5, err_msg=f"shape {shape}")
            assert np.array_equal(y, gen), f"shape {shape}: differs from the generic kernel in {int((y != gen).sum())} of {y.size} values, max {np.abs(y - gen).max()}"
            ran += 1
    finally:
        gpu_lib.op_conv2d_shape(-1)
    assert ran >= 1


def test_conv_rows16_weight_resident_form_is_bit_identical(gpu_lib):
    """conv_rows16_kernel<., RES = true> (force_shape 23: one workgroup per CU, the channel block's weights loaded into LDS once per
    workgroup instead of once per tile) and conv_rows16pc_kernel (force_shape 24: producer waves stage and store, consumer waves
    multiply) against the streaming form (force_shape 9): same MFMAs in the same order -> identical bits; and against the fp16 reference. Ragged sizes, many tiles per workgroup, residual / upsample epilogues, 1..6 chunks."""
    rng = np.random.default_rng(20261004)
    cases = [(2, 41, 42, 64, 32, 2, 0, False), (1, 70, 33, 96, 32, 2, 0, True), (3, 16, 16, 128, 32, 2, 0, False), (2, 52, 33, 160, 32, 2, 0, True),
             (1, 96, 96, 192, 64, 0, 0, True), (2, 17, 16, 64, 64, 2, 1, False), (1, 1, 1, 64, 32, 2, 0, False), (40, 24, 24, 128, 32, 2, 0, False),
             (1, 20, 20, 64, 128, 1, 0, False)]
    try:
        for n, h, w, cin, cout, act, up, has_res in cases:
            x = rng.standard_normal((n, h, w, cin), dtype=np.float32)
            wt = (rng.standard_normal((cout, cin, 3, 3), dtype=np.float32) / np.sqrt(cin * 9)).astype(np.float32)
            b = rng.standard_normal(cout, dtype=np.float32) * 0.1
            ho, wo = (h * 2, w * 2) if up else (h, w)
            res = rng.standard_normal((n, ho, wo, cout), dtype=np.float32) if has_res else None
            kw = dict(stride=1, act=act, up=bool(up), res=res, res_scale=0.2 if has_res else 1.0, precision=gpu_lib.PREC_F16)
            gpu_lib.op_conv2d_shape(9)
            y9 = gpu_lib.op_conv2d(x, wt, b, **kw)
            gpu_lib.op_conv2d_shape(23)
            y23 = gpu_lib.op_conv2d(x, wt, b, **kw)
            assert np.array_equal(y9, y23), (n, h, w, cin, cout, int((y9 != y23).sum()))
            gpu_lib.op_conv2d_shape(24)                       # producer / consumer waves (conv_rows16pc.hip)
            y24 = gpu_lib.op_conv2d(x, wt, b, **kw)
            assert np.array_equal(y9, y24), ("rows16pc", n, h, w, cin, cout, int((y9 != y24).sum()))
            np.testing.assert_allclose(y23, ref_conv(x, wt, b, 1, 1, act, up, res, 0.2, True), rtol=2e-3, atol=2e-3)
    finally:
        gpu_lib.op_conv2d_shape(-1)
