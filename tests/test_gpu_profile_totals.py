"""The figures behind bench.py's roofline line: algorithmic bytes in the per-variant profile table (ffp_sr_profile_bytes) and the process-wide launch totals
(ffp_conv_totals_*) that let a rocprofv3 --pmc pass and the library count the same launches — eager runs, the capture run and hipGraph replays alike."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_profile_bytes_and_lifetime_totals(gpu_lib):
    from ffp_amd import synth
    enh = gpu_lib.Enhancer(synth.rrdbnet_weights(4, 23), 4, 23, half=True)
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (s, s, 3), dtype=np.uint8) for s in (24, 48, 32, 64)]
    px = sum(s * s for s in (24, 48, 32, 64))
    enh.enhance_batch(imgs)                                   # lay the plan out (and tune) before counting
    gpu_lib.conv_totals_enable(True)
    try:
        calls = 0
        for rep in range(4):                                  # eager, capture, replay, replay
            enh.enhance_batch(imgs)
            calls += 1
        enh.set_profile(True)
        enh.enhance_batch(imgs)
        calls += 1
        enh.set_profile(False)
        prof = {p["variant"]: p for p in enh.profile()}
        tot = gpu_lib.conv_totals()
    finally:
        gpu_lib.conv_totals_enable(False)
    r16 = prof["f16_k3s1_rows16"]
    assert r16["launches"] == 349 and r16["flops"] > 0
    # inputs + outputs + residuals once, fp16, + weights: between the bare output bytes and the "every input channel read once per layer" sum
    body_out_bytes = px * 2 * (23 * 3 * (4 * 32 + 64))
    assert body_out_bytes < r16["bytes"] < 12 * body_out_bytes
    flops_per_px = r16["flops"] / px
    assert 35.0e6 < flops_per_px < 38.5e6                    # 35.8 MFLOP per LR pixel for the body + conv_body / up / hr convs at x1, x4, x16 pixels
    for v, p in prof.items():
        assert tot[v]["launches"] == calls * p["launches"], v
        assert tot[v]["flops"] == pytest.approx(calls * p["flops"], rel=1e-9) and tot[v]["bytes"] == pytest.approx(calls * p["bytes"], rel=1e-9), v
    assert gpu_lib.conv_totals() == tot                       # switched off: nothing is added any more
    enh.enhance_batch(imgs)
    assert gpu_lib.conv_totals() == tot


def test_drop_plans_releases_and_rebuilds(gpu_lib):
    from ffp_amd import synth
    enh = gpu_lib.Enhancer(synth.rrdbnet_weights(4, 23), 4, 23, half=True)
    img = np.random.default_rng(1).integers(0, 256, (40, 36, 3), dtype=np.uint8)
    a = enh.enhance(img)
    m0 = enh.mem_bytes()
    assert m0["plans"] > 0 and m0["plans_resident"] >= 1
    enh.drop_plans()
    m1 = enh.mem_bytes()
    assert m1["plans"] == 0 and m1["plans_resident"] == 0 and m1["weights"] == m0["weights"]
    assert np.array_equal(enh.enhance(img), a)
    W = synth.yolo11_pose_weights("n")
    det = gpu_lib.Detector(W, arch="n", nc=int(W["model.23.cv3.0.2.weight"].shape[0]), nkpt=int(W["model.23.cv4.0.2.weight"].shape[0]) // 3, device=0,
                           precision=gpu_lib.PREC_F32X3)
    frame = synth.synthetic_frame(256, 256, seed=2, n_blobs=6)
    r0 = det.infer_tiles(frame, [[0, 0, 256, 256]], 256, 0.25, 0.7, 300)
    assert det.mem_bytes()["plans_resident"] == 1
    det.drop_plans()
    assert det.mem_bytes()["plans"] == 0
    r1 = det.infer_tiles(frame, [[0, 0, 256, 256]], 256, 0.25, 0.7, 300)
    assert all(np.array_equal(x, y) for x, y in zip(r0, r1))
