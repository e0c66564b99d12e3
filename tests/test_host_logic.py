"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/ffp.h declares, the pure host
entry points agree with the oracle, compute entry points fail loudly without a GPU, the weight container round-trips."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import ffp_amd  # noqa: F401
from ffp_amd import _lib, arch, pipeline, synth, weights_io
from oracle import sahi_ref, ultra_post

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ffp_amd import build
    build.build(verbose=False)
    return _lib.lib()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ffp.h")).read()
    declared = set(re.findall(r"\b(ffp_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ffp.h but not exported by libffp.so"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert lib.ffp_version() >= 200


def test_slice_bboxes_matches_oracle(lib):
    for (H, W, s, o) in [(2160, 3840, 512, 0.2), (4320, 7680, 640, 0.25), (300, 400, 512, 0.2), (513, 1025, 512, 0.0), (1000, 777, 333, 0.37)]:
        assert _lib.slice_bboxes(H, W, s, s, o, o).tolist() == sahi_ref.get_slice_bboxes(H, W, s, s, o, o)
    with pytest.raises(_lib.FfpError):
        _lib.slice_bboxes(100, 100, 0, 10)


def test_letterbox_geometry_matches_oracle(lib):
    rng = np.random.default_rng(0)
    for _ in range(200):
        h, w = int(rng.integers(8, 5000)), int(rng.integers(8, 5000))
        s = int(rng.choice([256, 512, 640, 1024]))
        assert _lib.letterbox_geometry(h, w, s) == ultra_post.letterbox_geometry(h, w, s), (h, w, s)


def test_compute_calls_fail_loudly_without_gpu(lib):
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.FfpError) as e:
        _lib.Detector(synth.yolo11_pose_weights("n"), arch="n")
    assert "no HIP device" in str(e.value)
    with pytest.raises(_lib.FfpError):
        _lib.merge(np.zeros((3, 6), np.float32))
    with pytest.raises(_lib.FfpError):
        _lib.Enhancer({"conv_first.weight": np.zeros((64, 3, 3, 3), np.float32)})
    with pytest.raises(_lib.FfpError):
        _lib.op_conv2d(np.zeros((1, 4, 4, 4), np.float32), np.zeros((4, 4, 1, 1), np.float32), None)


def test_bad_arguments_are_reported(lib):
    n = C.c_int32(0)
    assert lib.ffp_slice_bboxes(10, 10, 4, 4, 1.5, 0.2, None, 0, C.byref(n)) == 1
    assert b"overlap" in lib.ffp_last_error()
    assert lib.ffp_letterbox_geometry(0, 10, 512, None) == 1


def test_weights_container_roundtrip():
    W = synth.yolo11_pose_weights("n")
    buf = weights_io.pack(W)
    back = weights_io.unpack(buf)
    assert list(back) == list(W) and all(np.array_equal(back[k], W[k]) for k in W)
    with pytest.raises(ValueError):
        weights_io.unpack(b"nope" + buf[4:])


def test_bn_fold_converter():
    rng = np.random.default_rng(1)
    sd = {"model.0.conv.weight": rng.standard_normal((8, 3, 3, 3)).astype(np.float32), "model.0.bn.weight": rng.uniform(0.5, 2, 8).astype(np.float32),
          "model.0.bn.bias": rng.standard_normal(8).astype(np.float32), "model.0.bn.running_mean": rng.standard_normal(8).astype(np.float32),
          "model.0.bn.running_var": rng.uniform(0.5, 2, 8).astype(np.float32), "model.23.cv2.0.2.weight": rng.standard_normal((64, 8, 1, 1)).astype(np.float32),
          "model.23.cv2.0.2.bias": rng.standard_normal(64).astype(np.float32), "model.23.dfl.conv.weight": np.arange(16, dtype=np.float32).reshape(1, 16, 1, 1)}
    out = weights_io.from_ultralytics_state_dict(sd)
    assert set(out) == {"model.0.conv.weight", "model.0.conv.bias", "model.23.cv2.0.2.weight", "model.23.cv2.0.2.bias"}
    import torch
    import torch.nn.functional as F
    x = torch.randn(1, 3, 9, 9)
    y = F.batch_norm(F.conv2d(x, torch.from_numpy(sd["model.0.conv.weight"]), padding=1), torch.from_numpy(sd["model.0.bn.running_mean"]),
                     torch.from_numpy(sd["model.0.bn.running_var"]), torch.from_numpy(sd["model.0.bn.weight"]), torch.from_numpy(sd["model.0.bn.bias"]), eps=1e-3)
    y2 = F.conv2d(x, torch.from_numpy(out["model.0.conv.weight"]), torch.from_numpy(out["model.0.conv.bias"]), padding=1)
    assert torch.allclose(y, y2, atol=1e-5)


def test_frame_items_and_sharding(lib):
    cfg = pipeline.PipeConfig()
    it = pipeline.frame_items(2160, 3840, cfg, 1)
    assert it.shape == (61, 4) and it[-1].tolist() == [0, 0, 3840, 2160]
    it2 = pipeline.frame_items(2160, 3840, cfg, 2)
    assert it2.shape == (122, 4) and it2[61].tolist() == [0, 2160, 512, 2672] and it2[-1].tolist() == [0, 2160, 3840, 4320]
    for world in (1, 2, 3, 4, 8):
        cover = []
        for r in range(world):
            lo, hi, per = pipeline.shard(61, r, world)
            assert hi - lo <= per
            cover += list(range(lo, hi))
        assert cover == list(range(61))
    sizes = pipeline.sr_crop_sizes(32, 0)
    assert set(sizes.tolist()) <= {24, 32, 48, 64, 96}
    b = pipeline.crop_boxes_for_sr(np.zeros((0, 21), np.float32), 2160, 3840, 32, sizes, seed=3)
    assert ((b[:, 2] - b[:, 0]) == sizes).all() and (b[:, 0] >= 0).all() and (b[:, 2] <= 3840).all() and (b[:, 3] <= 2160).all()


def test_pmc_symbol_to_variant_mapping():
    """tools/pmc_traffic.py keys the PMC table by the variant names bench.py reports (roofline.traffic lookup)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pmc_traffic.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    v = m.variant
    assert v("void ffp::(anonymous namespace)::conv_rows_kernel<1, 2>(ffp::ConvArgs)") == "f16_k3s1_rows"
    assert v("void ffp::conv_mfma_kernel<ffp::X3, 1, 1, 2, 2, 4, 2, 16>(ffp::ConvArgs)") == "f32x3_k1s1_wide"
    assert v("void ffp::conv_mfma_kernel<ffp::X3, 3, 2, 4, 1, 1, 2, 16>(ffp::ConvArgs)") == "f32x3_k3s2_narrow2"
    assert v("void ffp::conv_mfma_kernel<ffp::X3, 3, 1, 4, 1, 1, 1, 16>(ffp::ConvArgs)") == "f32x3_k3s1_narrow1H"
    assert v("_ZN3ffp16conv_mfma_kernelIDF16_Li3ELi1ELi4ELi1ELi2ELi1ELi16EEEvNS_8ConvArgsE") == "f16_k3s1_narrow1"
    assert v("void ffp::conv_mfma_kernel<float, 1, 1, 2, 2, 2, 2, 32>(ffp::ConvArgs)") == "f32_k1s1_wideH"
    assert v("ffp::(anonymous namespace)::conv_rows16_kernel(ffp::ConvArgs)") == "f16_k3s1_rows16"
    assert v("void ffp::(anonymous namespace)::conv_pw_kernel<8, 1, 4, 3, false>(ffp::ConvArgs)") == "f32x3_k1s1_pw1x4w"
    assert v("_ZN3ffp12_GLOBAL__N_114conv_pw_kernelILi4ELi2ELi2ELi2ELb1EEEvNS_8ConvArgsE") == "f32x3_k1s1_pw2x2"
    assert v("void ffp::(anonymous namespace)::conv_pw_kernel<8, 2, 2, 2, true, false>(ffp::ConvArgs)") == "f32x3_k1s1_pw2x2w"
    assert v("void ffp::(anonymous namespace)::conv_pw_kernel<7, 1, 4, 4, false, true>(ffp::ConvArgs)") == "f32x3_k1s1_pw1x4s"
    assert v("_ZN3ffp12_GLOBAL__N_114conv_pw_kernelILi7ELi1ELi4ELi4ELb1ELb1EEEvNS_8ConvArgsE") == "f32x3_k1s1_pw1x4s"
    assert v("__amd_rocclr_copyBuffer") is None


@pytest.mark.parametrize("order", ["torch_first", "libffp_first"])
def test_single_hip_runtime(order):
    """libffp.so, torch and RCCL must share ONE HIP runtime whatever the import order (the wheel bundles a libamdhip64 with
    the same SONAME as the system one; two copies, or the wheel running on the system copy, is the hazard)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n" % ROOT) + (
        "import torch\nimport ffp_amd\nfrom ffp_amd import _lib\n_lib.lib()\n" if order == "torch_first"
        else "import ffp_amd\nfrom ffp_amd import _lib\n_lib.lib()\nimport torch\n") + (
        "import torch.distributed\n"
        "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
        "print(len(libs), libs)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    n, libs = out.stdout.strip().split(" ", 1)
    assert n == "1", libs
    import importlib.util
    if importlib.util.find_spec("torch") is not None:
        assert "torch" in libs            # the wheel's own copy, in both orders


def test_bench_lanes_deal_every_group_exactly_once():
    """bench.py --lanes L: detection groups are dealt round-robin, so the lanes together walk exactly the single-lane sequence of steps
    (same frames, crops and seeds); the lane that owns the last group is the one whose final SR batch is event-timed."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for n_steps in (0, 1, 3, 5, 7, 20, 23, 200):
        for db in (1, 4, 5):
            for L in (1, 2, 3, 4):
                lanes = []
                for k in range(L):
                    r = bench.Runner.__new__(bench.Runner)
                    r.DB, r.n_lanes, r.lane = db, L, k
                    lanes.append(r)
                all_groups = lanes[0].groups(n_steps)
                assert sum(g[1] for g in all_groups) == n_steps and all(0 < g[1] <= db for g in all_groups)
                dealt = sorted(g for r in lanes for g in r.my_groups(n_steps))
                assert [(g0, gsz) for _, g0, gsz in dealt] == all_groups and [gi for gi, _, _ in dealt] == list(range(len(all_groups)))
                steps = sorted(s for r in lanes for _, g0, gsz in r.my_groups(n_steps) for s in range(g0, g0 + gsz))
                assert steps == list(range(n_steps))
                if all_groups:
                    owner = (len(all_groups) - 1) % L
                    assert lanes[owner].my_groups(n_steps)[-1][0] == len(all_groups) - 1
