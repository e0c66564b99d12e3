"""oracle/widerface_eval_ref.py against outputs of the reference's own evaluation arithmetic (tests/golden/eval_expected.npz, written by
tests/golden/make_eval_fixtures.py from /root/reference/eval/eval_dual.py and eval_official_widerface.py): bit-exact."""
import os

import numpy as np
import pytest

from oracle import widerface_eval_ref as R

FX = os.path.join(os.path.dirname(__file__), "golden", "eval_expected.npz")


@pytest.fixture(scope="module")
def fx():
    return np.load(FX)


def dual_images(fx, ds, valid_cats):
    faces, fo, pred, po, assign = (fx[f"dual{ds}_{k}"] for k in ("faces", "face_off", "pred", "pred_off", "assign"))
    out = []
    for i in range(len(fo) - 1):
        a = assign[fo[i]:fo[i + 1]]
        out.append({"faces": faces[fo[i]:fo[i + 1]], "pred": pred[po[i]:po[i + 1]], "valid": [int(j) for j in np.where(np.isin(a, valid_cats))[0]]})
    return out


def test_dual_iou_matches_reference(fx):
    got = np.asarray([R.calculate_iou(a, b) for a, b in zip(fx["iou_b1"], fx["iou_b2"])])
    assert np.array_equal(got, fx["iou_out"])
    assert np.allclose(fx["iou_out"][:40], 1.0, rtol=0, atol=1e-12) and (fx["iou_out"][40:60] == 0.0).all()


@pytest.mark.parametrize("ds", range(4))
@pytest.mark.parametrize("si", range(3))
def test_dual_evaluate_single_set_matches_reference(fx, ds, si):
    res = R.evaluate_single_set(dual_images(fx, ds, fx[f"dual{ds}_{si}_valid_cats"]), 0.5, 0.25)
    got = np.asarray([res[k] for k in ("total_gt", "total_pred", "true_positives", "false_positives", "false_negatives", "precision", "recall", "f1_score", "ap")], np.float64)
    assert np.array_equal(got, fx[f"dual{ds}_{si}_res"]), (got, fx[f"dual{ds}_{si}_res"])


@pytest.mark.parametrize("k", range(5))
def test_dual_ap11_matches_reference(fx, k):
    assert R.average_precision_11pt(fx[f"ap11_{k}_conf"], fx[f"ap11_{k}_tp"], int(fx[f"ap11_{k}_total"])) == float(fx[f"ap11_{k}_out"])


@pytest.mark.parametrize("k", range(4))
def test_official_numpy_parts_match_reference(fx, k):
    T = int(fx[f"pr_{k}_T"])
    assert np.array_equal(R.img_pr_info(T, fx[f"pr_{k}_pred"], fx[f"pr_{k}_prop"], fx[f"pr_{k}_rec"]), fx[f"pr_{k}_out"])
    assert np.array_equal(R.dataset_pr_info(T, fx[f"dpr_{k}_counts"], int(fx[f"dpr_{k}_faces"])), fx[f"dpr_{k}_out"])
    assert R.voc_ap(fx[f"ap_{k}_rec"], fx[f"ap_{k}_prec"]) == float(fx[f"ap_{k}_out"])


def test_bbox_overlaps_and_image_eval_micro_cases():
    """Hand-computed cases for the parts that could not be pinned (the Cython bbox_overlaps is not in the reference tree)."""
    a = np.asarray([[0, 0, 9, 9]], np.float64)                       # inclusive pixels: a 10x10 box
    assert R.bbox_overlaps(a, np.asarray([[0, 0, 9, 9]]))[0, 0] == 1.0
    assert R.bbox_overlaps(a, np.asarray([[5, 0, 14, 9]]))[0, 0] == 50 / 150
    assert R.bbox_overlaps(a, np.asarray([[10, 0, 19, 9]]))[0, 0] == 0.0          # adjacent pixels do not overlap
    assert R.bbox_overlaps(a, np.asarray([[9, 9, 12, 12]]))[0, 0] == 1 / (100 + 16 - 1)
    # two predictions on one face: the second is neither a new match nor dropped; a match with an ignored face drops the proposal
    gt = np.asarray([[0, 0, 10, 10], [100, 100, 10, 10]], np.float64)
    pred = np.asarray([[0, 0, 10, 10, .9], [1, 0, 10, 10, .8], [100, 100, 10, 10, .7], [300, 300, 5, 5, .6]], np.float64)
    rec, prop = R.image_eval(pred, gt, np.asarray([1, 0]), 0.5)
    assert rec.tolist() == [1, 1, 1, 1] and prop.tolist() == [1, 1, -1, 1]
    rec, prop = R.image_eval(pred, gt, np.asarray([1, 1]), 0.5)
    assert rec.tolist() == [1, 1, 2, 2] and prop.tolist() == [1, 1, 1, 1]
    ap, recall, prec, counts, n = R.evaluate_setting([{"pred": pred, "gt": gt, "keep": np.asarray([1, 2])}], 10, 0.5)
    assert n == 2 and counts[0].tolist() == [1, 1] and counts[3].tolist() == [4, 2] and recall[-1] == 1.0 and prec[-1] == 0.5


def test_adaptive_slice_size_and_difficulty_map_match_reference(fx):
    """The drop-in evaluator classes (compat/eval) against the reference's own helper outputs."""
    import ffp_amd.compat
    ffp_amd.compat.install()
    from eval.eval_dual import DualWiderFaceEvaluator
    from eval.eval_official_widerface import OfficialWiderFaceEvaluator
    o = OfficialWiderFaceEvaluator.__new__(OfficialWiderFaceEvaluator)
    assert [o._get_slice_size_adaptive(int(w), int(h)) for w, h in fx["adaptive_dims"]] == fx["adaptive_official"].tolist() == fx["adaptive_dual"].tolist()
    d = DualWiderFaceEvaluator(subcategory_gt={})
    subcats = ["large_clear", "large_degraded", "medium_clear", "medium_degraded", "small_clear", "small_degraded"]
    got = [[int(x in d.map_subcategory_to_difficulty(c)) for x in ("easy", "medium", "hard")] for c in subcats]
    assert got == fx["difficulty_map"].tolist()


@pytest.mark.parametrize("k", range(5))
def test_compat_ap11_and_voc_ap_match_reference(fx, k):
    """The shipped evaluator classes' host-side arithmetic (vectorised) against the reference's own outputs."""
    import ffp_amd.compat
    ffp_amd.compat.install()
    from eval.eval_dual import DualWiderFaceEvaluator
    from eval.eval_official_widerface import OfficialWiderFaceEvaluator
    d = DualWiderFaceEvaluator(subcategory_gt={})
    dets = [{"confidence": float(c), "is_tp": bool(t)} for c, t in zip(fx[f"ap11_{k}_conf"], fx[f"ap11_{k}_tp"])]
    assert d.calculate_average_precision(dets, int(fx[f"ap11_{k}_total"])) == float(fx[f"ap11_{k}_out"])
    if k < 4:
        o = OfficialWiderFaceEvaluator.__new__(OfficialWiderFaceEvaluator)
        o.thresh_num = int(fx[f"pr_{k}_T"])
        assert o._voc_ap(fx[f"ap_{k}_rec"], fx[f"ap_{k}_prec"]) == float(fx[f"ap_{k}_out"])
        assert np.array_equal(o._dataset_pr_info(fx[f"dpr_{k}_counts"], int(fx[f"dpr_{k}_faces"])), fx[f"dpr_{k}_out"])
