"""GPU parity of the WIDER FACE evaluation kernels (csrc/eval.hip through ffp_eval_wider_pr / ffp_eval_dual_match) — SURVEY.md §8 f4.
Integer results, so the bar is bit-exact: against oracle/widerface_eval_ref.py on seeded datasets (ties, duplicates, unsorted scores,
empty images, crowds) and against tests/golden/eval_expected.npz, i.e. the outputs of the reference's own `evaluate_single_set`."""
import os

import numpy as np
import pytest

from oracle import widerface_eval_ref as R

pytestmark = pytest.mark.gpu
FX = os.path.join(os.path.dirname(__file__), "golden", "eval_expected.npz")


def random_dataset(rng, n_img, max_faces, max_pred, integer, sort_scores):
    imgs = []
    for _ in range(n_img):
        nf, npred = int(rng.integers(0, max_faces + 1)), int(rng.integers(0, max_pred + 1))
        faces = np.concatenate([rng.uniform(0, 1000, (nf, 2)), rng.uniform(4, 150, (nf, 2))], 1)
        pred = np.zeros((npred, 5))
        for i in range(npred):
            if nf and rng.random() < 0.75:
                f = faces[rng.integers(nf)]
                pred[i, :4] = f + rng.normal(0, 0.1, 4) * f[[2, 3, 2, 3]] * (rng.random() < 0.8)
            else:
                pred[i, :4] = np.concatenate([rng.uniform(0, 1000, 2), rng.uniform(4, 150, 2)])
            pred[i, 4] = np.round(rng.uniform(0.01, 1.0), 2 if rng.random() < 0.5 else 7)
        if integer:
            faces, pred[:, :4] = np.round(faces), np.round(pred[:, :4])
        pred[:, 2:4] = np.maximum(pred[:, 2:4], 1)
        if sort_scores and npred:
            pred = pred[np.argsort(-pred[:, 4], kind="stable")]
        keep = np.sort(rng.choice(nf, int(rng.integers(0, nf + 1)), replace=False)) + 1 if nf else np.zeros(0, np.int64)
        imgs.append({"pred": pred, "gt": faces, "keep": keep.astype(np.int64)})
    return imgs


@pytest.mark.parametrize("case", [(40, 12, 30, False, True, 1000), (25, 8, 20, True, True, 1000), (30, 10, 25, True, False, 37), (6, 300, 500, False, True, 1000),
                                  (3, 0, 5, False, True, 100)], ids=lambda c: "img%d_f%d_p%d_int%d_sorted%d_T%d" % c)
def test_official_protocol_counts_match_oracle(gpu_lib, case):
    n_img, mf, mp, integer, srt, T = case
    imgs = random_dataset(np.random.default_rng(hash(case) % 2 ** 31), n_img, mf, mp, integer, srt)
    ap, recall, prec, counts, n_faces = R.evaluate_setting(imgs, T, 0.5)
    ev = []
    for im in imgs:
        e = np.zeros(len(im["gt"]), np.uint8)
        e[im["keep"] - 1] = 1
        ev.append(e)
    use = [i for i, im in enumerate(imgs) if len(im["pred"]) and len(im["gt"])]
    got = gpu_lib.eval_wider_pr([imgs[i]["pred"] for i in use], [imgs[i]["gt"] for i in use], [ev[i] for i in use], 0.5, T)
    assert np.array_equal(got, counts.astype(np.int64))
    # handing over the skipped images too changes nothing (the kernel applies the reference's `continue`)
    got_all = gpu_lib.eval_wider_pr([im["pred"] for im in imgs], [im["gt"] for im in imgs], ev, 0.5, T)
    assert np.array_equal(got_all, got)
    if n_faces:
        pr = R.dataset_pr_info(T, got.astype(np.float64), n_faces)
        assert R.voc_ap(pr[:, 1], pr[:, 0]) == ap


def test_official_evaluator_class_from_mat_files(gpu_lib, tmp_path):
    """The drop-in class on .mat ground truth in the official nested layout (written with scipy) == oracle on the same data."""
    from scipy.io import savemat
    import ffp_amd.compat
    ffp_amd.compat.install()
    from eval.eval_official_widerface import OfficialWiderFaceEvaluator
    rng = np.random.default_rng(11)
    events = ["0--Parade", "1--Handshaking", "2--Demonstration"]
    per_event = [random_dataset(rng, n, 9, 25, True, True) for n in (4, 3, 5)]

    def cell(items):
        c = np.empty((len(items), 1), object)
        for i, x in enumerate(items):
            c[i, 0] = x
        return c
    names = [[f"{e.split('--')[0]}_img_{k}" for k in range(len(per_event[ei]))] for ei, e in enumerate(events)]
    gt = {"event_list": cell([np.array([e]) for e in events]),
          "file_list": cell([cell([np.array([n]) for n in names[ei]]) for ei in range(3)]),
          "face_bbx_list": cell([cell([im["gt"] for im in per_event[ei]]) for ei in range(3)])}
    savemat(tmp_path / "wider_face_val.mat", gt)
    settings = {}
    for si, s in enumerate(("easy", "medium", "hard")):
        keeps = [[(im["keep"][: max(0, len(im["keep"]) - (2 - si))] if si < 2 else im["keep"]).reshape(-1, 1).astype(np.int32) for im in per_event[ei]] for ei in range(3)]
        settings[s] = keeps
        savemat(tmp_path / f"wider_{s}_val.mat", {"gt_list": cell([cell(k) for k in keeps])})
    ev = OfficialWiderFaceEvaluator(gt_path=str(tmp_path), images_path=str(tmp_path), load_model=False)
    preds = {e: {names[ei][k]: im["pred"] for k, im in enumerate(per_event[ei])} for ei, e in enumerate(events)}
    res = ev.run(all_predictions=preds)
    for s in ("easy", "medium", "hard"):
        flat = [{"pred": im["pred"], "gt": im["gt"], "keep": settings[s][ei][k].reshape(-1)} for ei in range(3) for k, im in enumerate(per_event[ei])]
        assert res[s] == R.evaluate_setting(flat, 1000, 0.5)[0]


@pytest.mark.parametrize("ds", range(4))
def test_dual_protocol_reproduces_reference_outputs(gpu_lib, ds):
    """flags from the GPU + the host-side AP == the numbers the reference's own evaluate_single_set produced on the same inputs."""
    import ffp_amd.compat
    ffp_amd.compat.install()
    from eval.eval_dual import DualWiderFaceEvaluator
    fx = np.load(FX)
    cats = ["large_clear", "large_degraded", "medium_clear", "small_hard"]
    faces, fo, pred, po, assign = (fx[f"dual{ds}_{k}"] for k in ("faces", "face_off", "pred", "pred_off", "assign"))
    gt, preds = {}, {}
    for i in range(len(fo) - 1):
        a = assign[fo[i]:fo[i + 1]]
        name = f"img{i}"
        gt[name] = {c: [int(j) for j in np.where(a == ci)[0]] for ci, c in enumerate(cats)}
        gt[name]["all_faces"] = [{"bbox": f.tolist()} for f in faces[fo[i]:fo[i + 1]]]
        preds[name] = [{"bbox": p[:4].tolist(), "confidence": float(p[4])} for p in pred[po[i]:po[i + 1]]]
    ev = DualWiderFaceEvaluator(subcategory_gt=gt, predictions=preds, iou_threshold=0.5, global_confidence=0.25)
    for si in range(3):
        valid_cats = [cats[c] for c in fx[f"dual{ds}_{si}_valid_cats"]]
        res = ev.evaluate_single_set("difficulty", f"set{si}", valid_cats)
        got = np.asarray([res[k] for k in ("total_gt", "total_pred", "true_positives", "false_positives", "false_negatives", "precision", "recall", "f1_score", "ap")], np.float64)
        assert np.array_equal(got, fx[f"dual{ds}_{si}_res"]), (si, got, fx[f"dual{ds}_{si}_res"])


def test_dual_match_flags_match_oracle_on_crowds(gpu_lib):
    rng = np.random.default_rng(3)
    imgs = random_dataset(rng, 20, 150, 300, True, False)
    preds, faces, valid, exp = [], [], [], []
    for im in imgs:
        v = np.zeros(len(im["gt"]), np.uint8)
        v[im["keep"] - 1] = 1
        preds.append(im["pred"]); faces.append(im["gt"]); valid.append(v)
        if v.any():
            exp.append(R.match_image(im["pred"], im["gt"][v == 1], im["gt"][v == 0], 0.5))
        else:
            exp.append(np.full(len(im["pred"]), 2, np.int32))
    got = gpu_lib.eval_dual_match(preds, faces, valid, 0.5)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)
