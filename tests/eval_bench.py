"""WIDER-val-sized synthetic evaluation: ffp_eval_wider_pr / ffp_eval_dual_match (host arrays in, counts out, upload included) against the
numpy oracle on a slice of the same data. 3226 images, faces per image ~ the val split's long tail (median 3, a few crowds > 500)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib
from oracle import widerface_eval_ref as R

rng = np.random.default_rng(0)
n_img = 3226
preds, gts, ev = [], [], []
for i in range(n_img):
    nf = int(min(1500, max(1, rng.lognormal(1.3, 1.3))))
    faces = np.concatenate([rng.uniform(0, 1000, (nf, 2)), rng.uniform(6, 120, (nf, 2))], 1)
    npred = int(nf * 1.5) + int(rng.integers(0, 30))
    src = faces[rng.integers(0, nf, npred)]
    p = np.concatenate([src + rng.normal(0, 0.08, (npred, 4)) * src[:, [2, 3, 2, 3]], rng.uniform(0.01, 1, (npred, 1))], 1)
    p[:, 2:4] = np.maximum(p[:, 2:4], 1)
    preds.append(p[np.argsort(-p[:, 4], kind="stable")]); gts.append(faces); ev.append((rng.random(nf) < 0.8).astype(np.uint8))
tot_p, tot_g = sum(map(len, preds)), sum(map(len, gts))
pairs = sum(len(p) * len(g) for p, g in zip(preds, gts))
_lib.eval_wider_pr(preds[:8], gts[:8], ev[:8])
t0 = time.perf_counter(); c = _lib.eval_wider_pr(preds, gts, ev, 0.5, 1000); t1 = time.perf_counter()
f = _lib.eval_dual_match(preds, gts, ev, 0.5); t2 = time.perf_counter()
print(f"{n_img} images, {tot_p} predictions, {tot_g} faces, {pairs / 1e6:.1f} M box pairs")
print(f"GPU official protocol (match + 1000-threshold PR counts): {(t1 - t0) * 1e3:.1f} ms host to host; dual matching: {(t2 - t1) * 1e3:.1f} ms")
sub = [i for i in range(n_img) if len(gts[i]) <= 40][:150]
t0 = time.perf_counter()
imgs = [{"pred": preds[i], "gt": gts[i], "keep": np.where(ev[i])[0] + 1} for i in sub]
_, _, _, counts, _ = R.evaluate_setting(imgs, 1000, 0.5)
dt = time.perf_counter() - t0
sp = sum(len(preds[i]) * len(gts[i]) for i in sub)
print(f"numpy oracle on {len(sub)} small images ({sp / 1e6:.2f} M pairs): {dt:.2f} s -> {dt * pairs / sp:.0f} s extrapolated by pairs for the whole set (1 core)")
g = _lib.eval_wider_pr([preds[i] for i in sub], [gts[i] for i in sub], [ev[i] for i in sub], 0.5, 1000)
print("subset counts identical:", bool(np.array_equal(g, counts.astype(np.int64))))
