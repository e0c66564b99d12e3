"""Shared helpers for the parity tests (oracle = checker, HIP library = system under test)."""
import numpy as np


def iou_xyxy(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Row-wise IoU of two (n,4) float arrays."""
    a = a.astype(np.float64); b = b.astype(np.float64)
    iw = np.clip(np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0]), 0, None)
    ih = np.clip(np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]), 0, None)
    inter = iw * ih
    ua = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(ua > 0, inter / ua, 1.0)


def psnr_u8(a: np.ndarray, b: np.ndarray) -> float:
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def match_by_iou(a: np.ndarray, b: np.ndarray, sa: np.ndarray = None, sb: np.ndarray = None):
    """Greedy one-to-one matching of box sets a (n,4), b (m,4) by best IoU. Returns list of (i, j, iou).
    With scores sa / sb, candidates whose IoU is within 1e-6 of the best are told apart by the closest score (several
    detections can share one box after clipping to the image: their pairing is otherwise arbitrary)."""
    out, used = [], set()
    for i in range(a.shape[0]):
        ious = iou_xyxy(np.repeat(a[i:i + 1], b.shape[0], 0), b) if b.shape[0] else np.zeros(0)
        free = [j for j in range(b.shape[0]) if j not in used]
        if not free:
            break
        best = max(ious[j] for j in free)
        cand = [j for j in free if ious[j] >= best - 1e-6]
        j = cand[0] if sa is None else min(cand, key=lambda q: abs(float(sb[q]) - float(sa[i])))
        used.add(int(j)); out.append((i, int(j), float(ious[j])))
    return out
