"""utils.visualization with the reference's surface (/root/reference/utils/visualization.py:37-285) on Pillow + numpy:
`draw_detections`, `draw_detections_on_image`, `draw_keypoints_and_skeleton`, `save_face_crops`, `create_detection_summary`
and the landmark tables, so that the import block of pipeline_v4_yolo/app_yolo_sahi.py:13-17 and
pipeline_v1_detection_first/app_v1.py:12 resolves and the scripts' output files (overlay JPEG, crops, summary text) appear
where the reference puts them. `save_face_crops` is the part ON the hot path (it defines the crops the enhancer sees,
:185-223: int box, clamp to the image, skip empty); drawing is presentation code, rendered with PIL primitives (pixel
values of lines and glyphs are not part of any parity claim). Arrays are BGR like cv2's.
"""
import os

import numpy as np
from PIL import Image, ImageDraw

# 5 face landmarks in WIDER FACE order, their skeleton and per-point colours (BGR), as the reference draws them (:5-35)
FACE_KEYPOINT_NAMES = ["left_eye", "right_eye", "nose", "left_mouth", "right_mouth"]
FACE_SKELETON = [[0, 1], [0, 2], [1, 2], [2, 3], [2, 4], [3, 4]]
FACE_KEYPOINT_COLORS = [(255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0), (255, 0, 255)]
SKELETON_COLOR = (0, 255, 255)


def _rgb(bgr):
    return (int(bgr[2]), int(bgr[1]), int(bgr[0]))


def _imread_bgr(path):
    """cv2.imread (a real OpenCV, or this build's shim: JPEG decoded on the GPU, pixel-identical to libjpeg-turbo): BGR or None."""
    try:
        import cv2
        return cv2.imread(str(path))
    except Exception:
        return None


def _imwrite_bgr(path, img, quality=95):
    """cv2.imwrite with [IMWRITE_JPEG_QUALITY, quality] (the shim encodes .jpg on the GPU, byte-identical to libjpeg-turbo)."""
    try:
        import cv2
        return bool(cv2.imwrite(str(path), np.ascontiguousarray(img), [cv2.IMWRITE_JPEG_QUALITY, int(quality)]))
    except Exception:
        return False


def _draw_kpts(draw, keypoints, thr, skeleton):
    if keypoints is None or len(keypoints) == 0:
        return
    if skeleton:
        for a, b in FACE_SKELETON:
            if a < len(keypoints) and b < len(keypoints) and keypoints[a][2] > thr and keypoints[b][2] > thr:
                draw.line([(int(keypoints[a][0]), int(keypoints[a][1])), (int(keypoints[b][0]), int(keypoints[b][1]))], fill=_rgb(SKELETON_COLOR), width=2)
    for i, (x, y, c) in enumerate(np.asarray(keypoints)[:, :3]):
        if c > thr:
            col = FACE_KEYPOINT_COLORS[i] if i < len(FACE_KEYPOINT_COLORS) else (255, 255, 255)
            x, y = int(x), int(y)
            draw.ellipse([x - 3, y - 3, x + 3, y + 3], outline=(255, 255, 255))
            draw.ellipse([x - 2, y - 2, x + 2, y + 2], fill=_rgb(col))


def draw_keypoints_and_skeleton(image, keypoints, confidence_threshold=0.3, draw_skeleton=True):
    """In the reference this draws into `image` and returns it (:37-76); same here for a BGR ndarray."""
    if keypoints is None or len(keypoints) == 0:
        return image
    pil = Image.fromarray(np.ascontiguousarray(image[..., ::-1]))
    _draw_kpts(ImageDraw.Draw(pil), keypoints, confidence_threshold, draw_skeleton)
    image[...] = np.asarray(pil)[..., ::-1]
    return image


def _render(pil, result, show_confidence, show_keypoints, box_color, text_color, kpt_conf_threshold, draw_skeleton, verbose):
    draw = ImageDraw.Draw(pil)
    found = 0
    for idx, det in enumerate(result.object_prediction_list):
        x1, y1, x2, y2 = [int(c) for c in det.bbox.to_xyxy()]
        draw.rectangle([x1, y1, x2, y2], outline=_rgb(box_color), width=2)
        if show_confidence:
            label = f"Face: {det.score.value:.2f}"
            l, t, r, b = draw.textbbox((0, 0), label)
            draw.rectangle([x1, y1 - (b - t) - 10, x1 + (r - l), y1], fill=_rgb(box_color))
            draw.text((x1, y1 - (b - t) - 7), label, fill=_rgb(text_color))
        if show_keypoints:
            k = getattr(det, "keypoints", None)
            if k is not None:
                found += 1
                if verbose:
                    print(f"   ✓ Face #{idx + 1}: Keypoints shape = {np.asarray(k).shape}")
                _draw_kpts(draw, k, kpt_conf_threshold, draw_skeleton)
            elif verbose:
                print(f"   ⚠️  Face #{idx + 1}: NO keypoints attached!")
    return found


def draw_detections(image_path, result, output_path, show_confidence=True, show_keypoints=True, box_color=(0, 255, 0), text_color=(0, 0, 0),
                    kpt_conf_threshold=0.3, draw_skeleton=False):
    """Boxes, "Face: 0.xx" labels and landmarks over the image at `image_path`, written to `output_path` (:78-148)."""
    img = _imread_bgr(image_path)
    if img is None:
        print(f"❌ Error: Gagal membaca gambar dari {image_path}")
        return
    n = len(result.object_prediction_list)
    print(f"   📊 Jumlah deteksi untuk visualisasi = {n}")
    pil = Image.fromarray(np.ascontiguousarray(img[..., ::-1]))
    found = _render(pil, result, show_confidence, show_keypoints, box_color, text_color, kpt_conf_threshold, draw_skeleton, True)
    print(f"   📍 Total faces with keypoints: {found}/{n}")
    d = os.path.dirname(output_path)
    if d:
        os.makedirs(d, exist_ok=True)
    if _imwrite_bgr(output_path, np.asarray(pil)[..., ::-1]):
        print(f"   ✓ Hasil visualisasi disimpan ke: {output_path}")
    else:
        print(f"   ❌ Gagal menyimpan visualisasi ke: {output_path}")


def draw_detections_on_image(image, result, show_confidence=True, show_keypoints=True, box_color=(0, 255, 0), text_color=(255, 255, 255),
                             kpt_conf_threshold=0.3, draw_skeleton=False):
    """The same overlay on a BGR ndarray, returned as a new array (:151-183)."""
    pil = Image.fromarray(np.ascontiguousarray(image[..., ::-1]))
    _render(pil, result, show_confidence, show_keypoints, box_color, text_color, kpt_conf_threshold, draw_skeleton, False)
    return np.asarray(pil)[..., ::-1].copy()


def crop_boxes(result, width, height):
    """int box, clamped to the image; empty crops dropped. Returns [(index, x1, y1, x2, y2, score)] (:204-213)."""
    out = []
    for i, det in enumerate(result.object_prediction_list):
        x1, y1, x2, y2 = [int(c) for c in det.bbox.to_xyxy()]
        x1, y1, x2, y2 = max(0, x1), max(0, y1), min(width, x2), min(height, y2)
        if x2 > x1 and y2 > y1:
            out.append((i, x1, y1, x2, y2, det.score.value))
    return out


def save_face_crops(image_path, result, output_dir, prefix="face_crop"):
    img = _imread_bgr(image_path)
    if img is None:
        print(f"Error: Gagal membaca gambar dari {image_path}")
        return []
    os.makedirs(output_dir, exist_ok=True)
    paths = []
    for i, x1, y1, x2, y2, score in crop_boxes(result, img.shape[1], img.shape[0]):
        p = os.path.join(output_dir, f"{prefix}_{i + 1}_conf_{score:.2f}.jpg")
        _imwrite_bgr(p, img[y1:y2, x1:x2])
        paths.append(p)
    return paths


def create_detection_summary(result, image_path, processing_time, output_path, img_width, img_height, slice_width, slice_height):
    """The run's text report, line for line in the reference's layout (:225-285) — other tooling of the reference parses it."""
    preds = result.object_prediction_list
    scores = [p.score.value for p in preds]
    mean = float(np.mean(scores)) if scores else 0
    lines = ["", "=== Ringkasan Deteksi Wajah dengan Keypoints ===", "", "--- Informasi Proses ---",
             f"Gambar Sumber: {os.path.basename(image_path)}", f"Ukuran Gambar Asli: {img_width}x{img_height} px",
             f"Ukuran Slice: {slice_width}x{slice_height} px", f"Waktu Proses Total: {processing_time:.2f} detik", "",
             "--- Statistik Deteksi ---", f"Total Wajah Ditemukan: {len(preds)}", f"Rata-rata Skor Kepercayaan: {mean:.3f}",
             f"Skor Kepercayaan Minimum: {(min(scores) if scores else 0):.3f}", f"Skor Kepercayaan Maksimum: {(max(scores) if scores else 0):.3f}", "",
             "--- Detail Deteksi ---"]
    text = "\n".join(lines) + "\n"
    if not preds:
        text += "Tidak ada wajah yang terdeteksi.\n"
    for i, det in enumerate(preds):
        x1, y1, x2, y2 = [int(c) for c in det.bbox.to_xyxy()]
        text += f"\nWajah #{i + 1}:\n  - Bounding Box: [x1: {x1}, y1: {y1}, x2: {x2}, y2: {y2}]\n  - Skor Kepercayaan: {det.score.value:.3f}\n"
        k = getattr(det, "keypoints", None)
        if k is not None:
            text += "  - Keypoints:\n"
            for j, name in enumerate(FACE_KEYPOINT_NAMES):
                if j < len(k):
                    x, y, c = k[j]
                    text += f"      {name}: ({x:.1f}, {y:.1f}) [conf: {c:.3f}]\n"
    os.makedirs(os.path.dirname(output_path), exist_ok=True)
    with open(output_path, "w", encoding="utf-8") as fh:
        fh.write(text)
    print(f"✓ Summary disimpan ke: {output_path}")
