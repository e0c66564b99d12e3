"""FaceEnhancer / RealESRGANer / RRDBNet with the reference's surface (utils/enhancer.py:21-478) over libffp.so.

`enhance_image` keeps the reference's error convention — it never raises and returns `(image, success)` (:189-235).
File I/O uses Pillow (cv2 is not a dependency of this build): BGR arrays in, BGR arrays out, JPEG quality honoured.
"""
from __future__ import annotations

import os
import time

import numpy as np
from PIL import Image

import ffp_amd  # noqa: F401
from ffp_amd import _lib, synth, weights_io


class RRDBNet:
    """Architecture descriptor (basicsr.archs.rrdbnet_arch.RRDBNet signature); the network itself lives in libffp.so."""

    def __init__(self, num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32):
        if (num_in_ch, num_out_ch, num_feat, num_grow_ch) != (3, 3, 64, 32):
            raise ValueError("only RRDBNet(3, 3, num_feat=64, num_grow_ch=32) is built")
        self.scale, self.num_block = scale, num_block


class RealESRGANer:
    """realesrgan.RealESRGANer(scale, model_path, dni_weight, model, tile, tile_pad, pre_pad, half, gpu_id) -> .enhance()."""

    def __init__(self, scale, model_path, dni_weight=None, model=None, tile=0, tile_pad=10, pre_pad=0, half=False, device=None, gpu_id=None):
        self.scale, self.tile_size, self.tile_pad, self.pre_pad, self.half = scale, tile, tile_pad, pre_pad, half
        if model_path is None:
            raise AttributeError("'NoneType' object has no attribute 'startswith'")   # what realesrgan 0.3.0 does
        nb = model.num_block if model is not None else 23
        if isinstance(model_path, dict):
            W = model_path
        elif str(model_path).startswith("synthetic:"):
            W = synth.rrdbnet_weights(scale, nb)
        elif str(model_path).startswith("https://"):
            raise RuntimeError("weight download is not available offline; pass a local .pth / .ffpw")
        elif str(model_path).endswith(".ffpw"):
            W = weights_io.load(model_path)
        else:
            W = weights_io.load_esrgan_pth(model_path)
        self._sr = _lib.Enhancer(W, scale, nb, device=gpu_id or 0, half=bool(half))

    def enhance(self, img, outscale=None, alpha_upsampler="realesrgan"):
        if img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
            raise ValueError("this build enhances 3-channel uint8 BGR images (the face-crop path)")
        if outscale is not None and float(outscale) != float(self.scale):
            raise NotImplementedError("outscale != scale (Lanczos resize) is never used by the reference (utils/enhancer.py:214)")
        return self._sr.enhance(np.ascontiguousarray(img), self.tile_size, self.tile_pad, self.pre_pad), "RGB"


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


class FaceEnhancer:
    def __init__(self, model_name="RealESRGAN_x4plus", model_path=None, scale=4, tile=400, half=True):
        self.model_name, self.scale, self.tile, self.half, self.upsampler = model_name, scale, tile, half, None
        self.device = self._check_device()
        print(f"Using device: {self.device}")
        try:
            self._setup_model(model_name, model_path)
        except Exception as e:
            print(f" Failed to setup model: {e}")
            raise

    def _check_device(self):
        if _lib.device_count() > 0:
            return "cuda"
        return "cpu"

    def _find_model_path(self, model_name):
        for p in (f"models/{model_name}.pth", f"./models/{model_name}.pth", f"../models/{model_name}.pth", f"weights/{model_name}.pth",
                  f"./weights/{model_name}.pth", f"{model_name}.pth", f"./{model_name}.pth"):
            if os.path.exists(p):
                return os.path.abspath(p)
        return None

    def _setup_model(self, model_name, model_path):
        if self.device == "cpu":
            raise RuntimeError("no MI355X visible: this build has no CPU enhancement path")
        if model_path is None:
            model_path = self._find_model_path(model_name)
        if "anime_6B" in model_name:
            model = RRDBNet(3, 3, num_feat=64, num_block=6, num_grow_ch=32, scale=self.scale)
        elif "x2" in model_name:
            model = RRDBNet(3, 3, num_feat=64, num_block=23, num_grow_ch=32, scale=2)
            self.scale = 2                                           # reference quirk kept (utils/enhancer.py:109-119)
        else:
            model = RRDBNet(3, 3, num_feat=64, num_block=23, num_grow_ch=32, scale=self.scale)
        self.upsampler = RealESRGANer(scale=self.scale, model_path=model_path, dni_weight=None, model=model, tile=self.tile, tile_pad=10,
                                      pre_pad=0, half=self.half, gpu_id=0)

    def enhance_image(self, image):
        if self.upsampler is None:
            return image, False
        try:
            if isinstance(image, Image.Image):
                image = np.asarray(image.convert("RGB"))[..., ::-1].copy()
            if image is None or image.size == 0:
                return image, False
            h, w = image.shape[:2]
            if h < 4 or w < 4:
                print(f" Image too small ({w}x{h}), skipping enhancement")
                return image, False
            out, _ = self.upsampler.enhance(image, outscale=self.scale)
            return out, True
        except Exception as e:
            print(f" Enhancement failed: {type(e).__name__}: {e}")
            return image, False

    def enhance_images(self, images):
        """Extension (not in the reference): enhance_image over a list as ONE ragged GPU batch -> [(image, ok)], the same bytes as len(images)
        calls of enhance_image (a crop's result does not depend on its batch: tests/test_gpu_sr_crops.py). enhance_face_crops_batch uses it."""
        out = [(im, False) for im in images]
        if self.upsampler is None or not hasattr(self.upsampler, "_sr"):
            return [self.enhance_image(im) for im in images]
        good = []
        for k, im in enumerate(images):
            if isinstance(im, Image.Image):
                im = np.asarray(im.convert("RGB"))[..., ::-1].copy()
            if im is None or im.size == 0 or im.ndim != 3 or im.shape[2] != 3 or im.dtype != np.uint8 or im.shape[0] < 4 or im.shape[1] < 4:
                if im is not None and im.size and im.ndim == 3 and (im.shape[0] < 4 or im.shape[1] < 4):
                    print(f" Image too small ({im.shape[1]}x{im.shape[0]}), skipping enhancement")
                continue
            if max(im.shape[0], im.shape[1]) > self.tile > 0:          # larger than one tile: the tiled single-image path
                out[k] = self.enhance_image(im)
                continue
            good.append((k, np.ascontiguousarray(im)))
        if good:
            try:
                res = self.upsampler._sr.enhance_batch([im for _, im in good])
                for (k, _), r in zip(good, res):
                    out[k] = (r, True)
            except Exception as e:
                print(f" Enhancement failed: {type(e).__name__}: {e}")
                for k, im in good:
                    out[k] = self.enhance_image(im)
        return out

    def enhance_face_crop(self, crop_path, output_path, quality=95):
        info = {"original_path": crop_path, "output_path": output_path, "original_size": None, "enhanced_size": None,
                "scale_factor": self.scale, "success": False}
        try:
            if not os.path.exists(crop_path):
                return False, info
            img = _imread_bgr(crop_path)
            if img is None:
                return False, info
            info["original_size"] = (img.shape[1], img.shape[0])
            out, ok = self.enhance_image(img)
            if not ok:
                return False, info
            info["enhanced_size"] = (out.shape[1], out.shape[0])
            os.makedirs(os.path.dirname(output_path) or ".", exist_ok=True)
            if not _imwrite_bgr(output_path, out, quality):
                return False, info
            info["success"] = True
            return True, info
        except Exception as e:
            print(f" Error enhancing {os.path.basename(crop_path)}: {e}")
            return False, info

    def get_model_info(self):
        return {"model_name": self.model_name, "scale": self.scale, "tile": self.tile, "half_precision": self.half, "device": self.device,
                "is_loaded": self.upsampler is not None, "backend": f"libffp {_lib.lib().ffp_version()}", "cuda_available": self.device == "cuda"}


def enhance_face_crops_batch(crops_dir, enhancer, prefix="enhanced", progress_callback=None):
    results = {"enhanced_files": [], "failed_files": [], "enhancement_info": [],
               "statistics": {"total_files": 0, "successful": 0, "failed": 0, "total_time": 0}}
    if not os.path.exists(crops_dir):
        return results
    out_dir = os.path.join(os.path.dirname(crops_dir), f"{prefix}_enhanced")
    os.makedirs(out_dir, exist_ok=True)
    files = [f for f in os.listdir(crops_dir) if f.lower().endswith((".png", ".jpg", ".jpeg", ".bmp", ".tiff"))]
    results["statistics"]["total_files"] = len(files)
    t0 = time.time()
    batched = {}
    if type(enhancer) is FaceEnhancer and len(files) > 1:
        # every readable crop through ONE ragged GPU batch (same bytes as the per-file loop below, which still runs for whatever the batch
        # could not take: unreadable files, failures, an enhancer subclass with its own enhance_face_crop)
        try:
            imgs = [(f, _imread_bgr(os.path.join(crops_dir, f))) for f in files]
            outs = enhancer.enhance_images([im for _, im in imgs if im is not None])
            it = iter(outs)
            for f, im in imgs:
                if im is not None:
                    o, ok = next(it)
                    if ok:
                        batched[f] = (im, o)
        except Exception:
            batched = {}
    for i, f in enumerate(files, 1):
        name, ext = os.path.splitext(f)
        dst = os.path.join(out_dir, f"{prefix}_{name}{ext}")
        if progress_callback:
            try:
                progress_callback(i, len(files), f)
            except Exception:
                pass
        ok, info = False, None
        if f in batched:
            im, o = batched[f]
            info = {"original_path": os.path.join(crops_dir, f), "output_path": dst, "original_size": (im.shape[1], im.shape[0]),
                    "enhanced_size": (o.shape[1], o.shape[0]), "scale_factor": enhancer.scale, "success": False}
            try:
                os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
                ok = bool(_imwrite_bgr(dst, o, 95))
                info["success"] = ok
            except Exception:
                ok = False
        for _attempt in range(0 if ok else 2):                       # reference retries once (utils/enhancer.py:362-377)
            try:
                ok, info = enhancer.enhance_face_crop(os.path.join(crops_dir, f), dst)
                if ok:
                    break
            except Exception:
                ok = False
        if ok and info:
            results["enhanced_files"].append(dst)
            results["enhancement_info"].append(info)
            results["statistics"]["successful"] += 1
        else:
            results["failed_files"].append(os.path.join(crops_dir, f))
            results["statistics"]["failed"] += 1
    results["statistics"]["total_time"] = time.time() - t0
    return results


def create_enhancement_summary(results, output_path):
    """The batch report in the reference's text layout (utils/enhancer.py:409-452), line for line (pinned by tests/golden/wrapper_expected.json)."""
    import datetime
    st = results["statistics"]
    n = max(st["total_files"], 1)
    lines = ["", "=== LAPORAN ENHANCEMENT WAJAH ===", f"Generated: {datetime.datetime.now().strftime('%Y-%m-%d %H:%M:%S')}", "",
             "--- RINGKASAN STATISTIK ---", f"Total File Diproses: {st['total_files']}", f" Berhasil: {st['successful']}", f" Gagal: {st['failed']}",
             f" Tingkat Keberhasilan: {(st['successful'] / n * 100):.1f}%", f" Waktu Total: {st['total_time']:.2f} detik",
             f" Waktu Rata-rata per File: {(st['total_time'] / n):.2f} detik", "", "--- DETAIL FILE BERHASIL ---"]
    text = "\n".join(lines) + "\n"
    for i, info in enumerate(results["enhancement_info"], 1):
        o, e = info["original_size"], info["enhanced_size"]
        text += (f"\nFile #{i}: {os.path.basename(info['original_path'])}\n   Ukuran Asli: {o[0]}x{o[1]} px\n   Ukuran Enhanced: {e[0]}x{e[1]} px\n"
                 f"   Scale Factor: {info['scale_factor']}x\n   Output: {os.path.basename(info['output_path'])}\n")
    if results["failed_files"]:
        text += f"\n--- FILE GAGAL ({len(results['failed_files'])}) ---\n"
        for i, f in enumerate(results["failed_files"], 1):
            text += f"{i}. {os.path.basename(f)}\n"
    try:
        os.makedirs(os.path.dirname(output_path), exist_ok=True)
        with open(output_path, "w", encoding="utf-8") as fh:
            fh.write(text)
        print(f" Enhancement summary saved to: {os.path.basename(output_path)}")
    except Exception as e:
        print(f" Error saving enhancement summary: {e}")


def get_available_models():
    """The reference's model table (utils/enhancer.py:454-480) plus `num_block`, which this build needs to lay the network out."""
    return {
        "RealESRGAN_x4plus": {"description": "Model utama untuk gambar natural (4x)", "scale": 4, "best_for": "Foto natural, potret, wajah", "file_size": "~65MB",
                              "recommended_tile": 400, "num_block": 23},
        "RealESRGAN_x2plus": {"description": "Model untuk enhancement 2x (lebih cepat)", "scale": 2, "best_for": "Enhancement ringan, GPU terbatas", "file_size": "~65MB",
                              "recommended_tile": 600, "num_block": 23},
        "RealESRGAN_x4plus_anime_6B": {"description": "Model khusus untuk anime/kartun (4x)", "scale": 4, "best_for": "Gambar anime, kartun, ilustrasi", "file_size": "~18MB",
                                       "recommended_tile": 400, "num_block": 6},
    }
