"""ctypes binding of libffp.so (include/ffp.h) + thin numpy-facing handles.

The library is the product: if it is missing or cannot be loaded every entry point raises — there is no Python or
PyTorch fallback for any operator.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import weights_io

_HERE = os.path.dirname(os.path.abspath(__file__))
# FFP_LIB=<path> loads a DIAGNOSTIC build instead (build.build_variant: in-kernel stamps, phase-skip instantiations). Probe scripts set it;
# the shipped libffp.so is never overwritten. Nothing else changes: same symbols, same checks, it is still the only implementation.
LIB_PATH = os.environ.get("FFP_LIB") or os.path.join(_HERE, "libffp.so")

PREC_F32, PREC_F16, PREC_F32X3 = 0, 1, 2
CHAN_AS_BGR, CHAN_AS_RGB = 0, 1
PP_NMS, PP_GREEDYNMM = 0, 1
METRIC_IOU, METRIC_IOS = 0, 1
PP_TYPES = {"NMS": PP_NMS, "GREEDYNMM": PP_GREEDYNMM}
METRICS = {"IOU": METRIC_IOU, "IOS": METRIC_IOS}

_lib = None


class FfpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libffp error {code}: {msg}")
        self.code = code


def _p(t):
    return C.POINTER(t)


_SIGS = {
    "ffp_last_error": (C.c_char_p, []),
    "ffp_version": (C.c_int, []),
    "ffp_device_count": (C.c_int, [_p(C.c_int)]),
    "ffp_slice_bboxes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _p(C.c_int32), C.c_int, _p(C.c_int32)]),
    "ffp_letterbox_geometry": (C.c_int, [C.c_int, C.c_int, C.c_int, _p(C.c_int32)]),
    "ffp_det_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p(C.c_void_p)]),
    "ffp_det_destroy": (None, [C.c_void_p]),
    "ffp_det_infer_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _p(C.c_int32), C.c_int, C.c_int, C.c_float,
                                      C.c_float, C.c_int, C.c_int, _p(C.c_float), _p(C.c_int32)]),
    "ffp_det_infer_tiles_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _p(C.c_int32), C.c_int, C.c_int, C.c_float,
                                          C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ffp_det_truncate_shift_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ffp_det_forward_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, _p(C.c_int32), C.c_int, C.c_int, _p(C.c_float),
                                      C.c_size_t, _p(C.c_int32)]),
    "ffp_merge": (C.c_int, [C.c_int, _p(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _p(C.c_float), _p(C.c_int32),
                            _p(C.c_int32)]),
    "ffp_merge_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                C.c_int, C.c_void_p]),
    "ffp_sliced_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                     C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _p(C.c_float),
                                     C.c_int, _p(C.c_int32)]),
    "ffp_det_stage_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                    C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    _p(C.c_int32), _p(C.c_int32)]),
    "ffp_det_last_ms": (C.c_int, [C.c_void_p, C.c_int, _p(C.c_float)]),
    "ffp_det_last_conv_stats": (C.c_int, [C.c_void_p, _p(C.c_double), _p(C.c_float), _p(C.c_int32)]),
    "ffp_det_set_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "ffp_det_profile_count": (C.c_int, [C.c_void_p, _p(C.c_int32)]),
    "ffp_det_profile_get": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, _p(C.c_double), _p(C.c_float), _p(C.c_int32)]),
    "ffp_det_profile_detail": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, _p(C.c_double), _p(C.c_float)]),
    "ffp_sr_profile_detail": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, _p(C.c_double), _p(C.c_float)]),
    "ffp_sr_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, _p(C.c_void_p)]),
    "ffp_sr_destroy": (None, [C.c_void_p]),
    "ffp_sr_enhance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ffp_sr_enhance_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ffp_sr_enhance_batch": (C.c_int, [C.c_void_p, C.c_int, _p(C.c_void_p), _p(C.c_int32), _p(C.c_int32), C.c_int, C.c_int, C.c_int,
                                       _p(C.c_void_p)]),
    "ffp_sr_enhance_crops_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _p(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_size_t, _p(C.c_int64)]),
    "ffp_sr_enhance_crops_dev_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _p(C.c_int32), C.c_int, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_size_t, _p(C.c_int64)]),
    "ffp_sr_wait": (C.c_int, [C.c_void_p]),
    "ffp_sr_enhance_crops_multi_dev_async": (C.c_int, [C.c_void_p, C.c_int, _p(C.c_void_p), _p(C.c_int32), C.c_int, C.c_int, _p(C.c_int32),
                                                       C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, _p(C.c_int64)]),
    "ffp_sr_plan_state": (C.c_int, [C.c_void_p, _p(C.c_int32), _p(C.c_int32)]),
    "ffp_det_graph_status": (C.c_int, [C.c_void_p, _p(C.c_int32)]),
    "ffp_sr_set_fused_body": (C.c_int, [C.c_void_p, C.c_int]),
    "ffp_det_drop_plans": (C.c_int, [C.c_void_p]),
    "ffp_sr_drop_plans": (C.c_int, [C.c_void_p]),
    "ffp_det_profile_bytes": (C.c_int, [C.c_void_p, C.c_int, _p(C.c_double)]),
    "ffp_sr_profile_bytes": (C.c_int, [C.c_void_p, C.c_int, _p(C.c_double)]),
    "ffp_conv_totals_enable": (C.c_int, [C.c_int]),
    "ffp_conv_totals_count": (C.c_int, [_p(C.c_int32)]),
    "ffp_conv_totals_get": (C.c_int, [C.c_int, C.c_char_p, C.c_int, _p(C.c_double), _p(C.c_double), _p(C.c_int64)]),
    "ffp_sr_mem_bytes": (C.c_int, [C.c_void_p, _p(C.c_uint64), _p(C.c_uint64), _p(C.c_int32)]),
    "ffp_det_mem_bytes": (C.c_int, [C.c_void_p, _p(C.c_uint64), _p(C.c_uint64), _p(C.c_int32)]),
    "ffp_det_set_lanes": (C.c_int, [C.c_void_p, C.c_int]),
    "ffp_det_stream_wait_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ffp_sr_last_ms": (C.c_int, [C.c_void_p, _p(C.c_float)]),
    "ffp_sr_last_conv_stats": (C.c_int, [C.c_void_p, _p(C.c_double), _p(C.c_float), _p(C.c_int32)]),
    "ffp_sr_set_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "ffp_sr_profile_count": (C.c_int, [C.c_void_p, _p(C.c_int32)]),
    "ffp_sr_profile_get": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, _p(C.c_double), _p(C.c_float), _p(C.c_int32)]),
    "ffp_eval_wider_pr": (C.c_int, [C.c_int, _p(C.c_double), _p(C.c_int64), _p(C.c_double), _p(C.c_int64), _p(C.c_uint8), C.c_int, C.c_double, C.c_int,
                                    _p(C.c_int64)]),
    "ffp_eval_dual_match": (C.c_int, [C.c_int, _p(C.c_double), _p(C.c_int64), _p(C.c_double), _p(C.c_int64), _p(C.c_uint8), C.c_int, C.c_double, _p(C.c_int32)]),
    "ffp_jpeg_encode": (C.c_int, [C.c_int, _p(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int, _p(C.c_uint8), C.c_int64, _p(C.c_int64)]),
    "ffp_jpeg_encode_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, _p(C.c_uint8), C.c_int64, _p(C.c_int64)]),
    "ffp_jpeg_encode_batch_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int, _p(C.c_int64), _p(C.c_int32), _p(C.c_int32), _p(C.c_int64), C.c_int, C.c_int, _p(C.c_uint8),
                                            C.c_int64, _p(C.c_int64)]),
    "ffp_jpeg_info": (C.c_int, [_p(C.c_uint8), C.c_int64, _p(C.c_int32), _p(C.c_int32), _p(C.c_int32)]),
    "ffp_jpeg_decode": (C.c_int, [C.c_int, _p(C.c_uint8), C.c_int64, C.c_int, _p(C.c_uint8), C.c_int64]),
    "ffp_jpeg_decode_dev": (C.c_int, [C.c_int, _p(C.c_uint8), C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int64]),
    "ffp_jpeg_decode_stats": (C.c_int, [_p(C.c_int64), _p(C.c_int64), _p(C.c_int64)]),
    "ffp_op_conv2d_shape": (C.c_int, [C.c_int]),
    "ffp_op_conv1x1_up2": (C.c_int, [C.c_int, C.c_int, _p(C.c_float), _p(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p(C.c_float),
                                     _p(C.c_float), C.c_int, C.c_int, _p(C.c_float)]),
    "ffp_op_conv2d_time": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, _p(C.c_float)]),
    "ffp_op_conv2d": (C.c_int, [C.c_int, C.c_int, _p(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, _p(C.c_float), _p(C.c_float), C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p(C.c_float), C.c_float, _p(C.c_float)]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)


def lib() -> C.CDLL:
    """Load libffp.so (in-tree). Raises if it has not been built — the product has no other implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m ffp_amd.build` "
                               "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.")
        # ONE HIP runtime per process. libffp.so needs `libamdhip64.so.7` and the PyTorch-ROCm wheel bundles a library with
        # that very SONAME (torch/lib/libamdhip64.so, with its own libhsa-runtime64 / librccl next to it): the dynamic loader
        # binds every later user to whichever copy was mapped first. If the system copy (/opt/rocm) came first, a later
        # `import torch` would run the wheel's libraries on a runtime they were not built with (observed: "No HIP GPUs are
        # available"). So when a torch wheel is installed its copy is mapped first — whether or not torch has been imported —
        # and libffp.so, torch and RCCL then share one runtime, one set of device contexts and one allocator view
        # (tests/test_host_logic.py::test_single_hip_runtime checks both load orders).
        if "torch" not in sys.modules:
            try:
                import importlib.util
                spec = importlib.util.find_spec("torch")
                cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else None
                if cand and os.path.exists(cand):
                    C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except Exception:
                pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def _check(rc: int):
    if rc != 0:
        raise FfpError(rc, lib().ffp_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int(0)
    _check(lib().ffp_device_count(C.byref(n)))
    return n.value


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_p(C.c_float))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(_p(C.c_int32))


def slice_bboxes(H: int, W: int, slice_h: int, slice_w: int, overlap_h: float = 0.2, overlap_w: float = 0.2) -> np.ndarray:
    n = C.c_int32(0)
    _check(lib().ffp_slice_bboxes(H, W, slice_h, slice_w, overlap_h, overlap_w, None, 0, C.byref(n)))
    out = np.zeros((n.value, 4), np.int32)
    _check(lib().ffp_slice_bboxes(H, W, slice_h, slice_w, overlap_h, overlap_w, _ip(out), n.value, C.byref(n)))
    return out


def letterbox_geometry(h: int, w: int, imgsz: int) -> Tuple[int, int, int, int, int, int]:
    out = np.zeros(6, np.int32)
    _check(lib().ffp_letterbox_geometry(h, w, imgsz, _ip(out)))
    return tuple(int(v) for v in out)


def _weights_bytes(weights) -> bytes:
    if isinstance(weights, (bytes, bytearray, memoryview)):
        return bytes(weights)
    if isinstance(weights, str):
        with open(weights, "rb") as f:
            return f.read()
    return weights_io.pack(weights)


def _as_frame(frame: np.ndarray) -> np.ndarray:
    f = np.ascontiguousarray(frame)
    if f.dtype != np.uint8 or f.ndim != 3 or f.shape[2] != 3:
        raise ValueError(f"expected an HxWx3 uint8 array, got {f.dtype} {f.shape}")
    return f


class Detector:
    """YOLO11{n,s}-pose on one GPU (ffp_det_*)."""

    def __init__(self, weights, arch: str = "s", nc: int = 1, nkpt: int = 5, device: int = 0, precision: int = PREC_F32):
        buf = _weights_bytes(weights)
        self._h = C.c_void_p()
        self.nc, self.nkpt, self.arch, self.device, self.precision = nc, nkpt, arch, device, precision
        cbuf = C.create_string_buffer(buf, len(buf))
        _check(lib().ffp_det_create(cbuf, len(buf), ord(arch), nc, nkpt, device, precision, C.byref(self._h)))

    @property
    def stride(self) -> int:
        return 6 + 3 * self.nkpt

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().ffp_det_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def infer_tiles(self, frame: np.ndarray, tiles: Sequence[Sequence[int]], imgsz: int, conf: float, iou: float = 0.7,
                    max_det: int = 300, chan_order: int = CHAN_AS_BGR, round_boxes: bool = False) -> List[np.ndarray]:
        """-> per tile (n_i, 6+3*nkpt) float32 rows [x1,y1,x2,y2,score,cls,kpts] in tile-local pixels (not truncated)."""
        f = _as_frame(frame)
        t = np.ascontiguousarray(np.asarray(tiles, np.int32).reshape(-1, 4))
        n = t.shape[0]
        dets = np.zeros((n, max_det, self.stride), np.float32)
        counts = np.zeros(n, np.int32)
        _check(lib().ffp_det_infer_tiles(self._h, f.ctypes.data, f.shape[0], f.shape[1], chan_order, _ip(t), n, imgsz, conf, iou,
                                         max_det, int(round_boxes), _fp(dets), _ip(counts)))
        return [dets[i, :counts[i]].copy() for i in range(n)]

    def forward_raw(self, frame: np.ndarray, tiles: Sequence[Sequence[int]], imgsz: int,
                    chan_order: int = CHAN_AS_BGR) -> List[np.ndarray]:
        """-> per tile the (4+nc+3*nkpt, A) inference-mode head output."""
        f = _as_frame(frame)
        t = np.ascontiguousarray(np.asarray(tiles, np.int32).reshape(-1, 4))
        n = t.shape[0]
        no = 4 + self.nc + 3 * self.nkpt
        cap = 0
        for x0, y0, x1, y1 in t:
            sz = imgsz if imgsz > 0 else (max(t[0][2] - t[0][0], t[0][3] - t[0][1]) + 31) // 32 * 32
            cap += no * (sz // 8) ** 2 * 2
        out = np.zeros(cap, np.float32)
        cnt = np.zeros(n, np.int32)
        _check(lib().ffp_det_forward_raw(self._h, f.ctypes.data, f.shape[0], f.shape[1], chan_order, _ip(t), n, imgsz, _fp(out), cap, _ip(cnt)))
        res, o = [], 0
        for i in range(n):
            a = int(cnt[i])
            res.append(out[o:o + no * a].reshape(no, a).copy())
            o += no * a
        return res

    def sliced_predict(self, frame: np.ndarray, slice_h: int, slice_w: int, overlap_h: float = 0.2, overlap_w: float = 0.2,
                       perform_standard_pred: bool = True, imgsz: int = 1024, conf: float = 0.3, iou: float = 0.7, max_det: int = 300,
                       pp_type: str = "GREEDYNMM", pp_metric: str = "IOS", pp_thr: float = 0.5, class_agnostic: bool = False,
                       chan_order: int = CHAN_AS_BGR, round_boxes: bool = False, cap: int = 0) -> np.ndarray:
        """Fused get_sliced_prediction -> (n, 6+3*nkpt) rows in frame coordinates (boxes int-valued)."""
        f = _as_frame(frame)
        H, W = f.shape[:2]
        if cap <= 0:
            cap = (len(slice_bboxes(H, W, slice_h, slice_w, overlap_h, overlap_w)) + 1) * max_det
        out = np.zeros((cap, self.stride), np.float32)
        n = C.c_int32(0)
        _check(lib().ffp_sliced_predict(self._h, f.ctypes.data, H, W, chan_order, slice_h, slice_w, overlap_h, overlap_w,
                                        int(perform_standard_pred), imgsz, conf, iou, max_det, int(round_boxes), PP_TYPES[pp_type],
                                        METRICS[pp_metric], float(pp_thr), int(class_agnostic), _fp(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def last_ms(self) -> dict:
        names = ["total", "preprocess", "network", "decode_nms", "merge"]
        v = C.c_float(0)
        out = {}
        for i, k in enumerate(names):
            _check(lib().ffp_det_last_ms(self._h, i, C.byref(v)))
            out[k] = v.value
        return out

    def graph_status(self) -> int:
        """1: the last call replayed a captured hipGraph, 0: not captured yet, -1: capture failed (eager launches)."""
        v = C.c_int32(0)
        _check(lib().ffp_det_graph_status(self._h, C.byref(v)))
        return v.value

    def mem_bytes(self) -> dict:
        """Device memory held: packed weights, resident plans (activations + tables), number of plans."""
        a, b, n = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        _check(lib().ffp_det_mem_bytes(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return {"weights": int(a.value), "plans": int(b.value), "plans_resident": int(n.value)}

    def drop_plans(self):
        """Release every resident plan now (rebuilt on demand); the packed weights stay."""
        _check(lib().ffp_det_drop_plans(self._h))

    def set_lanes(self, mode: int):
        """0 (default) one stream; 1 head towers and C3k side convs as parallel graph branches (detector-only deployments)."""
        _check(lib().ffp_det_set_lanes(self._h, int(mode)))

    def stream_wait_event(self, hip_event: int):
        """Order the handle's following work behind a hipEvent_t of this process (torch.cuda.Event.cuda_event)."""
        _check(lib().ffp_det_stream_wait_event(self._h, C.c_void_p(int(hip_event))))

    def set_profile(self, on: bool):
        _check(lib().ffp_det_set_profile(self._h, int(on)))

    def profile(self) -> List[dict]:
        return _profile(self._h, lib().ffp_det_profile_count, lib().ffp_det_profile_get, lib().ffp_det_profile_bytes)

    def profile_detail(self) -> List[dict]:
        return _profile_detail(self._h, lib().ffp_det_profile_detail)

    def conv_stats(self) -> dict:
        fl, ms, n = C.c_double(0), C.c_float(0), C.c_int32(0)
        _check(lib().ffp_det_last_conv_stats(self._h, C.byref(fl), C.byref(ms), C.byref(n)))
        return {"flops": fl.value, "ms": ms.value, "launches": n.value}


def _profile_detail(h, fn) -> List[dict]:
    out, i = [], 0
    while True:
        name = C.create_string_buffer(128)
        fl, ms = C.c_double(0), C.c_float(0)
        if fn(h, i, name, 128, C.byref(fl), C.byref(ms)) != 0:
            return out
        out.append({"name": name.value.decode(), "flops": fl.value, "ms": ms.value})
        i += 1


def _profile(h, count_fn, get_fn, bytes_fn=None) -> List[dict]:
    n = C.c_int32(0)
    _check(count_fn(h, C.byref(n)))
    out = []
    for i in range(n.value):
        name = C.create_string_buffer(64)
        fl, ms, ln, by = C.c_double(0), C.c_float(0), C.c_int32(0), C.c_double(0)
        _check(get_fn(h, i, name, 64, C.byref(fl), C.byref(ms), C.byref(ln)))
        if bytes_fn is not None:
            _check(bytes_fn(h, i, C.byref(by)))
        out.append({"variant": name.value.decode(), "flops": fl.value, "ms": ms.value, "launches": ln.value, "bytes": by.value})
    return out


def conv_totals_enable(on: bool = True):
    """Start (and zero) / stop the process-wide per-variant totals of every plan execution (ffp_conv_totals_*)."""
    _check(lib().ffp_conv_totals_enable(int(on)))


def conv_totals() -> dict:
    """{variant: {"launches", "flops", "bytes"}} since conv_totals_enable(True): algorithmic figures of every launch, graph replays included."""
    n = C.c_int32(0)
    _check(lib().ffp_conv_totals_count(C.byref(n)))
    out = {}
    for i in range(n.value):
        name = C.create_string_buffer(64)
        fl, by, ln = C.c_double(0), C.c_double(0), C.c_int64(0)
        _check(lib().ffp_conv_totals_get(i, name, 64, C.byref(fl), C.byref(by), C.byref(ln)))
        out[name.value.decode()] = {"launches": int(ln.value), "flops": fl.value, "bytes": by.value}
    return out


def merge(rows: np.ndarray, pp_type: str = "GREEDYNMM", metric: str = "IOS", thr: float = 0.5, class_agnostic: bool = False,
          device: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """SAHI post-process on (n, stride>=6) float32 rows -> (merged rows, source indices)."""
    r = np.ascontiguousarray(rows, np.float32)
    if r.ndim != 2 or r.shape[1] < 6:
        raise ValueError("rows must be (n, >=6)")
    n, stride = r.shape
    out = np.zeros((max(n, 1), stride), np.float32)
    src = np.zeros(max(n, 1), np.int32)
    k = C.c_int32(0)
    _check(lib().ffp_merge(device, _fp(r), n, stride, PP_TYPES[pp_type], METRICS[metric], float(thr), int(class_agnostic), _fp(out),
                           _ip(src), C.byref(k)))
    return out[:k.value].copy(), src[:k.value].copy()


class Enhancer:
    """Real-ESRGAN RRDBNet on one GPU (ffp_sr_*)."""

    def __init__(self, weights, scale: int = 4, num_block: int = 23, device: int = 0, half: bool = True):
        buf = _weights_bytes(weights)
        self._h = C.c_void_p()
        self.scale, self.num_block, self.device, self.half = scale, num_block, device, half
        cbuf = C.create_string_buffer(buf, len(buf))
        _check(lib().ffp_sr_create(cbuf, len(buf), scale, num_block, device, int(half), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().ffp_sr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enhance(self, img_bgr: np.ndarray, tile: int = 0, tile_pad: int = 10, pre_pad: int = 0) -> np.ndarray:
        return self.enhance_batch([img_bgr], tile, tile_pad, pre_pad)[0]

    def enhance_batch(self, imgs: Sequence[np.ndarray], tile: int = 0, tile_pad: int = 10, pre_pad: int = 0) -> List[np.ndarray]:
        ins = [_as_frame(i) for i in imgs]
        n = len(ins)
        outs = [np.zeros((i.shape[0] * self.scale, i.shape[1] * self.scale, 3), np.uint8) for i in ins]
        ip = (C.c_void_p * n)(*[i.ctypes.data for i in ins])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        hs = np.asarray([i.shape[0] for i in ins], np.int32)
        ws = np.asarray([i.shape[1] for i in ins], np.int32)
        _check(lib().ffp_sr_enhance_batch(self._h, n, ip, _ip(hs), _ip(ws), tile, tile_pad, pre_pad, op))
        return outs

    def enhance_crops_dev(self, d_frames: Sequence[int], H: int, W: int, boxes: np.ndarray, d_out: int, out_cap: int,
                          frame_of_box: Optional[np.ndarray] = None, tile: int = 400, tile_pad: int = 10, wait: bool = True) -> np.ndarray:
        """Crops of resident BGR frame(s) (device pointers) -> enhanced crops packed in device memory at d_out.
        One frame: ffp_sr_enhance_crops_dev (wait) / _dev_async; several: ffp_sr_enhance_crops_multi_dev_async (+ ffp_sr_wait when
        wait). Returns the n+1 byte offsets (a crop that is empty after clamping has a zero-length entry)."""
        b = np.ascontiguousarray(boxes, np.int32).reshape(-1, 4)
        n = b.shape[0]
        offs = np.zeros(n + 1, np.int64)
        op = offs.ctypes.data_as(_p(C.c_int64))
        if len(d_frames) == 1 and frame_of_box is None:
            fn = lib().ffp_sr_enhance_crops_dev if wait else lib().ffp_sr_enhance_crops_dev_async
            _check(fn(self._h, d_frames[0], H, W, _ip(b), n, tile, tile_pad, d_out, out_cap, op))
        else:
            f = np.ascontiguousarray(frame_of_box if frame_of_box is not None else np.zeros(n, np.int32), np.int32)
            ptrs = (C.c_void_p * len(d_frames))(*d_frames)
            _check(lib().ffp_sr_enhance_crops_multi_dev_async(self._h, len(d_frames), ptrs, _ip(f), H, W, _ip(b), n, tile, tile_pad, d_out,
                                                              out_cap, op))
            if wait:
                self.wait()
        return offs

    def wait(self):
        _check(lib().ffp_sr_wait(self._h))

    def last_ms(self) -> float:
        v = C.c_float(0)
        _check(lib().ffp_sr_last_ms(self._h, C.byref(v)))
        return v.value

    def set_fused_body(self, on: bool):
        """The 345 body convs as ONE persistent launch (default in fp16) or one launch per layer; bit-identical results (drops the plans)."""
        _check(lib().ffp_sr_set_fused_body(self._h, int(on)))

    def mem_bytes(self) -> dict:
        a, b, n = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        _check(lib().ffp_sr_mem_bytes(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return {"weights": int(a.value), "plans": int(b.value), "plans_resident": int(n.value)}

    def drop_plans(self):
        """Release every resident plan now (rebuilt on demand); the packed weights stay."""
        _check(lib().ffp_sr_drop_plans(self._h))

    def plan_state(self) -> dict:
        """plans_built: network layouts built so far (capacity-keyed: varying crop sizes must not grow it);
        last_graph: the last call replayed a captured hipGraph."""
        a, b = C.c_int32(0), C.c_int32(0)
        _check(lib().ffp_sr_plan_state(self._h, C.byref(a), C.byref(b)))
        return {"plans_built": a.value, "last_graph": bool(b.value)}

    def set_profile(self, on: bool):
        _check(lib().ffp_sr_set_profile(self._h, int(on)))

    def profile(self) -> List[dict]:
        return _profile(self._h, lib().ffp_sr_profile_count, lib().ffp_sr_profile_get, lib().ffp_sr_profile_bytes)

    def profile_detail(self) -> List[dict]:
        return _profile_detail(self._h, lib().ffp_sr_profile_detail)

    def conv_stats(self) -> dict:
        fl, ms, n = C.c_double(0), C.c_float(0), C.c_int32(0)
        _check(lib().ffp_sr_last_conv_stats(self._h, C.byref(fl), C.byref(ms), C.byref(n)))
        return {"flops": fl.value, "ms": ms.value, "launches": n.value}


def op_conv2d(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], stride: int = 1, groups: int = 1, act: int = 0, up: bool = False,
              res: Optional[np.ndarray] = None, res_scale: float = 1.0, precision: int = PREC_F32, device: int = 0) -> np.ndarray:
    """Single convolution through the HIP kernels. x: (n,h,w,cin) fp32 NHWC, w: (cout, cin/groups, k, k) -> (n,ho,wo,cout)."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    n, h, wd, cin = x.shape
    cout, _, k, _ = w.shape
    hi, wi = (h * 2, wd * 2) if up else (h, wd)
    ho, wo = (hi + 2 * (k // 2) - k) // stride + 1, (wi + 2 * (k // 2) - k) // stride + 1
    y = np.zeros((n, ho, wo, cout), np.float32)
    bb = np.ascontiguousarray(b, np.float32) if b is not None else None
    rr = np.ascontiguousarray(res, np.float32) if res is not None else None
    _check(lib().ffp_op_conv2d(device, precision, _fp(x), n, h, wd, cin, _fp(w), _fp(bb) if bb is not None else None, cout, k, stride,
                               groups, act, int(up), _fp(rr) if rr is not None else None, res_scale, _fp(y)))
    return y


def jpeg_encode(img: np.ndarray, quality: int = 95, bgr: bool = False, device: int = 0) -> bytes:
    """Baseline JPEG file (4:2:0, what cv2.imwrite / PIL write at this quality) of an h x w x 3 uint8 image, encoded on the GPU."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    cap = 1024 + h * w * 3
    out = np.empty(cap, np.uint8)
    n = C.c_int64(0)
    _check(lib().ffp_jpeg_encode(device, img.ctypes.data_as(_p(C.c_uint8)), h, w, int(bgr), quality, out.ctypes.data_as(_p(C.c_uint8)), cap, C.byref(n)))
    return out[:n.value].tobytes()


def jpeg_encode_dev(d_ptr: int, h: int, w: int, row_stride: int, quality: int = 95, bgr: bool = True, device: int = 0) -> bytes:
    """Same for an image already in device memory (pointer + row pitch in bytes), e.g. one enhanced crop inside the SR output buffer."""
    cap = 1024 + h * w * 3
    out = np.empty(cap, np.uint8)
    n = C.c_int64(0)
    _check(lib().ffp_jpeg_encode_dev(device, C.c_void_p(d_ptr), h, w, row_stride, int(bgr), quality, out.ctypes.data_as(_p(C.c_uint8)), cap, C.byref(n)))
    return out[:n.value].tobytes()


def jpeg_encode_batch_dev(d_base: int, offsets, hs, ws, quality: int = 95, bgr: bool = True, device: int = 0):
    """n device-resident images (d_base + offsets[i], hs[i] x ws[i] x 3, tightly packed rows) -> list of JPEG files, one pass on the GPU."""
    offsets = np.ascontiguousarray(offsets, np.int64)
    hs, ws = np.ascontiguousarray(hs, np.int32), np.ascontiguousarray(ws, np.int32)
    n = len(offsets)
    cap = int(1024 * n + 3 * (hs.astype(np.int64) * ws).sum())
    out = np.empty(max(cap, 1), np.uint8)
    oo = np.zeros(n + 1, np.int64)
    _check(lib().ffp_jpeg_encode_batch_dev(device, C.c_void_p(d_base), n, offsets.ctypes.data_as(_p(C.c_int64)), hs.ctypes.data_as(_p(C.c_int32)),
                                           ws.ctypes.data_as(_p(C.c_int32)), None, int(bgr), quality, out.ctypes.data_as(_p(C.c_uint8)), cap, oo.ctypes.data_as(_p(C.c_int64))))
    return [out[oo[i]:oo[i + 1]].tobytes() for i in range(n)]


def jpeg_info(data: bytes):
    """(height, width, components) from the headers of a JPEG stream."""
    buf = np.frombuffer(data, np.uint8)
    h, w, nc = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _check(lib().ffp_jpeg_info(buf.ctypes.data_as(_p(C.c_uint8)), len(buf), C.byref(h), C.byref(w), C.byref(nc)))
    return h.value, w.value, nc.value


def jpeg_decode(data: bytes, bgr: bool = False, device: int = 0) -> np.ndarray:
    """JPEG stream -> h x w x 3 uint8 (host Huffman decoding, device IDCT / upsampling / colour conversion), what cv2.imread returns (bgr=True)."""
    h, w, _ = jpeg_info(data)
    buf = np.frombuffer(data, np.uint8)
    out = np.empty((h, w, 3), np.uint8)
    _check(lib().ffp_jpeg_decode(device, buf.ctypes.data_as(_p(C.c_uint8)), len(buf), int(bgr), out.ctypes.data_as(_p(C.c_uint8)), out.size))
    return out


def jpeg_decode_dev(data: bytes, d_ptr: int, row_stride: int, cap: int, bgr: bool = True, device: int = 0):
    """Same into device memory (pointer, row pitch and capacity in bytes); returns (h, w)."""
    h, w, _ = jpeg_info(data)
    buf = np.frombuffer(data, np.uint8)
    _check(lib().ffp_jpeg_decode_dev(device, buf.ctypes.data_as(_p(C.c_uint8)), len(buf), int(bgr), C.c_void_p(d_ptr), row_stride, cap))
    return h, w


def jpeg_decode_stats():
    """(files Huffman-decoded on the device, files that fell back to the host decoder, extra synchronisation rounds) since load"""
    a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    _check(lib().ffp_jpeg_decode_stats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def _ragged(rows, width, dtype):
    """list of [n_i][width] arrays -> (contiguous [sum n][width], int64 offsets [len + 1])"""
    rows = [np.asarray(r, dtype).reshape(-1, width) for r in rows]
    off = np.zeros(len(rows) + 1, np.int64)
    np.cumsum([len(r) for r in rows], out=off[1:])
    flat = np.ascontiguousarray(np.concatenate(rows + [np.zeros((0, width), dtype)], 0))
    return flat, off


def eval_wider_pr(preds, gts, evaluate, iou_thr: float = 0.5, thresh_num: int = 1000, device: int = 0) -> np.ndarray:
    """Official WIDER FACE protocol on the device (ffp_eval_wider_pr): per-image lists preds [N_i][5] (x, y, w, h, score), gts [G_i][4],
    evaluate [G_i] (1 = evaluated face). -> int64 [thresh_num][2] = {valid proposals, matched faces} per score threshold."""
    p, po = _ragged(preds, 5, np.float64)
    g, go = _ragged(gts, 4, np.float64)
    e, eo = _ragged([np.asarray(x, np.uint8).reshape(-1, 1) for x in evaluate], 1, np.uint8)
    if len(po) != len(go) or not np.array_equal(go, eo):
        raise ValueError("preds / gts / evaluate must describe the same images and faces")
    out = np.zeros((thresh_num, 2), np.int64)
    _check(lib().ffp_eval_wider_pr(device, p.ctypes.data_as(_p(C.c_double)), po.ctypes.data_as(_p(C.c_int64)), g.ctypes.data_as(_p(C.c_double)),
                                   go.ctypes.data_as(_p(C.c_int64)), e.ctypes.data_as(_p(C.c_uint8)), len(po) - 1, iou_thr, thresh_num,
                                   out.ctypes.data_as(_p(C.c_int64))))
    return out


def eval_dual_match(preds, faces, valid, iou_thr: float = 0.5, device: int = 0):
    """Dual-protocol matching on the device (ffp_eval_dual_match). -> list of int32 flag arrays per image (1 TP, 0 FP, 2 not counted)."""
    p, po = _ragged(preds, 5, np.float64)
    f, fo = _ragged(faces, 4, np.float64)
    v, vo = _ragged([np.asarray(x, np.uint8).reshape(-1, 1) for x in valid], 1, np.uint8)
    if len(po) != len(fo) or not np.array_equal(fo, vo):
        raise ValueError("preds / faces / valid must describe the same images and faces")
    flags = np.zeros(max(int(po[-1]), 1), np.int32)
    _check(lib().ffp_eval_dual_match(device, p.ctypes.data_as(_p(C.c_double)), po.ctypes.data_as(_p(C.c_int64)), f.ctypes.data_as(_p(C.c_double)),
                                     fo.ctypes.data_as(_p(C.c_int64)), v.ctypes.data_as(_p(C.c_uint8)), len(po) - 1, iou_thr, flags.ctypes.data_as(_p(C.c_int32))))
    return [flags[po[i]:po[i + 1]] for i in range(len(po) - 1)]


def op_conv1x1_up2(coarse: np.ndarray, fine: np.ndarray, w: np.ndarray, b: np.ndarray, act: int = 0, precision: int = PREC_F32X3, device: int = 0) -> np.ndarray:
    """1x1 conv over the virtual concat [nearest_x2(coarse) | fine]; coarse (n,h/2,w/2,c_up), fine (n,h,w,c_fine), w (cout, c_up+c_fine)."""
    coarse = np.ascontiguousarray(coarse, np.float32)
    fine = np.ascontiguousarray(fine, np.float32)
    w = np.ascontiguousarray(w, np.float32).reshape(w.shape[0], -1)
    b = np.ascontiguousarray(b, np.float32)
    n, h, wd, cf = fine.shape
    y = np.zeros((n, h, wd, w.shape[0]), np.float32)
    _check(lib().ffp_op_conv1x1_up2(device, precision, _fp(coarse), _fp(fine), n, h, wd, coarse.shape[3], cf, _fp(w), _fp(b), w.shape[0], act, _fp(y)))
    return y


def op_conv2d_shape(shape: int = -1) -> None:
    """Pin the workgroup shape of the following op_conv2d calls (tests of one kernel variant); -1 restores the automatic choice."""
    _check(lib().ffp_op_conv2d_shape(int(shape)))


def op_conv2d_time(n, h, w, cin, cout, k=3, stride=1, up=False, precision=PREC_F16, iters=50, dbg=0, shape=-1, device=0) -> float:
    """Tuning hook: mean microseconds per launch of one dense conv on synthetic data (see ffp_op_conv2d_time)."""
    us = C.c_float(0)
    _check(lib().ffp_op_conv2d_time(device, precision, n, h, w, cin, cout, k, stride, int(up), iters, dbg, shape, C.byref(us)))
    return us.value
