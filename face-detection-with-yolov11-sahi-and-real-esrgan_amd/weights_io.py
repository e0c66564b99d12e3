"""FFPW — the flat named-tensor weight container the native engine loads.

Layout (little endian):
    char[4]  magic  = b"FFPW"
    u32      version = 1
    u32      n_tensors
    u64      data_offset            (absolute file offset of the blob, 64-byte aligned)
    n_tensors x { u16 name_len; char name[name_len]; u8 dtype (0 = f32); u8 ndim;
                  u32 dims[ndim]; u64 offset (relative to data_offset, 64-byte aligned); u64 nbytes }
    blob

Weights are stored fp32 in PyTorch OIHW order with BatchNorm already folded into (weight, bias); the
engine repacks them into its MFMA fragment layout at load (csrc/weights.cpp). The reference loads
pickled Ultralytics / basicsr checkpoints instead (utils/yolo_wrapper.py:55, utils/enhancer.py:156);
`from_esrgan_state_dict` / `from_ultralytics_state_dict` are the converters for those.
"""
from __future__ import annotations

import struct
from typing import Dict, Mapping

import numpy as np

MAGIC = b"FFPW"
VERSION = 1


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def pack(tensors: Mapping[str, np.ndarray]) -> bytes:
    """Serialise {name: float32 ndarray} to FFPW bytes."""
    entries = []
    blob_parts = []
    off = 0
    for name, arr in tensors.items():
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.float32))
        nb = a.nbytes
        entries.append((name.encode("utf-8"), a.shape, off, nb))
        blob_parts.append(a.tobytes())
        pad = _align(nb) - nb
        if pad:
            blob_parts.append(b"\0" * pad)
        off += _align(nb)
    table = bytearray()
    for nm, shape, o, nb in entries:
        table += struct.pack("<H", len(nm)) + nm
        table += struct.pack("<BB", 0, len(shape))
        table += struct.pack(f"<{len(shape)}I", *shape) if shape else b""
        table += struct.pack("<QQ", o, nb)
    header_len = 4 + 4 + 4 + 8 + len(table)
    data_offset = _align(header_len)
    out = bytearray()
    out += MAGIC + struct.pack("<IIQ", VERSION, len(entries), data_offset)
    out += table
    out += b"\0" * (data_offset - len(out))
    out += b"".join(blob_parts)
    return bytes(out)


def unpack(buf: bytes) -> Dict[str, np.ndarray]:
    """Parse FFPW bytes back into {name: float32 ndarray} (copying)."""
    if buf[:4] != MAGIC:
        raise ValueError("not an FFPW container")
    version, n, data_offset = struct.unpack_from("<IIQ", buf, 4)
    if version != VERSION:
        raise ValueError(f"unsupported FFPW version {version}")
    p = 20
    out: Dict[str, np.ndarray] = {}
    for _ in range(n):
        (ln,) = struct.unpack_from("<H", buf, p); p += 2
        name = buf[p:p + ln].decode("utf-8"); p += ln
        dtype, ndim = struct.unpack_from("<BB", buf, p); p += 2
        dims = struct.unpack_from(f"<{ndim}I", buf, p) if ndim else (); p += 4 * ndim
        off, nb = struct.unpack_from("<QQ", buf, p); p += 16
        if dtype != 0:
            raise ValueError("only f32 tensors are defined in FFPW v1")
        out[name] = np.frombuffer(buf, dtype=np.float32, count=nb // 4, offset=data_offset + off).reshape(dims).copy()
    return out


def save(path: str, tensors: Mapping[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(pack(tensors))


def load(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        return unpack(f.read())


# ----------------------------------------------------------------------------------------------
# checkpoint converters
# ----------------------------------------------------------------------------------------------
def from_esrgan_state_dict(sd: Mapping[str, "np.ndarray"]) -> Dict[str, np.ndarray]:
    """basicsr RRDBNet state dict ('params_ema' | 'params' already selected) -> FFPW tensors.
    Key names are kept verbatim."""
    out = {}
    for k, v in sd.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        out[k] = a.astype(np.float32)
    return out


def load_esrgan_pth(path: str) -> Dict[str, np.ndarray]:
    """Read a Real-ESRGAN .pth (plain tensor dict) the way RealESRGANer does: prefer 'params_ema'."""
    import torch
    ck = torch.load(path, map_location="cpu", weights_only=True)
    if "params_ema" in ck:
        ck = ck["params_ema"]
    elif "params" in ck:
        ck = ck["params"]
    return from_esrgan_state_dict(ck)


def from_ultralytics_state_dict(sd: Mapping[str, "np.ndarray"], bn_eps: float = 1e-3) -> Dict[str, np.ndarray]:
    """Unfused Ultralytics state dict (`model.N...conv.weight` + `...bn.{weight,bias,running_mean,
    running_var}`) -> fused FFPW tensors (`...conv.weight`, `...conv.bias`). Plain Conv2d heads pass through.
    To be run where the checkpoint can be unpickled (needs `ultralytics`; not available offline)."""
    get = lambda k: (sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else np.asarray(sd[k])).astype(np.float64)
    out: Dict[str, np.ndarray] = {}
    for k in sd:
        if k.endswith(".conv.weight") and (k[: -len(".conv.weight")] + ".bn.weight") in sd:
            p = k[: -len(".conv.weight")]
            w = get(k)
            g, b = get(p + ".bn.weight"), get(p + ".bn.bias")
            m, v = get(p + ".bn.running_mean"), get(p + ".bn.running_var")
            s = g / np.sqrt(v + bn_eps)
            out[p + ".conv.weight"] = (w * s[:, None, None, None]).astype(np.float32)
            out[p + ".conv.bias"] = (b - m * s).astype(np.float32)
        elif (k.endswith(".weight") or k.endswith(".bias")) and ".bn." not in k and ".dfl." not in k:
            if k.endswith(".weight") and (k[:-7] + ".conv.weight") in sd:
                continue
            out[k] = get(k).astype(np.float32)
    return out
