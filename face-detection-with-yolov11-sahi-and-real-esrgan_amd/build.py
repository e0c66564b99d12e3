"""Build libffp.so (hand-written HIP for gfx950 + the C-ABI) in-tree with hipcc. No torch, no cmake.

    python -m <package>.build            or      from <package> import build; build.build()

hipcc cross-compiles without a GPU; objects are cached under csrc/build/ by source mtime.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libffp.so")
SOURCES = ["common.cpp", "weights.cpp", "engine.cpp", "conv_mfma.hip", "conv_rows.hip", "conv_rows16.hip", "conv_rows16pc.hip", "conv_trunk.hip", "conv_pw.hip", "conv_k3d.hip", "ops_misc.hip", "det_post.hip", "merge.hip", "eval.hip", "jpeg.hip", "jpeg_huff.hip", "jpeg_dec.cpp",
           "sr_ops.hip", "yolo11.cpp", "rrdb.cpp", "api.cpp"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result",
         "-ffp-contract=off"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libffp.so)")


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(HERE, "..", "include", "ffp.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool) -> str:
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
    spath = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), _newest_header()):
        return obj
    cmd = [_hipcc(), *FLAGS, *os.environ.get("FFP_EXTRA_FLAGS", "").split(), "-c", spath, "-o", obj]
    if src.endswith(".cpp"):
        cmd.insert(1, "-x")
        cmd.insert(2, "hip")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {OUT} ({os.path.getsize(OUT) / 1e6:.1f} MB)")
    return OUT


# diagnostic variants: the named sources recompiled with extra defines, linked with the shipped objects into csrc/build/libffp_<name>.so.
# Load one with FFP_LIB=<path> (_lib.py); libffp.so itself is never touched.
VARIANTS = {
    "pw_stamp": (["conv_pw.hip"], ["-DFFP_PW_STAMP=1"]),                    # s_memtime stamps per weight stage (profiles/r03_pw_stage_stamps.txt)
    "r16_stamp": (["conv_rows16.hip"], ["-DFFP_R16_STAMP=1"]),              # stamps per chunk phase (profiles/r03_rows16_in_kernel_stamps.txt)
    "r16_dbg": (["conv_rows16.hip"], ["-DFFP_R16_DBG=1"]),                  # compile-time phase-skip instantiations (tools/rows16_phase_probe.py)
    "r16_stash1": (["conv_rows16.hip"], ["-DFFP_R16_STASH=1"]),             # staging placement experiments (tools/rows16_stash_probe.sh)
    "r16_stash2": (["conv_rows16.hip"], ["-DFFP_R16_STASH=2"]),
    "r16_st16": (["conv_rows16.hip"], ["-DFFP_R16_STORE_AUX=16"]),          # output stores write-through (sc1) / sc0 sc1 / nt: is the kernel-end L2 write-back what a launch boundary costs?
    "r16_st17": (["conv_rows16.hip"], ["-DFFP_R16_STORE_AUX=17"]),
    "r16_st2": (["conv_rows16.hip"], ["-DFFP_R16_STORE_AUX=2"]),
    "k3d_dbg": (["conv_k3d.hip"], ["-DFFP_K3D_DBG=1"]),
    "trunk_dbg": (["conv_trunk.hip"], ["-DFFP_TRUNK_DBG=1"]),               # s_memtime stamps per phase of the fused-body kernel (tools/trunk_stamp_probe.py)
    "trunk_dbg_w5": (["conv_trunk.hip"], ["-DFFP_TRUNK_DBG=1", "-DFFP_TRUNK_STAMP_WAVE=5"]),      # stamps of a pixel loader (wave 5) instead of the control wave
    "trunk_dbg_w11": (["conv_trunk.hip"], ["-DFFP_TRUNK_DBG=1", "-DFFP_TRUNK_STAMP_WAVE=11"]),
    "trunk_dbg_skip18": (["conv_trunk.hip"], ["-DFFP_TRUNK_DBG=1", "-DFFP_TRUNK_SKIP=18"]),   # stamps without MFMAs and fragment reads: is the end-of-step wait the DMA itself?
    "trunk_dbg_skip4": (["conv_trunk.hip"], ["-DFFP_TRUNK_DBG=1", "-DFFP_TRUNK_SKIP=4"]),     # stamps without DMA
    "trunk_dbg_skip16": (["conv_trunk.hip"], ["-DFFP_TRUNK_DBG=1", "-DFFP_TRUNK_SKIP=16"]),   # stamps without fragment reads
    "trunk_skip1": (["conv_trunk.hip"], ["-DFFP_TRUNK_SKIP=1"]),            # no epilogue (sums kept alive)
    "trunk_skip5": (["conv_trunk.hip"], ["-DFFP_TRUNK_SKIP=5"]),            # no epilogue, no DMA
    "trunk_skip2": (["conv_trunk.hip"], ["-DFFP_TRUNK_SKIP=2"]),            # compile-time phase skips (tools/trunk_phase_probe.py): no MFMA
    "trunk_skip4": (["conv_trunk.hip"], ["-DFFP_TRUNK_SKIP=4"]),            # no DMA
    "trunk_skip16": (["conv_trunk.hip"], ["-DFFP_TRUNK_SKIP=16"]),          # no fragment reads
    "trunk_skip22": (["conv_trunk.hip"], ["-DFFP_TRUNK_SKIP=22"]),          # none of the three
    "trunk_skip23": (["conv_trunk.hip"], ["-DFFP_TRUNK_SKIP=23"]),          # bare skeleton: control, item set-up, barriers
}


def build_variant(name: str, verbose: bool = True) -> str:
    srcs, defs = VARIANTS[name]
    build(verbose=False)                                  # the shipped objects (and libffp.so) first
    bdir = os.path.join(CSRC, "build")
    objs = []
    for s_ in SOURCES:
        base = os.path.splitext(s_)[0]
        if s_ in srcs:
            obj = os.path.join(bdir, f"{base}.{name}.o")
            cmd = [_hipcc(), *FLAGS, *defs, "-c", os.path.join(CSRC, s_), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {s_} ({name}):\n{r.stdout}\n{r.stderr}")
            objs.append(obj)
        else:
            objs.append(os.path.join(bdir, base + ".o"))
    out = os.path.join(bdir, f"libffp_{name}.so")
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {out}")
    return out


ASAN_SOURCES = ["common.cpp", "weights.cpp", "jpeg_dec.cpp", "asan_host_check.cpp"]
ASAN_OUT = os.path.join(CSRC, "build", "asan_host_check")


def build_asan(verbose: bool = True) -> str:
    """Host-only AddressSanitizer + UBSan build of the parsers of untrusted bytes (FFPW containers, JPEG markers and Huffman
    tables) with their robustness driver (csrc/asan_host_check.cpp). No device code; runs on a CPU-only box."""
    clang = os.path.join(os.path.dirname(os.path.realpath(_hipcc())), "..", "lib", "llvm", "bin", "clang++")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang++"
    os.makedirs(os.path.dirname(ASAN_OUT), exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in ASAN_SOURCES]
    if os.path.exists(ASAN_OUT) and os.path.getmtime(ASAN_OUT) > max(max(os.path.getmtime(s) for s in srcs), _newest_header()):
        return ASAN_OUT
    cmd = [clang, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", *srcs, "-o", ASAN_OUT, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"sanitizer build failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {ASAN_OUT}")
    return ASAN_OUT


if __name__ == "__main__":
    if "--asan" in sys.argv:
        build_asan()
    elif "--variant" in sys.argv:
        build_variant(sys.argv[sys.argv.index("--variant") + 1])
    else:
        build(force="--force" in sys.argv)
