"""ctypes binding of libs2sr.so (include/s2sr.h).

This is the only place Python touches the native library.  There is NO fallback: if the
shared object is missing or no gfx950 device is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from pathlib import Path
from typing import List, Optional

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("S2SR_LIB", _HERE.parent / "csrc" / "libs2sr.so"))

PREC_F16, PREC_F16_HP, PREC_FP8 = 0, 1, 2
_ERR = {-1: "invalid argument", -2: "HIP error", -3: "weights not loaded", -4: "bad weight blob",
        -5: "no gfx950 device (no CPU fallback exists)", -6: "buffer too small", -7: "file could not be written"}


class S2srError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("num_block", C.c_int32), ("num_feat", C.c_int32), ("num_grow", C.c_int32), ("scale", C.c_int32),
                ("precision", C.c_int32), ("device", C.c_int32), ("group", C.c_int32), ("reserved", C.c_int32)]


class Window(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("y1", "y2", "x1", "x2", "crop_top", "crop_bottom", "crop_left",
                                         "crop_right", "oy1", "oy2", "ox1", "ox2")]


class PPParams(C.Structure):
    _fields_ = [("clahe_clip", C.c_float), ("clahe_grid", C.c_int32), ("blur_sigma", C.c_float),
                ("w_img", C.c_float), ("w_blur", C.c_float), ("hue_lo", C.c_int32), ("hue_hi", C.c_int32),
                ("sat_gain", C.c_float), ("stages", C.c_int32)]


class DebugConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("precision", "group", "trunk_w4", "lo_exp", "fp8_form", "fp8_x_exp", "fp8_g_exp",
                                         "fp8_hp_tail", "graphs_on", "trunk_wino")] + [("reserved", C.c_int32 * 6)]


class DebugTrunkArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "form", "N", "Cin", "H", "W")] + \
               [(n, C.c_void_p) for n in ("x", "weight", "bias", "lo", "skip", "y", "y_aux")]


class KStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


# constants of the two post-process variants (reference wow_sr.py:191-205, farm_sr.py:100,170-178)
def pp_wow() -> PPParams:
    return PPParams(2.5, 8, 1.2, 1.4, -0.4, 35, 85, 1.2, 7)


def pp_farm() -> PPParams:
    return PPParams(2.5, 8, 1.5, 2.2, -1.2, 35, 85, 1.3, 7)


PP_ORDER_BGR, PP_ORDER_SWAP_OUT = 1, 2      # s2sr_pp_band_begin_dev `order` bits


_lib = None

_PROTOS = {
    "s2sr_version": (C.c_char_p, []),
    "s2sr_device_count": (C.c_int, []),
    "s2sr_create": (C.c_int, [C.POINTER(_Config), C.POINTER(C.c_void_p)]),
    "s2sr_destroy": (None, [C.c_void_p]),
    "s2sr_last_error": (C.c_char_p, [C.c_void_p]),
    "s2sr_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "s2sr_host_free": (C.c_int, [C.c_void_p]),
    "s2sr_load_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "s2sr_expected_blob_floats": (C.c_size_t, [C.c_int32]),
    "s2sr_load_weights_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "s2sr_calibrate_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32)]),
    "s2sr_plan_tiles": (C.c_int, [C.c_int32] * 5 + [C.POINTER(Window), C.c_int32, C.POINTER(C.c_int32)]),
    "s2sr_forward_batch_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "s2sr_forward_batch_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_void_p]),
    "s2sr_forward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "s2sr_enhance_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "s2sr_enhance_job_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(PPParams), C.c_void_p]),
    "s2sr_enhance_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "s2sr_tile_process_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "s2sr_cut_windows_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p]),
    "s2sr_stitch_windows_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p]),
    "s2sr_stitch_rows_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p]),
    "s2sr_forward_part_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p]),
    "s2sr_copy_to_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "s2sr_postprocess_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(PPParams), C.c_void_p]),
    "s2sr_postprocess_batch_u8_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                C.POINTER(PPParams), C.c_void_p, C.c_void_p]),
    "s2sr_pp_band_begin_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(PPParams), C.c_int32, C.c_void_p]),
    "s2sr_pp_band_hist_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "s2sr_pp_band_lut_dev": (C.c_int, [C.c_void_p, C.c_void_p]),
    "s2sr_pp_band_rows_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "s2sr_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "s2sr_get_kernel_stats": (C.c_int, [C.c_void_p, C.POINTER(KStat), C.c_int32, C.POINTER(C.c_int32)]),
    "s2sr_reset_kernel_stats": (C.c_int, [C.c_void_p]),
    "s2sr_synchronize": (C.c_int, [C.c_void_p]),
    "s2sr_debug_f32_to_e4m3": (C.c_uint8, [C.c_float]),
    "s2sr_debug_pack_f8_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "s2sr_debug_pack_f8": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "s2sr_warp_bilinear_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_void_p]),
    "s2sr_tiles_base_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_void_p]),
    "s2sr_tiles_overview_u8": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]),
    "s2sr_tiles_write_png": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_int32)]),
    "s2sr_tiles_write_png_xyz": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                           C.POINTER(C.c_int32)]),
    "s2sr_tiff_lzw_encode": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "s2sr_tiff_lzw_decode": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "s2sr_png_bound": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "s2sr_png_encode": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "s2sr_png_idat_band": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_size_t, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]),
    "s2sr_png_write_tiles": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_size_t, C.POINTER(C.c_char_p), C.c_int32,
                                       C.POINTER(C.c_int32)]),
    "s2sr_graph_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "s2sr_debug_conv": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p] +
                        [C.c_int32] * 3 + [C.c_void_p]),
    "s2sr_debug_pick_mosaic": (C.c_int, [C.c_int32] * 3 + [C.POINTER(C.c_int32)] * 2),
    "s2sr_debug_mosaic_patches": (C.c_int, [C.c_int32] * 3 + [C.POINTER(C.c_int64)] * 2),
    "s2sr_debug_plan_chunks": (C.c_int, [C.c_int32] * 6 + [C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]),
    "s2sr_debug_get_config": (C.c_int, [C.c_void_p, C.POINTER(DebugConfig)]),
    "s2sr_debug_conv_trunk": (C.c_int, [C.c_void_p, C.POINTER(DebugTrunkArgs)]),
    "s2sr_debug_mfma_ceiling": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_float)]),
    "s2sr_debug_bench_conv": (C.c_int, [C.c_void_p] + [C.c_int32] * 6 + [C.POINTER(C.c_float), C.c_void_p, C.c_int32]),
    "s2sr_debug_rdb_persistent": (C.c_int, [C.c_void_p] + [C.c_int32] * 5 + [C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32)]),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)


def _mapped_hip_runtimes() -> set:
    """Real paths of every libamdhip64 mapped into this process (Linux; empty when /proc is not readable)."""
    out = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    out.add(os.path.realpath(line.split()[-1]))
    except OSError:
        pass
    return out


def load_library():
    """dlopen libs2sr.so and attach prototypes.  Raises if it is not built: no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise S2srError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"or `make -C {LIB_PATH.parent}`.  There is no CPU fallback.")
    # ONE HIP runtime per process.  libs2sr.so is linked against `libamdhip64.so.<N>` (RUNPATH /opt/rocm/lib) and the
    # PyTorch-ROCm wheel bundles a runtime with the same SONAME.  The dynamic loader resolves a DT_NEEDED entry against
    # already loaded objects by SONAME first, so when torch is imported BEFORE this dlopen, libs2sr binds to torch's
    # bundled runtime and both sides share one HIP context, one set of streams and device pointers (which is what the
    # `*_dev` entry points and the RCCL path need).  The other order maps the system runtime first and leaves torch with
    # "No HIP GPUs are available".  So: import torch first when it exists, then verify that exactly one libamdhip64 is
    # mapped -- two would mean the SONAMEs diverged (e.g. a ROCm major bump on one side) and device pointers could not
    # be shared; that is an error, not something to limp along with.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(LIB_PATH))
    hips = _mapped_hip_runtimes()
    if len(hips) > 1:
        raise S2srError(f"two HIP runtimes are mapped into this process ({sorted(hips)}): libs2sr.so and PyTorch must share one "
                        "libamdhip64 (same SONAME) -- rebuild libs2sr.so against the ROCm major version of the torch wheel")
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError here == ABI drift, let it surface
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def experimental() -> bool:
    """Is the loaded library the experimental build (make EXP=1 -> libs2sr_exp.so: the kernel forms the measurements buried
    and every stamped build)?  The shipped library answers requests for those with an error."""
    return b"+experimental" in load_library().s2sr_version()


def tiff_lzw_encode(data) -> bytes:
    """Host call (no GPU): raw strip bytes -> TIFF LZW.  ctypes drops the GIL: strips encode in parallel."""
    lib = load_library()
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8).ravel()
    cap = buf.size * 3 // 2 + 16
    out = np.empty(cap, np.uint8)
    n = C.c_size_t(0)
    rc = lib.s2sr_tiff_lzw_encode(buf.ctypes.data_as(C.c_void_p), buf.size, out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    if rc:
        raise S2srError(f"s2sr_tiff_lzw_encode: {_ERR.get(rc, rc)}")
    return out[:n.value].tobytes()


def _png_rows(img: np.ndarray):
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] not in (3, 4) or img.shape[0] < 1 or img.shape[1] < 1:
        raise ValueError(f"expected HxWx3 or HxWx4 uint8, got {img.shape} {img.dtype}")
    if img.strides[2] != 1 or img.strides[1] != img.shape[2] or img.strides[0] < img.shape[1] * img.shape[2]:
        img = np.ascontiguousarray(img)
    return img, img.shape[0], img.shape[1], img.shape[2], img.strides[0]


_PNG_TLS = threading.local()


def _png_out(cap: int) -> np.ndarray:
    """Output staging of the calling thread (a fresh 300-KB array per tile is an mmap / munmap pair per call)."""
    buf = getattr(_PNG_TLS, "buf", None)
    if buf is None or buf.size < cap or buf.size > max(4 * cap, 1 << 22):
        buf = _PNG_TLS.buf = np.empty(cap, np.uint8)
    return buf


def png_encode(img: np.ndarray) -> bytes:
    """Host call (no GPU): HxWx3 / HxWx4 uint8 -> a complete PNG file (Sub filter, run-length deflate, dynamic Huffman).
    ctypes drops the GIL: tiles encode in parallel on threads."""
    lib = load_library()
    img, h, w, c, stride = _png_rows(img)
    cap = lib.s2sr_png_bound(w, h, c)
    out = _png_out(cap)
    n = C.c_size_t(0)
    rc = lib.s2sr_png_encode(img.ctypes.data_as(C.c_void_p), w, h, c, stride, out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    if rc:
        raise S2srError(f"s2sr_png_encode: {_ERR.get(rc, rc)}")
    return out[:n.value].tobytes()


def png_idat_band(rows: np.ndarray, first: bool, last: bool):
    """Host call (no GPU): a band of rows of a big image -> (one complete IDAT chunk, Adler-32 of the band's filtered bytes,
    their count); see s2sr_png_idat_band in include/s2sr.h for how the bands make a file."""
    lib = load_library()
    rows, h, w, c, stride = _png_rows(rows)
    cap = lib.s2sr_png_bound(w, h, c)
    out = _png_out(cap)
    n, adler, raw_n = C.c_size_t(0), C.c_uint32(0), C.c_size_t(0)
    rc = lib.s2sr_png_idat_band(rows.ctypes.data_as(C.c_void_p), w, h, c, stride, int(first), int(last), out.ctypes.data_as(C.c_void_p),
                                cap, C.byref(n), C.byref(adler), C.byref(raw_n))
    if rc:
        raise S2srError(f"s2sr_png_idat_band: {_ERR.get(rc, rc)}")
    return out[:n.value].tobytes(), int(adler.value), int(raw_n.value)


def png_write_tiles(tiles_arr: np.ndarray, paths, skip_transparent: bool = True) -> np.ndarray:
    """Host call (no GPU): [n, S, S, 3|4] uint8 tiles -> PNG files at `paths` (str / Path, None = skip), parent directories made on
    demand, fully transparent RGBA tiles skipped.  Returns the 0/1 array of files written.  The whole loop runs without the GIL."""
    lib = load_library()
    if tiles_arr.dtype != np.uint8 or tiles_arr.ndim != 4 or tiles_arr.shape[1] != tiles_arr.shape[2] or tiles_arr.shape[3] not in (3, 4):
        raise ValueError(f"expected [n, S, S, 3|4] uint8, got {tiles_arr.shape} {tiles_arr.dtype}")
    n, size, _, c = tiles_arr.shape
    if len(paths) != n:
        raise ValueError(f"{n} tiles, {len(paths)} paths")
    if tiles_arr.strides[1:] != (size * c, c, 1) or tiles_arr.strides[0] < size * size * c:
        tiles_arr = np.ascontiguousarray(tiles_arr)
    cp = (C.c_char_p * n)(*[None if p is None else os.fsencode(p) for p in paths])
    written = np.zeros(n, np.int32)
    rc = lib.s2sr_png_write_tiles(tiles_arr.ctypes.data_as(C.c_void_p), n, size, c, tiles_arr.strides[0] if n else size * size * c, cp,
                                  int(skip_transparent), written.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc:
        raise S2srError(f"s2sr_png_write_tiles: {_ERR.get(rc, rc)}" + (f" ({os.strerror(C.get_errno())})" if rc == -7 and C.get_errno() else ""))
    return written


def tiff_lzw_decode(data: bytes, expected: int) -> bytes:
    """Host call (no GPU): TIFF LZW strip/tile -> raw bytes (at most `expected`)."""
    lib = load_library()
    out = C.create_string_buffer(max(int(expected), 1))
    n = C.c_size_t(0)
    rc = lib.s2sr_tiff_lzw_decode(data, len(data), out, int(expected), C.byref(n))
    if rc:
        raise S2srError(f"s2sr_tiff_lzw_decode: {_ERR.get(rc, rc)}")
    return out.raw[:n.value]


def tiff_lzw_decode_into(src: np.ndarray, offset: int, count: int, dst: np.ndarray) -> int:
    """Host call (no GPU): the LZW stream src[offset : offset + count] (a uint8 array over the file) decoded straight into the
    C-contiguous array `dst` (at most dst.nbytes) -> bytes produced.  No intermediate buffers: a strip of a chunky GeoTIFF lands in
    the rows of the image it belongs to."""
    lib = load_library()
    if not dst.flags["C_CONTIGUOUS"] or not dst.flags["WRITEABLE"] or offset < 0 or count < 0 or offset + count > src.size:
        raise ValueError("tiff_lzw_decode_into: a C-contiguous writable destination and a stream inside the source are required")
    n = C.c_size_t(0)
    rc = lib.s2sr_tiff_lzw_decode(C.c_void_p(src.ctypes.data + offset), count, dst.ctypes.data_as(C.c_void_p), dst.nbytes, C.byref(n))
    if rc:
        raise S2srError(f"s2sr_tiff_lzw_decode: {_ERR.get(rc, rc)}")
    return int(n.value)


def plan_tiles(H: int, W: int, tile: int = 256, pad: int = 10, scale: int = 4) -> List[Window]:
    """Pure host call (works without a GPU): window plan of `_tile_process`."""
    lib = load_library()
    n = C.c_int32(0)
    rc = lib.s2sr_plan_tiles(H, W, tile, pad, scale, None, 0, C.byref(n))
    if rc:
        raise S2srError(f"s2sr_plan_tiles: {_ERR.get(rc, rc)}")
    arr = (Window * n.value)()
    rc = lib.s2sr_plan_tiles(H, W, tile, pad, scale, arr, n.value, C.byref(n))
    if rc:
        raise S2srError(f"s2sr_plan_tiles: {_ERR.get(rc, rc)}")
    return list(arr)


def pick_mosaic(B: int, th: int, tw: int) -> tuple:
    """(kx, ky): windows per launch image the engine would use for B windows of th x tw (host arithmetic)."""
    kx, ky = C.c_int32(0), C.c_int32(0)
    rc = load_library().s2sr_debug_pick_mosaic(B, th, tw, C.byref(kx), C.byref(ky))
    if rc:
        raise S2srError(f"s2sr_debug_pick_mosaic failed ({_ERR.get(rc, rc)})")
    return int(kx.value), int(ky.value)


def mosaic_patches(B: int, th: int, tw: int) -> tuple:
    """(launched, plain): 32x32 patches the engine launches for B windows of th x tw, and what B plain images would cost."""
    a, b = C.c_int64(0), C.c_int64(0)
    rc = load_library().s2sr_debug_mosaic_patches(B, th, tw, C.byref(a), C.byref(b))
    if rc:
        raise S2srError(f"s2sr_debug_mosaic_patches failed ({_ERR.get(rc, rc)})")
    return int(a.value), int(b.value)


def plan_chunks(units: int, u_max: int, unit_windows: int, per: int, pimg: int, ncu: int = 256) -> List[int]:
    """Chunk sizes (row units, front to back) s2sr_enhance_u8 would use; host arithmetic, works without a GPU."""
    lib = load_library()
    n = C.c_int32(0)
    buf = (C.c_int32 * max(1, units))()
    rc = lib.s2sr_debug_plan_chunks(units, u_max, unit_windows, per, pimg, ncu, buf, max(1, units), C.byref(n))
    if rc:
        raise S2srError(f"s2sr_debug_plan_chunks failed ({_ERR.get(rc, rc)})")
    return [int(buf[i]) for i in range(n.value)]


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class _PinnedPool:
    """Page-locked output arrays (s2sr_host_alloc).  `RealESRGAN.enhance` returns a fresh numpy array per call
    (cnn_super_resolution.py:231-233); a fresh pageable array costs the device-to-host path a staging copy and a page fault per
    4 KB (805 MB: 130 ms), a page-locked one is filled by the DMA engines (20 ms).  Pinning is slow (hundreds of ms for 800 MB), so
    buffers are recycled: an array handed out here returns its buffer to the pool when the last view of it is garbage collected.
    Buffers come in 16-MB size classes, so images of nearby sizes share them (r03 ADVICE: exact-size keys pinned a new buffer
    per distinct image size).  Two caps: S2SR_PINNED_POOL_MB (default 4096) of IDLE buffers are kept, and S2SR_PINNED_MAX_MB
    (default 16384) bounds all page-locked bytes, in use and idle -- beyond it `empty` hands out ordinary pageable arrays (the
    library then takes its staged route).  A service that RETAINS results (job tables, caches) should copy them (`.copy()`) or
    set S2SR_PINNED_OUT=0: a retained result keeps its page-locked buffer."""
    BUCKET = 16 << 20

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._free = {}          # bucket bytes -> [address, ...]
        self._idle = 0
        self._total = 0          # page-locked bytes alive, in use + idle
        self._cap = int(os.environ.get("S2SR_PINNED_POOL_MB", "4096")) << 20
        self._max = int(os.environ.get("S2SR_PINNED_MAX_MB", "16384")) << 20
        self.on = os.environ.get("S2SR_PINNED_OUT", "1") != "0"
        self.hits = self.misses = self.refused = 0

    def empty(self, shape, dtype) -> np.ndarray:
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        if not self.on or n < (8 << 20):
            return np.empty(shape, dtype=dtype)
        nb = -(-n // self.BUCKET) * self.BUCKET
        lib = load_library()
        with self._lock:
            lst = self._free.get(nb)
            addr = lst.pop() if lst else None
            if addr is not None:
                self._idle -= nb
                self.hits += 1
            elif self._total + nb > self._max:
                self.refused += 1
                return np.empty(shape, dtype=dtype)
            else:
                self._total += nb           # reserved before the (slow) allocation, outside the lock
                self.misses += 1
        if addr is None:
            p = C.c_void_p()
            if lib.s2sr_host_alloc(nb, C.byref(p)) or not p.value:
                with self._lock:
                    self._total -= nb
                return np.empty(shape, dtype=dtype)      # no page-locked memory to be had: the pageable route still works
            addr = p.value
        buf = (C.c_uint8 * nb).from_address(addr)
        weakref.finalize(buf, self._give, addr, nb)       # runs when the last numpy view of `buf` is gone
        return np.frombuffer(buf, dtype=np.uint8, count=n).view(dtype).reshape(shape)

    def _give(self, addr, nb):
        with self._lock:
            if self._idle + nb <= self._cap:
                self._free.setdefault(nb, []).append(addr)
                self._idle += nb
                return
            self._total -= nb
        try:
            load_library().s2sr_host_free(C.c_void_p(addr))
        except Exception:
            pass

    def trim(self):
        """Release every idle buffer."""
        with self._lock:
            addrs = [(a, nb) for nb, lst in self._free.items() for a in lst]
            self._free.clear()
            self._idle = 0
            self._total -= sum(nb for _, nb in addrs)
        for a, _ in addrs:
            load_library().s2sr_host_free(C.c_void_p(a))


pinned_pool = _PinnedPool()


class Engine:
    """One native handle == one GPU.  Thread-safe (the library serialises calls per handle)."""

    def __init__(self, num_block: int = 23, precision: int = PREC_F16, device: int = 0, group: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.num_block = num_block
        self.precision, self.group = precision, group
        cfg = _Config(num_block, 64, 32, 4, precision, device, group, 0)
        rc = self._lib.s2sr_create(C.byref(cfg), C.byref(self._h))
        if rc:
            msg = self._lib.s2sr_last_error(None)
            raise S2srError(f"s2sr_create failed ({_ERR.get(rc, rc)}): {msg.decode() if msg else ''}")

    def group_images(self) -> int:
        """Images per launch group (engine.hip group_size: 16, 32 for the fp8 trunk, or what the constructor was given)."""
        return self.group if self.group > 0 else (32 if self.precision == PREC_FP8 else 16)

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc:
            msg = self._lib.s2sr_last_error(self._h)
            raise S2srError(f"{what} failed ({_ERR.get(rc, rc)}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.s2sr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights ----------------------------------------------------------------------------
    def load_blob(self, blob: np.ndarray):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self._check(self._lib.s2sr_load_weights(self._h, _ptr(blob), blob.size), "s2sr_load_weights")

    def load_blob_dev(self, d_blob: int, n_floats: int, stream: int = 0):
        """Device pointer to the fp32 blob (e.g. the tensor an RCCL broadcast filled)."""
        self._check(self._lib.s2sr_load_weights_dev(self._h, C.c_void_p(d_blob), n_floats, C.c_void_p(stream)),
                    "s2sr_load_weights_dev")

    def load_state_dict(self, sd):
        from .weights import flatten_state_dict
        self.load_blob(flatten_state_dict(sd, self.num_block))

    def calibrate_fp8(self, tiles: np.ndarray, headroom: float = 2.0) -> tuple:
        """PREC_FP8 engines: set the trunk's activation scales from representative tiles -> (x_exp, g_exp)."""
        tiles = np.ascontiguousarray(tiles, dtype=np.uint8)
        B, h, w, c = tiles.shape
        assert c == 3
        xe, ge = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.s2sr_calibrate_fp8(self._h, _ptr(tiles), B, h, w, headroom, C.byref(xe), C.byref(ge)), "s2sr_calibrate_fp8")
        return int(xe.value), int(ge.value)

    # -- forward ----------------------------------------------------------------------------
    def forward_batch_u8(self, tiles: np.ndarray) -> np.ndarray:
        tiles = np.ascontiguousarray(tiles, dtype=np.uint8)
        B, h, w, c = tiles.shape
        assert c == 3
        out = pinned_pool.empty((B, 4 * h, 4 * w, 3), np.uint8)
        self._check(self._lib.s2sr_forward_batch_u8(self._h, _ptr(tiles), B, h, w, _ptr(out)), "s2sr_forward_batch_u8")
        return out

    def forward_batch_u8_dev(self, d_in: int, B: int, h: int, w: int, d_out: int, stream: int = 0):
        """Device pointers (ints, e.g. torch.Tensor.data_ptr()); asynchronous on `stream`."""
        self._check(self._lib.s2sr_forward_batch_u8_dev(self._h, d_in, B, h, w, d_out, C.c_void_p(stream)),
                    "s2sr_forward_batch_u8_dev")

    def forward_f32(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        N, c, H, W = x.shape
        assert c == 3
        y = np.empty((N, 3, 4 * H, 4 * W), dtype=np.float32)
        self._check(self._lib.s2sr_forward_f32(self._h, _ptr(x), N, H, W, _ptr(y)), "s2sr_forward_f32")
        return y

    def enhance_u8(self, img: np.ndarray, tile: int = 256, pad: int = 10) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        H, W, c = img.shape
        assert c == 3
        out = pinned_pool.empty((4 * H, 4 * W, 3), np.uint8)
        self._check(self._lib.s2sr_enhance_u8(self._h, _ptr(img), H, W, tile, pad, _ptr(out)), "s2sr_enhance_u8")
        return out

    def enhance_job_u8(self, rgb: np.ndarray, prm: Optional[PPParams] = None, tile: int = 256, pad: int = 10) -> np.ndarray:
        """RGB in -> (BGR) net -> RGB -> post-process `prm` (None: none) -> RGB out: a job's device work in one call."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        H, W, c = rgb.shape
        assert c == 3
        out = pinned_pool.empty((4 * H, 4 * W, 3), np.uint8)
        self._check(self._lib.s2sr_enhance_job_u8(self._h, _ptr(rgb), H, W, tile, pad, C.byref(prm) if prm is not None else None, _ptr(out)),
                    "s2sr_enhance_job_u8")
        return out

    def enhance_f32(self, img: np.ndarray, tile: int = 256, pad: int = 10) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        H, W, c = img.shape
        out = np.empty((4 * H, 4 * W, 3), dtype=np.float32)
        self._check(self._lib.s2sr_enhance_f32(self._h, _ptr(img), H, W, tile, pad, _ptr(out)), "s2sr_enhance_f32")
        return out

    def tile_process_f32(self, img: np.ndarray, tile: int = 256, pad: int = 10) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        H, W, c = img.shape
        out = np.empty((4 * H, 4 * W, 3), dtype=np.float32)
        self._check(self._lib.s2sr_tile_process_f32(self._h, _ptr(img), H, W, tile, pad, _ptr(out)),
                    "s2sr_tile_process_f32")
        return out

    # -- multi-GPU building blocks (device pointers) ------------------------------------------
    def cut_windows_u8_dev(self, d_img: int, H: int, W: int, tile: int, pad: int, first: int, count: int,
                           d_tiles: int, stream: int = 0):
        self._check(self._lib.s2sr_cut_windows_u8_dev(self._h, d_img, H, W, tile, pad, first, count, d_tiles,
                                                      C.c_void_p(stream)), "s2sr_cut_windows_u8_dev")

    def stitch_windows_u8_dev(self, d_tiles: int, H: int, W: int, tile: int, pad: int, d_out: int, stream: int = 0):
        self._check(self._lib.s2sr_stitch_windows_u8_dev(self._h, d_tiles, H, W, tile, pad, d_out, C.c_void_p(stream)),
                    "s2sr_stitch_windows_u8_dev")

    def stitch_rows_u8_dev(self, d_tiles: int, H: int, W: int, tile: int, pad: int, oy0: int, oy1: int, d_out: int, stream: int = 0):
        """Output rows [oy0, oy1) of the paste; d_out is the whole [4H,4W,3] image."""
        self._check(self._lib.s2sr_stitch_rows_u8_dev(self._h, d_tiles, H, W, tile, pad, oy0, oy1, d_out, C.c_void_p(stream)),
                    "s2sr_stitch_rows_u8_dev")

    def forward_part_u8_dev(self, d_in: int, B: int, h: int, w: int, job_windows: int, d_out: int, stream: int = 0):
        """B windows that are a part of a job of `job_windows` (one mosaic plan and workspace for the whole job)."""
        self._check(self._lib.s2sr_forward_part_u8_dev(self._h, d_in, B, h, w, job_windows, d_out, C.c_void_p(stream)),
                    "s2sr_forward_part_u8_dev")

    def copy_to_host(self, dst: np.ndarray, d_src: int, stream: int = 0):
        """Device -> the (C-contiguous) host array `dst`, behind everything on `stream`; blocks until the bytes are there."""
        assert dst.flags["C_CONTIGUOUS"]
        self._check(self._lib.s2sr_copy_to_host(self._h, _ptr(dst), C.c_void_p(d_src), dst.nbytes, C.c_void_p(stream)), "s2sr_copy_to_host")

    # -- post-process -----------------------------------------------------------------------
    def postprocess_u8(self, rgb: np.ndarray, prm: PPParams) -> np.ndarray:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        H, W, c = rgb.shape
        assert c == 3
        out = np.empty_like(rgb)
        self._check(self._lib.s2sr_postprocess_u8(self._h, _ptr(rgb), H, W, C.byref(prm), _ptr(out)),
                    "s2sr_postprocess_u8")
        return out

    def postprocess_batch_u8_dev(self, d_in: int, B: int, H: int, W: int, prm: PPParams, d_out: int, stream: int = 0):
        self._check(self._lib.s2sr_postprocess_batch_u8_dev(self._h, d_in, B, H, W, C.byref(prm), d_out,
                                                            C.c_void_p(stream)), "s2sr_postprocess_batch_u8_dev")

    # the same over ONE device image in row bands (include/s2sr.h: begin, hist over every row, lut, rows from row 0 on)
    def pp_band_begin_dev(self, H: int, W: int, prm: PPParams, order: int = 0, stream: int = 0):
        self._check(self._lib.s2sr_pp_band_begin_dev(self._h, H, W, C.byref(prm), order, C.c_void_p(stream)), "s2sr_pp_band_begin_dev")

    def pp_band_hist_dev(self, d_img: int, y0: int, y1: int, stream: int = 0):
        self._check(self._lib.s2sr_pp_band_hist_dev(self._h, C.c_void_p(d_img), y0, y1, C.c_void_p(stream)), "s2sr_pp_band_hist_dev")

    def pp_band_lut_dev(self, stream: int = 0):
        self._check(self._lib.s2sr_pp_band_lut_dev(self._h, C.c_void_p(stream)), "s2sr_pp_band_lut_dev")

    def pp_band_rows_dev(self, d_img: int, y0: int, y1: int, d_out: int, stream: int = 0):
        self._check(self._lib.s2sr_pp_band_rows_dev(self._h, C.c_void_p(d_img), y0, y1, C.c_void_p(d_out), C.c_void_p(stream)),
                    "s2sr_pp_band_rows_dev")

    def mfma_ceiling(self, mode: int, stages: int, launches: int) -> dict:
        """csrc/ceiling.hip: mode 0 bare MFMA loop, 1 + LDS operand reads, 2 + LDS-DMA ring refill (3 half the bytes, 4 from cache, 5 conv1-4's
        traffic mix, 6 conv5's, 7 / 8 from a 100-MB / 200-MB source inside the Infinity Cache) -> TFLOP/s over `launches` launches."""
        fl, by, ms = C.c_double(0), C.c_double(0), C.c_float(0)
        self._check(self._lib.s2sr_debug_mfma_ceiling(self._h, mode, stages, launches, C.byref(fl), C.byref(by), C.byref(ms)),
                    "s2sr_debug_mfma_ceiling")
        return {"ms": float(ms.value), "launches": launches, "us_per_launch": float(ms.value) * 1e3 / launches,
                "TFLOP_per_s": fl.value * launches / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0,
                "dma_GB_per_s": by.value * launches / (ms.value * 1e-3) / 1e9 if ms.value > 0 else 0.0}

    def rdb_persistent(self, variant: int, grid: int, P: int, rdbs: int, launches: int) -> dict:
        """csrc/persist.hip (diagnostic prototype): the RDB-shaped loop whose workgroups stay across layers -> TFLOP/s, dependency-wait timeouts.
        variant: bit 0 device-scope plane loads + written-through stores, bit 1 the deeper ring (32-KiB stages, three stages of look-ahead);
        4 / 5: variant 0 / 1 with every handed-over piece checked (halo_mismatches, own_mismatches)."""
        fl, ms, to, mm = C.c_double(0), C.c_float(0), C.c_int32(0), (C.c_int32 * 2)(0, 0)
        self._check(self._lib.s2sr_debug_rdb_persistent(self._h, int(variant), grid, P, rdbs, launches, C.byref(fl), C.byref(ms), C.byref(to), mm),
                    "s2sr_debug_rdb_persistent")
        return {"ms": float(ms.value), "launches": launches, "timeouts": int(to.value), "halo_mismatches": int(mm[0]), "own_mismatches": int(mm[1]), "working_set_MB": grid * P * (0.46875 if int(variant) & 2 else 0.5),
                "TFLOP_per_s": fl.value * launches / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0}

    # -- measurement ------------------------------------------------------------------------
    def set_profiling(self, every: int):
        """0/False: off; N>=1: HIP-event pair around every N-th launch of each kernel family."""
        self._check(self._lib.s2sr_set_profiling(self._h, int(every)), "s2sr_set_profiling")

    def reset_kernel_stats(self):
        self._check(self._lib.s2sr_reset_kernel_stats(self._h), "s2sr_reset_kernel_stats")

    def kernel_stats(self) -> dict:
        n = C.c_int32(0)
        self._check(self._lib.s2sr_get_kernel_stats(self._h, None, 0, C.byref(n)), "s2sr_get_kernel_stats")
        arr = (KStat * n.value)()
        self._check(self._lib.s2sr_get_kernel_stats(self._h, arr, n.value, C.byref(n)), "s2sr_get_kernel_stats")
        return {s.name.decode(): {"launches": s.launches, "total_ms": s.total_ms, "flops": s.flops, "bytes": s.bytes}
                for s in arr}

    # -- XYZ tile pyramid -----------------------------------------------------------------
    def warp_bilinear_u8(self, rgb: np.ndarray, grid: np.ndarray, step: int, out_h: int, out_w: int) -> np.ndarray:
        rgb = np.ascontiguousarray(rgb, np.uint8)
        grid = np.ascontiguousarray(grid, np.float32)
        out = pinned_pool.empty((out_h, out_w, 4), np.uint8)        # (page-locked when large: the raster comes back with one DMA)
        self._check(self._lib.s2sr_warp_bilinear_u8(self._h, _ptr(rgb), rgb.shape[0], rgb.shape[1], _ptr(grid), grid.shape[0],
                                                    grid.shape[1], step, out_h, out_w, _ptr(out)), "s2sr_warp_bilinear_u8")
        return out

    def tiles_base_u8(self, rgba, col_lo, col_hi, row_lo, row_hi, fetch: bool = True, on_device: bool = False) -> Optional[np.ndarray]:
        """fetch=False: the level is computed and left on the device (for tiles_write_png / the next overview); returns None.
        on_device: the raster is the one the previous call on this engine -- warp_bilinear_u8 -- produced and nothing else ran on the
        engine since: its device copy is used instead of an upload; `rgba` is then only read for its shape (array or (H, W))."""
        if on_device:
            H, W = rgba.shape[:2] if hasattr(rgba, "shape") else rgba
            src = None
        else:
            rgba = np.ascontiguousarray(rgba, np.uint8)
            H, W = rgba.shape[:2]
            src = _ptr(rgba)
        t = [np.ascontiguousarray(a, np.int32) for a in (col_lo, col_hi, row_lo, row_hi)]
        nx, ny = t[0].size // 256, t[2].size // 256
        out = np.empty((ny, nx, 256, 256, 4), np.uint8) if fetch else None
        self._check(self._lib.s2sr_tiles_base_u8(self._h, src, H, W, _ptr(t[0]), _ptr(t[1]), _ptr(t[2]),
                                                 _ptr(t[3]), nx, ny, _ptr(out) if fetch else None), "s2sr_tiles_base_u8")
        return out

    def tiles_overview_u8(self, child, ox: int, oy: int, pnx: int, pny: int, on_device: bool = False,
                          fetch: bool = True) -> Optional[np.ndarray]:
        """on_device: the children are the level the previous tiles_base_u8 / tiles_overview_u8 call on this engine produced and
        nothing else ran on the engine since: its device copy is used instead of an upload; `child` is then only read for its
        shape (the array, or a (cny, cnx) pair).  fetch=False: the new level stays on the device, returns None."""
        out = np.empty((pny, pnx, 256, 256, 4), np.uint8) if fetch else None
        if on_device:
            cny, cnx = child.shape[:2] if hasattr(child, "shape") else child
            src = None
        else:
            child = np.ascontiguousarray(child, np.uint8)
            cny, cnx = child.shape[:2]
            src = _ptr(child)
        self._check(self._lib.s2sr_tiles_overview_u8(self._h, src, cnx, cny, ox, oy, pnx, pny, _ptr(out) if fetch else None),
                    "s2sr_tiles_overview_u8")
        return out

    def tiles_write_png(self, nx: int, ny: int, paths, skip_transparent: bool = True, host_encoder: bool = False,
                        row_threads: bool = False, small_groups: bool = False) -> np.ndarray:
        """The PNG files of the level the previous tiles call left on the device, encoded there (s2sr_tiles_write_png): `paths` has
        ny * nx entries in the tile array's order (None = skip).  Returns the 0/1 array [ny, nx] of files written."""
        if len(paths) != nx * ny:
            raise ValueError(f"{nx * ny} tiles, {len(paths)} paths")
        cp = (C.c_char_p * (nx * ny))(*[None if p is None else os.fsencode(p) for p in paths])
        written = np.zeros(nx * ny, np.int32)
        self._check(self._lib.s2sr_tiles_write_png(self._h, nx, ny, cp, int(skip_transparent) | (2 if host_encoder else 0) | (4 if row_threads else 0)
                                                   | (8 if small_groups else 0), written.ctypes.data_as(C.POINTER(C.c_int32))),
                    "s2sr_tiles_write_png")
        return written.reshape(ny, nx)

    def tiles_write_png_xyz(self, nx: int, ny: int, directory, zoom: int, x0: int, y_rows, skip_transparent: bool = True) -> np.ndarray:
        """tiles_write_png for the XYZ layout: tile (j, i) -> <directory>/<zoom>/<x0 + i>/<y_rows[j]>.png; the paths are built natively."""
        rows = np.ascontiguousarray(y_rows, np.int32)
        if rows.size != ny:
            raise ValueError(f"{ny} tile rows, {rows.size} row numbers")
        written = np.zeros(nx * ny, np.int32)
        self._check(self._lib.s2sr_tiles_write_png_xyz(self._h, nx, ny, os.fsencode(str(directory)), zoom, x0, _ptr(rows),
                                                       1 if skip_transparent else 0, written.ctypes.data_as(C.POINTER(C.c_int32))),
                    "s2sr_tiles_write_png_xyz")
        return written.reshape(ny, nx)

    def synchronize(self):
        self._check(self._lib.s2sr_synchronize(self._h), "s2sr_synchronize")

    def graph_stats(self) -> tuple[int, int]:
        """(groups captured into hipGraphs, graph replays) since the engine was created."""
        c, r = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.s2sr_graph_stats(self._h, C.byref(c), C.byref(r)), "s2sr_graph_stats")
        return int(c.value), int(r.value)

    # -- test hook --------------------------------------------------------------------------
    def debug_conv(self, x: np.ndarray, weight: np.ndarray, bias: np.ndarray, upsample: bool = False,
                   act: bool = False) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        weight = np.ascontiguousarray(weight, dtype=np.float32)
        bias = np.ascontiguousarray(bias, dtype=np.float32)
        N, cin, H, W = x.shape
        cout = weight.shape[0]
        assert weight.shape == (cout, cin, 3, 3) and bias.shape == (cout,)
        s = 2 if upsample else 1
        y = np.empty((N, cout, s * H, s * W), dtype=np.float32)
        self._check(self._lib.s2sr_debug_conv(self._h, _ptr(x), N, cin, H, W, _ptr(weight), _ptr(bias), cout,
                                              int(upsample), int(act), _ptr(y)), "s2sr_debug_conv")
        return y


TRUNK_F16_CONV14, TRUNK_F16_CONV5, TRUNK_F16_CONV5_RRDB, TRUNK_F8_CONV14, TRUNK_F8_CONV5, TRUNK_F8_CONV5_RRDB = range(6)


def _debug_config(self) -> dict:
    """What s2sr_create read from the environment for this handle (kernel forms, scales)."""
    c = DebugConfig()
    self._check(self._lib.s2sr_debug_get_config(self._h, C.byref(c)), "s2sr_debug_get_config")
    d = {n: int(getattr(c, n)) for n, _ in DebugConfig._fields_ if n != "reserved"}
    d["mosaic_on"] = int(c.reserved[0])
    d["f16_loader"] = int(c.reserved[1])
    d["last_fold"] = int(c.reserved[2])
    d["tail_w4"] = int(c.reserved[3])
    d["f16_full"] = int(c.reserved[4])
    d["ws_allocs"] = int(c.reserved[5])
    return d


def _debug_conv_trunk(self, kind, x, weight, bias, lo=None, skip=None, form=0):
    """One RDB-shaped conv through conv_trunk_f16 / conv_trunk_f8 (include/s2sr.h: s2sr_debug_conv_trunk).
    Returns y, or (y, y_aux) for the fp8 conv5 kinds."""
    x = np.ascontiguousarray(x, np.float32)
    weight = np.ascontiguousarray(weight, np.float32)
    bias = np.ascontiguousarray(bias, np.float32)
    N, cin, H, W = x.shape
    cout = weight.shape[0]
    assert weight.shape == (cout, cin, 3, 3) and bias.shape == (cout,)
    y = np.empty((N, cout, H, W), np.float32)
    aux = np.empty((N, 64, H, W), np.float32) if kind in (TRUNK_F8_CONV5, TRUNK_F8_CONV5_RRDB) else None
    lo = None if lo is None else np.ascontiguousarray(lo, np.float32)
    skip = None if skip is None else np.ascontiguousarray(skip, np.float32)
    p = lambda a: None if a is None else a.ctypes.data
    args = DebugTrunkArgs(kind, form, N, cin, H, W, p(x), p(weight), p(bias), p(lo), p(skip), p(y), p(aux))
    self._check(self._lib.s2sr_debug_conv_trunk(self._h, C.byref(args)), "s2sr_debug_conv_trunk")
    return y if aux is None else (y, aux)


Engine.debug_config = _debug_config
Engine.debug_conv_trunk = _debug_conv_trunk


def _bench_conv(self, N, H, W, cin, cout, iters=20, trace_wgs=0):
    """Diagnostic: (avg launch us, trace[wgs,24] of s_memtime ticks or None)."""
    us = C.c_float(0)
    tr = np.zeros((max(trace_wgs, 1), 24), dtype=np.uint64)
    self._check(self._lib.s2sr_debug_bench_conv(self._h, N, H, W, cin, cout, iters, C.byref(us),
                                                _ptr(tr) if trace_wgs else None, trace_wgs), "s2sr_debug_bench_conv")
    return float(us.value), (tr if trace_wgs else None)


Engine.bench_conv = _bench_conv


def device_count() -> int:
    return int(load_library().s2sr_device_count())
