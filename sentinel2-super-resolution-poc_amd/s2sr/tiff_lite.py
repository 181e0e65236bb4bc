"""Baseline-TIFF / BigTIFF reader for the rasters the path is fed with: multi-band 8/16/32-bit
GeoTIFFs as Sentinel-2 deliveries come (the reference reads them with rasterio,
server/app/wow_sr.py:59-79; PIL cannot decode more than one 16-bit band).

Supports: little/big endian, classic TIFF and BigTIFF, strips or tiles, PlanarConfiguration 1
(chunky) and 2 (separate planes), compression none / LZW (5) / Deflate (8, 32946) / PackBits
(32773), Predictor 1 and 2 (horizontal differencing), uint8/uint16/uint32/int16/int32/float32/
float64 samples of 8/16/32/64 bits.  Returns the first image of the file as [H, W, bands] plus
the raw tag dictionary (GeoTIFF tags included).  LZW strips are decoded by the native library
(host code, s2sr_tiff_lzw_decode); everything else is numpy + zlib.

Uploaded files reach this reader (app/sr_routes.py /api/enhance), so a malformed file is a TiffError whatever it breaks
(offsets past the end, zero or absurd dimensions, missing tags, corrupt streams), a Deflate chunk is inflated to the size its
geometry says and no further, and an image above S2SR_TIFF_MAX_BYTES decoded bytes (default 4 GiB) is refused before anything
is allocated.  tests/test_tiff_cpu.py mutates valid files at random to hold that line.
"""
from __future__ import annotations

import os
import struct
import zlib
from pathlib import Path
from typing import Dict, Tuple

import numpy as np

# tag ids
IMAGE_WIDTH, IMAGE_LENGTH, BITS_PER_SAMPLE, COMPRESSION, PHOTOMETRIC = 256, 257, 258, 259, 262
STRIP_OFFSETS, SAMPLES_PER_PIXEL, ROWS_PER_STRIP, STRIP_BYTE_COUNTS = 273, 277, 278, 279
PLANAR_CONFIG, PREDICTOR = 284, 317
TILE_WIDTH, TILE_LENGTH, TILE_OFFSETS, TILE_BYTE_COUNTS = 322, 323, 324, 325
SAMPLE_FORMAT = 339

_TYPE = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 7: ("B", 1), 8: ("h", 2),
         9: ("i", 4), 10: ("ii", 8), 11: ("f", 4), 12: ("d", 8), 16: ("Q", 8), 17: ("q", 8), 18: ("Q", 8)}


class TiffError(ValueError):
    pass


def _read_ifd(buf: memoryview, bo: str, big: bool, off: int) -> Dict[int, tuple]:
    tags: Dict[int, tuple] = {}
    if big:
        (n,) = struct.unpack_from(bo + "Q", buf, off)
        off += 8
        esz, cfmt, inl = 20, "Q", 8
    else:
        (n,) = struct.unpack_from(bo + "H", buf, off)
        off += 2
        esz, cfmt, inl = 12, "I", 4
    for i in range(n):
        e = off + i * esz
        tag, typ = struct.unpack_from(bo + "HH", buf, e)
        (cnt,) = struct.unpack_from(bo + cfmt, buf, e + 4)
        if typ not in _TYPE:
            continue
        fmt, sz = _TYPE[typ]
        nbytes = sz * cnt
        voff = e + 4 + (8 if big else 4)
        if nbytes > inl:
            (voff,) = struct.unpack_from(bo + cfmt, buf, voff)
        if voff + nbytes > len(buf):
            raise TiffError(f"tag {tag}: value runs past the end of the file")
        if typ == 2:
            tags[tag] = (bytes(buf[voff:voff + nbytes]).split(b"\0")[0].decode("latin-1"),)
        elif typ in (5, 10):
            v = struct.unpack_from(bo + fmt[0] * (2 * cnt), buf, voff)
            tags[tag] = tuple(v[2 * k] / v[2 * k + 1] if v[2 * k + 1] else 0.0 for k in range(cnt))
        else:
            tags[tag] = struct.unpack_from(bo + fmt * cnt, buf, voff)
    return tags


def _lzw(data: bytes, expected: int) -> bytes:
    from . import native
    return native.tiff_lzw_decode(data, expected)


def _packbits(data: bytes, expected: int) -> bytes:
    out = bytearray()
    i, n = 0, len(data)
    while i < n and len(out) < expected:
        h = data[i]
        i += 1
        if h < 128:
            out += data[i:i + h + 1]
            i += h + 1
        elif h > 128:
            out += bytes([data[i]]) * (257 - h)
            i += 1
    return bytes(out)


def _decompress(comp: int, data: bytes, expected: int) -> bytes:
    if comp == 1:
        return data
    if comp in (8, 32946):
        return zlib.decompressobj().decompress(data, expected)      # a chunk never needs more than its geometry: no inflate bombs
    if comp == 5:
        return _lzw(data, expected)
    if comp == 32773:
        return _packbits(data, expected)
    raise TiffError(f"TIFF compression {comp} is not supported (none, LZW, Deflate, PackBits are)")


def _dtype(bits: int, fmt: int, bo: str) -> np.dtype:
    kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
    if kind is None or bits not in (8, 16, 32, 64) or (kind == "f" and bits < 32):
        raise TiffError(f"unsupported sample layout: {bits} bits, SampleFormat {fmt}")
    return np.dtype(f"{'<' if bo == '<' else '>'}{kind}{bits // 8}")


def max_decoded_bytes() -> int:
    from .hostpool import env_int
    return env_int("S2SR_TIFF_MAX_BYTES", 1 << 32)      # (a malformed value is ignored, not turned into a TiffError of the file)


def read_tiff(path) -> Tuple[np.ndarray, Dict[int, tuple]]:
    """-> (array [H, W, bands] in native byte order, {tag: values}).  TiffError for anything it cannot or will not decode."""
    try:
        return _read_tiff(path)
    except TiffError:
        raise
    except (struct.error, KeyError, IndexError, ZeroDivisionError, OverflowError, zlib.error, MemoryError, ValueError,
            RuntimeError) as e:      # RuntimeError: the native LZW decoder's S2srError on a corrupt stream
        raise TiffError(f"{path}: malformed TIFF ({type(e).__name__}: {e})") from e


def _read_file(path):
    """The file's bytes.  Large files (an SR output is 40 - 60 MB) are read in 2-MB slices from the host pool into one buffer: the
    copy out of the page cache is what `read_bytes` spends its 8 - 9 ms in, on one thread."""
    size = os.stat(path).st_size
    if size < (8 << 20) or not hasattr(os, "preadv"):
        return Path(path).read_bytes()
    from . import hostpool
    buf = np.empty(size, np.uint8)            # (not a bytearray: that would be zero-filled, on this thread, first)
    mv = memoryview(buf)
    step = 2 << 20
    fd = os.open(path, os.O_RDONLY)
    try:
        def get(a):
            b = min(size, a + step)
            while a < b:
                n = os.preadv(fd, [mv[a:b]], a)
                if n <= 0:
                    raise TiffError(f"{path}: the file shrank while it was read")
                a += n
        list(hostpool.pool().map(get, range(0, size, step)))
    finally:
        os.close(fd)
    return mv


def _read_tiff(path) -> Tuple[np.ndarray, Dict[int, tuple]]:
    raw = _read_file(path)
    buf = memoryview(raw)
    if len(raw) < 8 or raw[:2] not in (b"II", b"MM"):
        raise TiffError(f"{path}: not a TIFF file")
    bo = "<" if raw[:2] == b"II" else ">"
    (magic,) = struct.unpack_from(bo + "H", buf, 2)
    if magic == 42:
        big = False
        (ifd_off,) = struct.unpack_from(bo + "I", buf, 4)
    elif magic == 43:
        big = True
        (ifd_off,) = struct.unpack_from(bo + "Q", buf, 8)
    else:
        raise TiffError(f"{path}: bad TIFF magic {magic}")
    t = _read_ifd(buf, bo, big, ifd_off)
    try:
        W, H = int(t[IMAGE_WIDTH][0]), int(t[IMAGE_LENGTH][0])
    except KeyError as e:
        raise TiffError(f"{path}: ImageWidth / ImageLength missing") from e
    spp = int(t.get(SAMPLES_PER_PIXEL, (1,))[0])
    bits = t.get(BITS_PER_SAMPLE, (1,))
    if len(set(bits)) != 1:
        raise TiffError(f"{path}: bands of different bit depth {bits}")
    fmts = t.get(SAMPLE_FORMAT, (1,))
    dt = _dtype(int(bits[0]), int(fmts[0]), bo)
    comp = int(t.get(COMPRESSION, (1,))[0])
    planar = int(t.get(PLANAR_CONFIG, (1,))[0])
    pred = int(t.get(PREDICTOR, (1,))[0])
    if pred not in (1, 2):
        raise TiffError(f"{path}: Predictor {pred} is not supported")
    tiled = TILE_OFFSETS in t
    if tiled:
        cw, ch = int(t[TILE_WIDTH][0]), int(t[TILE_LENGTH][0])
        offs, cnts = t[TILE_OFFSETS], t[TILE_BYTE_COUNTS]
    else:
        cw, ch = W, int(t.get(ROWS_PER_STRIP, (H,))[0])
        ch = min(ch, H) if ch > 0 else H
        offs, cnts = t[STRIP_OFFSETS], t[STRIP_BYTE_COUNTS]
    if min(W, H, spp, cw, ch) <= 0 or planar not in (1, 2):
        raise TiffError(f"{path}: {W}x{H}x{spp}, chunks {cw}x{ch}, PlanarConfiguration {planar}")
    nx, ny = (W + cw - 1) // cw, (H + ch - 1) // ch
    cap = max_decoded_bytes()
    if H * W * spp * dt.itemsize > cap or nx * ny * cw * ch * spp * dt.itemsize > 4 * cap:
        raise TiffError(f"{path}: {W}x{H}x{spp} {dt} in {cw}x{ch} chunks is above S2SR_TIFF_MAX_BYTES = {cap}")
    planes = spp if planar == 2 else 1
    cspp = 1 if planar == 2 else spp          # samples per pixel inside one chunk
    if len(offs) < nx * ny * planes or len(cnts) < nx * ny * planes:
        raise TiffError(f"{path}: {len(offs)} chunk offsets and {len(cnts)} byte counts listed, {nx * ny * planes} needed")
    # LZW strips of a chunky image in the machine's byte order (what this path writes and what rasterio writes for it) decode
    # straight into their rows of the result: no copy of the compressed bytes, no decode buffer, no copy into place, and -- the
    # strips covering every row -- no zero fill of the result first (a 4096 x 4096 RGB read: 24 -> ~12 ms on 16 threads)
    direct = comp == 5 and not tiled and pred == 1 and planar == 1 and (dt.itemsize == 1 or dt.byteorder in ("=", "|") or
                                                                        (dt.byteorder == "<") == (struct.pack("=H", 1)[0] == 1))
    out = (np.empty if direct else np.zeros)((H, W, spp), dtype=dt.newbyteorder("="))
    raw_u8 = np.frombuffer(raw, np.uint8) if direct else None

    def chunk(k: int) -> None:          # chunks are independent: decoded on a thread pool (zlib and the
        pl, rem = divmod(k, nx * ny)    # native LZW decoder both run without the GIL)
        iy, ix = divmod(rem, nx)
        o, c = int(offs[k]), int(cnts[k])
        if o + c > len(raw):
            raise TiffError(f"{path}: chunk {k} ({c} bytes at {o}) runs past the end of the file")
        rows = ch if tiled else min(ch, H - iy * ch)    # strips are not padded, tiles are
        expected = rows * cw * cspp * dt.itemsize
        if direct:
            from . import native
            dst = out[iy * ch:iy * ch + rows]
            got = native.tiff_lzw_decode_into(raw_u8, o, c, dst)
            if got < expected:
                raise TiffError(f"{path}: chunk {k} decodes to {got} bytes, {expected} expected")
            return
        data = _decompress(comp, bytes(buf[o:o + c]), expected)
        if len(data) < expected:
            raise TiffError(f"{path}: chunk {k} decodes to {len(data)} bytes, {expected} expected")
        a = np.frombuffer(data, dtype=dt, count=rows * cw * cspp).reshape(rows, cw, cspp)
        if pred == 2:     # horizontal differencing, per sample, modulo the sample width
            a = np.cumsum(a.astype(dt.newbyteorder("=")), axis=1, dtype=dt.newbyteorder("="))
        y0, x0 = iy * ch, ix * cw
        h, w = min(rows, H - y0), min(cw, W - x0)
        if planar == 2:
            out[y0:y0 + h, x0:x0 + w, pl] = a[:h, :w, 0]
        else:
            out[y0:y0 + h, x0:x0 + w, :] = a[:h, :w, :]

    nchunks = nx * ny * planes
    if comp != 1 and nchunks >= 4 and H * W * spp * dt.itemsize >= (1 << 20):
        from . import hostpool
        list(hostpool.pool().map(chunk, range(nchunks)))
    else:
        for k in range(nchunks):
            chunk(k)
    return out, t
