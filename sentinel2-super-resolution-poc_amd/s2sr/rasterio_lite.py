"""Minimal raster I/O for the file-level glue of the path (PIL only; rasterio / GDAL / cv2
are not required).  Covers what the reference does around the operator
(reference server/app/wow_sr.py:59-79,126-164): read bands 1-3 of a GeoTIFF (or any image
PIL decodes) as uint8 RGB, carry the georeferencing tags, write an LZW GeoTIFF whose pixel
size is divided by the SR scale, write a PNG.

Supported inputs: any image PIL decodes (8-bit RGB/RGBA/gray, 16-bit single band), and -- through
`tiff_lite` -- the multi-band 8/16/32-bit GeoTIFFs PIL refuses (Sentinel-2 deliveries: 4+ bands of
uint16, strips or tiles, LZW / Deflate / none, predictor 2, classic or BigTIFF).
"""
from __future__ import annotations

import os
import threading
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
from PIL import Image, TiffImagePlugin, TiffTags

Image.MAX_IMAGE_PIXELS = None

TAG_PIXEL_SCALE = 33550      # ModelPixelScaleTag  (sx, sy, sz)
TAG_TIEPOINT = 33922         # ModelTiepointTag
TAG_TRANSFORM = 34264        # ModelTransformationTag (4x4 row-major)
TAG_GEOKEYS = 34735          # GeoKeyDirectoryTag
TAG_GEODOUBLE = 34736
TAG_GEOASCII = 34737
_GEO_TAGS = (TAG_PIXEL_SCALE, TAG_TIEPOINT, TAG_TRANSFORM, TAG_GEOKEYS, TAG_GEODOUBLE, TAG_GEOASCII)


@dataclass
class GeoRef:
    """The GeoTIFF tags that place the raster; `scaled(s)` is the reference's
    `Affine(a/s, b, c, d, e/s, f)` (wow_sr.py:128-135): same origin, pixel size / s."""
    tags: Dict[int, tuple] = field(default_factory=dict)

    def scaled(self, s: float) -> "GeoRef":
        t = dict(self.tags)
        if TAG_PIXEL_SCALE in t:
            sx, sy, sz = (list(t[TAG_PIXEL_SCALE]) + [0.0, 0.0, 0.0])[:3]
            t[TAG_PIXEL_SCALE] = (sx / s, sy / s, sz)
        if TAG_TRANSFORM in t:
            m = list(t[TAG_TRANSFORM])
            m[0] /= s      # a
            m[5] /= s      # e
            t[TAG_TRANSFORM] = tuple(m)
        return GeoRef(t)

    @property
    def pixel_size(self) -> Optional[Tuple[float, float]]:
        if TAG_PIXEL_SCALE in self.tags:
            return float(self.tags[TAG_PIXEL_SCALE][0]), float(self.tags[TAG_PIXEL_SCALE][1])
        if TAG_TRANSFORM in self.tags:
            return float(self.tags[TAG_TRANSFORM][0]), -float(self.tags[TAG_TRANSFORM][5])
        return None


def _to_u8(img: np.ndarray, minmax_eps: float) -> np.ndarray:
    """Reference normalisation (wow_sr.py:67-73): min-max to 0..255 (truncating) only when
    max > 255, plain astype otherwise.  `minmax_eps` is the +1e-6 of apply_cnn_sr (:309)."""
    if img.dtype == np.uint8:
        return img
    if img.max() > 255:
        lo, hi = img.min(), img.max()
        return ((img - lo) / (hi - lo + minmax_eps) * 255).astype(np.uint8)
    return img.astype(np.uint8)


def read_rgb_u8(path: Path, minmax_eps: float = 0.0) -> Tuple[np.ndarray, Optional[GeoRef]]:
    """-> (HxWx3 uint8 RGB, GeoRef or None).  GeoRef is returned for .tif/.tiff inputs only,
    mirroring the reference's suffix switch (wow_sr.py:59,77-79)."""
    path = Path(path)
    is_tif = path.suffix.lower() in (".tif", ".tiff")
    pil_error = None
    if is_tif:
        # TIFFs go through tiff_lite first: like rasterio it returns the raw band values (PIL quietly
        # squeezes 3/4-band uint16 files to 8 bits and refuses more bands); PIL is the fallback for
        # layouts tiff_lite does not decode (JPEG-in-TIFF, ...)
        from . import tiff_lite
        try:
            arr, tv = tiff_lite.read_tiff(path)
        except tiff_lite.TiffError as e:
            pil_error = e
        else:
            # bands 1-3, or the single band three times (wow_sr.py:61-65)
            arr = arr[..., :3] if arr.shape[2] >= 3 else np.repeat(arr[..., :1], 3, axis=2)
            georef = GeoRef({t: (tv[t][0] if t == TAG_GEOASCII else tuple(tv[t])) for t in _GEO_TAGS if t in tv})
            return np.ascontiguousarray(_to_u8(arr, minmax_eps)), georef
    try:
        im = Image.open(path)
        im.load()
    except Exception as e:
        raise ValueError(f"{path}: unsupported raster layout (PIL: {e}"
                         + (f"; tiff_lite: {pil_error}" if pil_error else "") + ")") from e
    georef = None
    if is_tif:
        tags = {}
        tv2 = getattr(im, "tag_v2", {})
        for t in _GEO_TAGS:
            if t in tv2:
                v = tv2[t]
                tags[t] = tuple(v) if isinstance(v, (tuple, list)) else v
        georef = GeoRef(tags)
    if im.mode in ("RGB", "RGBA", "P", "CMYK", "LA", "1"):
        arr = np.asarray(im.convert("RGB"))
    else:                        # single band: L, I;16, I, F
        band = np.asarray(im)
        arr = np.stack([band, band, band], axis=-1)   # gray -> 3 bands (wow_sr.py:63-65)
    return np.ascontiguousarray(_to_u8(arr, minmax_eps)), georef


_ADLER = 65521


def _adler32_combine(a1: int, a2: int, len2: int) -> int:
    """adler32(A + B) from adler32(A), adler32(B), len(B) (zlib's adler32_combine)."""
    rem = len2 % _ADLER
    s1, s2 = a1 & 0xFFFF, (rem * (a1 & 0xFFFF)) % _ADLER
    s1 += (a2 & 0xFFFF) + _ADLER - 1
    s2 += ((a1 >> 16) & 0xFFFF) + ((a2 >> 16) & 0xFFFF) + _ADLER - rem
    if s1 >= _ADLER:
        s1 -= _ADLER
    if s1 >= _ADLER:
        s1 -= _ADLER
    if s2 >= 2 * _ADLER:
        s2 -= 2 * _ADLER
    if s2 >= _ADLER:
        s2 -= _ADLER
    return (s2 << 16) | s1


def encode_png_pieces(img: np.ndarray, level: int = 1, band_rows: int = 128, workers: Optional[int] = None,
                      strategy: Optional[int] = None, stream: bool = False):
    """HxWx3 (RGB) or HxWx4 (RGBA) uint8 -> the PNG file as a list of byte strings to be written in order.  The SR outputs are
    tens of megapixels and PNG deflate is what a job's writer spends its time in, so the image is cut into bands that are
    filtered (Sub) and deflated in parallel, each band ending on a sync flush so that the pieces concatenate into one valid zlib
    stream (the pigz construction).  Every band is its own IDAT chunk (libpng writes many IDAT chunks too): its CRC is computed
    in the band's thread and nothing walks or copies the 30-MB stream afterwards.
    Encoder settings = what `cv2.imwrite(path, img)` uses when the reference calls it without parameters (wow_sr.py:156,163;
    OpenCV 4.x grfmt_png.cpp: filter Sub, Z_BEST_SPEED, strategy Z_RLE).  With the defaults the bands go through the native
    encoder (csrc/pngenc.hip: the same filter and run-length matching, 2-3x zlib's speed); `level` / `strategy` other than the
    defaults select zlib.  Any setting decodes to the same pixels.  stream: a generator over the same pieces -- a band's chunk is
    handed out as soon as it and its predecessors are deflated (the bands run on the shared host pool), so a writer's loop runs under
    the deflating of the bands behind it."""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    img = np.ascontiguousarray(img)
    h, w, c = img.shape
    if img.dtype != np.uint8 or c not in (3, 4):
        raise ValueError(f"expected HxWx3 or HxWx4 uint8, got {img.shape} {img.dtype}")
    rows = img.reshape(h, w * c)
    bands = [(y, min(h, y + band_rows)) for y in range(0, h, band_rows)]
    idat_crc0 = zlib.crc32(b"IDAT")
    use_native = level == 1 and strategy in (None, zlib.Z_RLE)

    def work_native(i):                                            # -> (one complete IDAT chunk, adler, filtered bytes)
        from . import native
        y0, y1 = bands[i]
        return native.png_idat_band(img[y0:y1], i == 0, i == len(bands) - 1)

    def work_zlib(i):
        y0, y1 = bands[i]
        blk = rows[y0:y1]
        raw = np.empty((y1 - y0, w * c + 1), np.uint8)
        raw[:, 0] = 1                                              # filter type Sub
        raw[:, 1:c + 1] = blk[:, :c]
        np.subtract(blk[:, c:], blk[:, :-c], out=raw[:, c + 1:])
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_RLE if strategy is None else strategy)
        out = (b"\x78\x5e" if i == 0 else b"") + co.compress(raw) + co.flush(zlib.Z_FINISH if i == len(bands) - 1 else zlib.Z_SYNC_FLUSH)
        if len(out) >= (1 << 31):
            raise ValueError("a band deflates to more than an IDAT chunk holds; lower band_rows")
        piece = struct.pack(">I", len(out)) + b"IDAT" + out + struct.pack(">I", zlib.crc32(out, idat_crc0) & 0xFFFFFFFF)
        return piece, zlib.adler32(raw), raw.size

    work = work_native if use_native else work_zlib

    def chunk(kind: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)

    head = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 6, 0, 0, 0))

    def pieces():
        # signature, IHDR, the bands' IDAT chunks in order, one small IDAT with the stream's Adler-32 (IDAT payloads concatenate), IEND
        yield head
        adler = 1
        if len(bands) > 1 and workers is None:
            from . import hostpool
            it = hostpool.pool().map(work, range(len(bands)))
            for p, a, ln in it:
                adler = _adler32_combine(adler, a, ln)
                yield p
        elif len(bands) > 1 and workers > 1:
            with ThreadPoolExecutor(max_workers=workers) as pool:
                for p, a, ln in pool.map(work, range(len(bands))):
                    adler = _adler32_combine(adler, a, ln)
                    yield p
        else:
            for i in range(len(bands)):
                p, a, ln = work(i)
                adler = _adler32_combine(adler, a, ln)
                yield p
        yield chunk(b"IDAT", struct.pack(">I", adler)) + chunk(b"IEND", b"")

    return pieces() if stream else list(pieces())


def encode_png(img: np.ndarray, level: int = 1, band_rows: int = 128, workers: Optional[int] = None, strategy: Optional[int] = None) -> bytes:
    """The pieces of `encode_png_pieces` as one byte string (for callers that want the file in memory)."""
    return b"".join(encode_png_pieces(img, level, band_rows, workers, strategy))


def write_pieces(path: Path, pieces) -> None:
    """The byte strings `pieces` one behind the other into `path`.  One write loop: measured on the 16-CPU GPU box, 43 MB of encoded
    strips go out in 5.0 ms this way and in 5.9 ms as positional writes from the host pool into a pre-sized file
    (tools/write_probe.py, r05) -- the page cache takes 8 GB/s from one thread; what a job's writers wait for is the encoders."""
    ok = False
    try:
        with open(path, "wb") as f:
            f.writelines(pieces)
        ok = True
    finally:
        if not ok:                 # (the pieces may be a generator that still encodes: no half-written file stays behind its failure)
            try:
                os.unlink(path)
            except OSError:
                pass


def write_png(path: Path, rgb: np.ndarray) -> None:
    write_pieces(path, encode_png_pieces(rgb, stream=True))       # (a generator: bands reach the file while later bands deflate)


# ---- the rasters this process wrote last, for the stage that reads them right back ---------------------------------------------
# A /api/wow job writes <stem>_wow_sr.tif and then hands that PATH to the tiler (reference main.py:347-359: process_wow_sr, then
# process_raster_to_tiles on its sr_tif): the pyramid's first 24 ms were the LZW decode of the 50 MB that had been an array in this
# process a moment before.  On request (`remember=True`: the job's writers, whose array nobody writes to again) write_geotiff_rgb
# keeps a read-only view of the last two arrays it wrote, keyed by the file's resolved path and valid while the file's size and
# mtime are what they were after the write; `recall_written` hands them to app.tiling.  Nothing else reads through it; a file touched by anyone else is read from disk like any other.
_WRITTEN: Dict[str, tuple] = {}
_WRITTEN_LOCK = threading.Lock()


def _remember_written(path, rgb: np.ndarray, georef: "GeoRef") -> None:
    try:
        key = str(Path(path).resolve())
        st = os.stat(key)
        view = rgb.view()
        view.flags.writeable = False
        tags = {t: (tuple(v) if isinstance(v, (tuple, list)) else (v,)) for t, v in georef.tags.items()}
        with _WRITTEN_LOCK:
            _WRITTEN.pop(key, None)
            while len(_WRITTEN) >= 2:
                _WRITTEN.pop(next(iter(_WRITTEN)))
            _WRITTEN[key] = (st.st_mtime_ns, st.st_size, view, tags)
    except OSError:
        pass


def recall_written(path):
    """-> (uint8 [H, W, 3] read-only array, {tag: values}) when `path` is a GeoTIFF this process wrote and nobody has touched since; else None."""
    try:
        key = str(Path(path).resolve())
        st = os.stat(key)
    except OSError:
        return None
    with _WRITTEN_LOCK:
        hit = _WRITTEN.get(key)
        if hit is None:
            return None
        if hit[0] != st.st_mtime_ns or hit[1] != st.st_size:
            _WRITTEN.pop(key, None)
            return None
        return hit[2], dict(hit[3])


def forget_written() -> None:
    with _WRITTEN_LOCK:
        _WRITTEN.clear()


def write_outputs(rgb: np.ndarray, png_path: Path, tif_path: Path, georef: "GeoRef", remember: bool = False) -> None:
    """The GeoTIFF and the PNG of one job (wow_sr.py:126-164) written side by side: two threads, each a pool over strips /
    bands of the same array (zlib and the native LZW encoder release the GIL).  remember: see write_geotiff_rgb."""
    import threading
    err = []

    def run(fn, *a):
        try:
            fn(*a)
        except BaseException as e:      # surfaced below: a failed writer must fail the job
            err.append(e)
    t = threading.Thread(target=run, args=(write_geotiff_rgb, tif_path, rgb, georef, 64, remember))
    t.start()
    run(write_png, png_path, rgb)
    t.join()
    if err:
        raise err[0]


def write_geotiff_rgb(path: Path, rgb: np.ndarray, georef: GeoRef, rows_per_strip: int = 64, remember: bool = False) -> None:
    """uint8 RGB, LZW compressed (compress="lzw", wow_sr.py:138-151) with the geo tags.  Classic
    little-endian TIFF, chunky RGB strips; the strips are LZW-encoded by the native library on a
    thread pool (the SR outputs are tens of megapixels; a single-threaded encoder is what a job would
    otherwise wait for).  remember: the caller will not write to `rgb` again -- a view of it is kept for the stage that reads this
    file right back (`recall_written`)."""
    import os
    import struct
    from concurrent.futures import ThreadPoolExecutor

    from . import native

    rgb = np.asarray(rgb)
    if rgb.ndim != 3 or rgb.shape[2] not in (3, 4) or (rgb.shape[2] == 4 and remember):
        raise ValueError(f"expected HxWx3 (or HxWx4, whose alpha is dropped), got {rgb.shape}")
    h, w, c = rgb.shape
    if c == 3:
        rgb = np.ascontiguousarray(rgb, np.uint8)
        strip_bytes = lambda s: rgb[s[0]:s[1]].reshape(-1)      # noqa: E731
    else:
        # an RGBA raster in hand (the tiler's warp result): its alpha is dropped strip by strip inside the pool -- a whole-image
        # rgba[..., :3] copy in front of the encoder is one thread moving a gigabyte for a 16k x 16k raster (tools/job_big_probe.py)
        rgb = rgb if rgb.dtype == np.uint8 else rgb.astype(np.uint8)
        strip_bytes = lambda s: np.ascontiguousarray(rgb[s[0]:s[1], :, :3]).reshape(-1)      # noqa: E731
    strips = [(y, min(h, y + rows_per_strip)) for y in range(0, h, rows_per_strip)]
    from . import hostpool
    # the strips go to the file as they come off the pool, in order (the offsets of a strip are known once its predecessors' sizes
    # are): the write loop runs under the encoding of the strips behind it instead of after all of it; the IFD follows the data, and
    # the header's pointer to it is patched in last
    offs, sizes, pos = [], [], 8
    ok = False
    try:
        with open(path, "wb") as f:
            f.write(b"II" + struct.pack("<HI", 42, 0))
            for e in hostpool.pool().map(lambda s: native.tiff_lzw_encode(strip_bytes(s)), strips):
                offs.append(pos)
                sizes.append(len(e))
                f.write(e)
                if len(e) & 1:
                    f.write(b"\0")
                pos += len(e) + (len(e) & 1)
                if pos >= (1 << 32) - (1 << 20):
                    raise ValueError("output exceeds the 4 GiB of a classic TIFF")
            ent = [(256, 4, (w,)), (257, 4, (h,)), (258, 3, (8, 8, 8)), (259, 3, (5,)), (262, 3, (2,)), (273, 4, tuple(offs)),
                   (277, 3, (3,)), (278, 4, (rows_per_strip,)), (279, 4, tuple(sizes)), (284, 3, (1,)), (339, 3, (1, 1, 1))]
            for tag, val in georef.tags.items():
                if tag == TAG_GEOKEYS:
                    ent.append((tag, 3, tuple(int(v) for v in val)))
                elif tag == TAG_GEOASCII:
                    ent.append((tag, 2, val if isinstance(val, str) else str(val)))
                else:
                    ent.append((tag, 12, tuple(float(v) for v in val)))
            ent.sort(key=lambda e: e[0])
            fmt = {3: "H", 4: "I", 12: "d"}
            ifd_off = pos
            val_pos = ifd_off + 2 + 12 * len(ent) + 4
            ifd, tail = b"", b""
            for tag, typ, vals in ent:
                if typ == 2:
                    data = vals.encode("latin-1") + b"\0"
                    cnt = len(data)
                else:
                    data = struct.pack("<" + fmt[typ] * len(vals), *vals)
                    cnt = len(vals)
                e = struct.pack("<HHI", tag, typ, cnt)
                if len(data) <= 4:
                    e += data.ljust(4, b"\0")
                else:
                    e += struct.pack("<I", val_pos + len(tail))
                    tail += data + (b"\0" if len(data) & 1 else b"")
                ifd += e
            f.write(struct.pack("<H", len(ent)) + ifd + struct.pack("<I", 0) + tail)
            f.seek(4)
            f.write(struct.pack("<I", ifd_off))
        ok = True
    finally:
        if not ok:
            try:
                os.unlink(path)            # no half-written GeoTIFF stays behind a failed writer
            except OSError:
                pass
    if remember:
        _remember_written(path, rgb, georef)
