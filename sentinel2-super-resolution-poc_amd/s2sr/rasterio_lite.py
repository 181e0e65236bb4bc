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


def write_png(path: Path, rgb: np.ndarray) -> None:
    Image.fromarray(np.ascontiguousarray(rgb), "RGB").save(str(path), format="PNG", compress_level=3)


def write_geotiff_rgb(path: Path, rgb: np.ndarray, georef: GeoRef) -> None:
    """uint8 RGB, LZW compressed (compress="lzw", wow_sr.py:138-151) with the geo tags."""
    ifd = TiffImagePlugin.ImageFileDirectory_v2()
    for tag, val in georef.tags.items():
        if tag == TAG_GEOKEYS:
            ifd[tag] = tuple(int(v) for v in val)
            ifd.tagtype[tag] = TiffTags.SHORT
        elif tag == TAG_GEOASCII:
            ifd[tag] = val if isinstance(val, str) else str(val)
            ifd.tagtype[tag] = TiffTags.ASCII
        else:
            ifd[tag] = tuple(float(v) for v in val)
            ifd.tagtype[tag] = TiffTags.DOUBLE
    Image.fromarray(np.ascontiguousarray(rgb), "RGB").save(str(path), format="TIFF", compression="tiff_lzw",
                                                           tiffinfo=ifd)
