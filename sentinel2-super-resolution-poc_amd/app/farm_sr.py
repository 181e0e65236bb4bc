"""Drop-in for the reference module `app.farm_sr` (reference server/app/farm_sr.py): the
`/api/sr` variant -- same RRDBNet x4, post-process constants CLAHE 2.5/8, unsharp 1.2/1.5,
vegetation x1.3 (farm_sr.py:170-178).  `enhance_crop_rows` (:18-58) is dead code in the
reference (never called) and is not reproduced."""
from __future__ import annotations

import json
from datetime import datetime
from pathlib import Path
from typing import Tuple

import numpy as np

from app.cnn_super_resolution import RealESRGAN
from app.wow_sr import _pp_engine
from s2sr import native


def apply_unsharp_mask(img: np.ndarray, strength: float = 1.5, radius: float = 1.0) -> np.ndarray:
    """addWeighted(img, 1+s, GaussianBlur(img, sigma=radius), -s) (farm_sr.py:61-71)."""
    p = native.PPParams(0.0, 8, float(radius), 1.0 + float(strength), -float(strength), 35, 85, 1.0, 2)
    return _pp_engine().postprocess_u8(img, p)


def enhance_local_contrast(img: np.ndarray, clip_limit: float = 3.0, grid_size: int = 8) -> np.ndarray:
    """RGB->Lab, CLAHE on L, Lab->RGB (farm_sr.py:74-88)."""
    p = native.PPParams(float(clip_limit), int(grid_size), 1.0, 1.0, 0.0, 35, 85, 1.0, 1)
    return _pp_engine().postprocess_u8(img, p)


def enhance_vegetation(img: np.ndarray) -> np.ndarray:
    """Saturation x1.3 where 35 < H < 85 in 8-bit HSV (farm_sr.py:91-108)."""
    p = native.PPParams(0.0, 8, 1.0, 1.0, 0.0, 35, 85, 1.3, 4)
    return _pp_engine().postprocess_u8(img, p)


def apply_farm_sr(input_path: Path, output_path: Path, scale: int = 4) -> Tuple[Path, dict]:
    """Reference farm_sr.py:111-241."""
    from s2sr import rasterio_lite as rio

    print(f"\nFarm Super-Resolution x{scale}\n   Input: {input_path}")
    input_path = Path(input_path)
    img, georef = rio.read_rgb_u8(input_path)
    original_shape = img.shape[:2]
    # scale 2|3 -> model name "realesrgan_x2|3" is not in MODELS -> ValueError, as in the reference (:162)
    # S2SR_FARM_PRECISION=fp8 puts the /api/sr path on the fp8 (e4m3) trunk -- BASELINE.json configs[4]; default: as /api/wow
    import os
    from app.cnn_super_resolution import thread_precision
    with thread_precision(os.environ.get("S2SR_FARM_PRECISION") or None):
        esrgan = RealESRGAN(scale=scale, tile_size=256)
        if hasattr(esrgan, "enhance_job"):      # RGB2BGR -> net -> BGR2RGB -> the three steps of :170-178, one native call
            final = esrgan.enhance_job(img, native.pp_farm())
        else:
            sr_rgb = np.ascontiguousarray(esrgan.enhance(np.ascontiguousarray(img[:, :, ::-1]))[:, :, ::-1])
            final = _pp_engine().postprocess_u8(sr_rgb, native.pp_farm())

    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    output_png = output_path.with_suffix(".png")
    if georef is not None:
        final_output = output_path.with_suffix(".tif")
        rio.write_outputs(final, output_png, final_output, georef.scaled(scale), remember=True)
    else:
        final_output = output_png
        rio.write_png(output_png, final)
    metadata = {
        "input_file": str(input_path),
        "output_file": str(final_output),
        "scale": scale,
        "model": f"RealESRGAN_farm_x{scale}",
        "enhancements": ["Real-ESRGAN super-resolution", "CLAHE local contrast", "Unsharp mask edge sharpening",
                         "Vegetation enhancement"],
        "original_size": list(original_shape),
        "output_size": list(final.shape[:2]),
        "original_resolution_m": 10.0,
        "optimized_for": "crop_row_visibility",
    }
    return final_output, metadata


def process_farm_sr(input_tif: Path, output_dir: Path, scale: int = 4) -> dict:
    """Reference farm_sr.py:244-286."""
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    base_name = Path(input_tif).stem
    sr_tif = output_dir / f"{base_name}_farm_sr_x{scale}.tif"
    _, sr_metadata = apply_farm_sr(input_path=input_tif, output_path=sr_tif, scale=scale)
    png = sr_tif.with_suffix(".png")
    result = {
        "timestamp": datetime.now().strftime("%Y%m%d_%H%M%S"),
        "input": str(input_tif),
        "outputs": {"sr_tif": str(sr_tif) if sr_tif.exists() else None, "sr_png": str(png) if png.exists() else None},
        "sr_metadata": sr_metadata,
    }
    with open(output_dir / f"{base_name}_farm_sr_metadata.json", "w") as f:
        json.dump(result, f, indent=2)
    return result
