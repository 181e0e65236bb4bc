"""Drop-in for the reference module `app.wow_sr` (reference server/app/wow_sr.py): Real-ESRGAN x4
followed by the crop-visibility post-process, both on the GPU through libs2sr.so."""
from __future__ import annotations

import json
import threading
from datetime import datetime
from pathlib import Path
from typing import Tuple

import numpy as np

from app.cnn_super_resolution import RealESRGAN
from s2sr import native

_PP_LOCK = threading.Lock()
_PP_ENGINE = {}


def _pp_engine(device_index: int = None) -> native.Engine:
    """Handle used for post-process only (it needs no weights); one per GPU, shared by all jobs
    (s2sr_postprocess_u8 holds the handle's lock from upload to download)."""
    if device_index is None:
        from app.cnn_super_resolution import current_device_index
        device_index = current_device_index()
    with _PP_LOCK:
        if device_index not in _PP_ENGINE:
            _PP_ENGINE[device_index] = native.Engine(num_block=1, device=device_index)
        return _PP_ENGINE[device_index]


def _enhance_for_crops(img: np.ndarray) -> np.ndarray:
    """CLAHE(2.5, 8x8) on L -> unsharp (sigma 1.2, 1.4/-0.4) -> saturation x1.2 on hue 36..84
    (reference wow_sr.py:187-209): HxWx3 uint8 RGB -> same, one fused GPU pass sequence."""
    return _pp_engine().postprocess_u8(img, native.pp_wow())


def apply_wow_sr(input_path: Path, output_path: Path, enhance_crops: bool = True,
                 model: str = "realesrgan_x4") -> Tuple[Path, dict]:
    """Reference wow_sr.py:28-184 -- same outputs (GeoTIFF and/or PNG) and metadata dict."""
    from s2sr import rasterio_lite as rio

    model_display = {"realesrgan_x4": "Real-ESRGAN x4",
                     "realesrgan_anime": "Real-ESRGAN Anime 6B (text/plates)"}.get(model, model)
    print(f"\nWOW Super-Resolution ({model_display} + Enhanced)\n   Input: {input_path}")
    input_path = Path(input_path)
    img, georef = rio.read_rgb_u8(input_path)           # u8 RGB (min-max normalised if >255, :67-73)
    original_shape = img.shape[:2]

    pipeline_stages = []
    print(f"   Stage 1/2: {model_display} (GAN upscaling)...")
    esrgan = RealESRGAN(model_name=model, tile_size=256)
    # RGB2BGR -> enhance -> BGR2RGB -> _enhance_for_crops (:85-110) as ONE native call: the 16x image crosses PCIe once
    if enhance_crops:
        print("   Stage 2/2: Crop visibility enhancement...")
    if hasattr(esrgan, "enhance_job"):
        output_rgb = esrgan.enhance_job(img, native.pp_wow() if enhance_crops else None)
    else:      # an operator with the reference's interface only: the reference's own sequence
        output_rgb = np.ascontiguousarray(esrgan.enhance(np.ascontiguousarray(img[:, :, ::-1]))[:, :, ::-1])   # the net is fed BGR (:85,94)
        if enhance_crops:
            output_rgb = _enhance_for_crops(output_rgb)
    scale = esrgan.scale
    del esrgan
    pipeline_stages.append({"model": model, "scale": scale, "purpose": "GAN upscaling"})
    if enhance_crops:
        pipeline_stages.append({"post_processing": "Enhanced", "purpose": "Crop visibility"})
    final_shape = output_rgb.shape[:2]

    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    output_png = output_path.with_suffix(".png")
    if georef is not None:
        final_output = output_path.with_suffix(".tif")
        # the two encoders side by side (both are thread pools over strips / bands of the same array)
        rio.write_outputs(output_rgb, output_png, final_output, georef.scaled(scale), remember=True)   # pixel size / scale (:128-135)
    else:
        final_output = output_png
        rio.write_png(output_png, output_rgb)

    metadata = {
        "input_file": str(input_path),
        "output_file": str(final_output),
        "scale": scale,
        "pipeline": "Real-ESRGAN x4 + Enhanced",
        "stages": pipeline_stages,
        "enhancements": (["CLAHE local contrast", "Unsharp mask", "Vegetation boost"] if enhance_crops else []),
        "original_size": list(original_shape),
        "output_size": list(final_shape),
        "original_resolution_m": 10.0,
        "effective_resolution_m": 10.0 / scale,
        "optimized_for": "z18_crop_visibility",
    }
    return final_output, metadata


def process_wow_sr(input_tif: Path, output_dir: Path, enhance_crops: bool = True,
                   model: str = "realesrgan_x4") -> dict:
    """Reference wow_sr.py:212-266 -- file naming, metadata JSON and result dict schema."""
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    base_name = Path(input_tif).stem
    wow_tif = output_dir / f"{base_name}_wow_sr.tif"
    _, sr_metadata = apply_wow_sr(input_path=input_tif, output_path=wow_tif, enhance_crops=enhance_crops, model=model)
    png = wow_tif.with_suffix(".png")
    result = {
        "timestamp": datetime.now().strftime("%Y%m%d_%H%M%S"),
        "input": str(input_tif),
        "outputs": {"sr_tif": str(wow_tif) if wow_tif.exists() else None,
                    "sr_png": str(png) if png.exists() else None},
        "sr_metadata": sr_metadata,
    }
    with open(output_dir / f"{base_name}_wow_sr_metadata.json", "w") as f:
        json.dump(result, f, indent=2)
    return result


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser(description="WOW Super-Resolution (MI355X)")
    ap.add_argument("input")
    ap.add_argument("-o", "--output", default="./wow_sr_output")
    ap.add_argument("--no-enhance", action="store_true")
    a = ap.parse_args()
    print(process_wow_sr(Path(a.input), Path(a.output), enhance_crops=not a.no_enhance)["outputs"])
