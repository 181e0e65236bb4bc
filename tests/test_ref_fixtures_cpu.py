"""The reference's own recorded artefacts as goldens (tests/golden/ref_*.json, copied as DATA by tools/make_golden.py from
/root/reference/data): the result dict / metadata JSON that `process_wow_sr` wrote for its two recorded /api/enhance jobs
(reference server/app/wow_sr.py:166-182,243-259: JPEG upload -> PNG-only output, "sr_tif": null; realesrgan_x4 and
realesrgan_anime) and the tileset.json contract of `create_tileset_metadata` (server/app/tiling.py:189-224).

The app mirror must reproduce them key for key, value type for value type, and value for value wherever the value is not a
path or a timestamp.  Runs without a GPU: the SR operator is replaced by a nearest x4 stand-in (the schema is host code)."""
import json
import re
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden"


def _shape(v):
    """Recursive (key, type) skeleton of a JSON value."""
    if isinstance(v, dict):
        return {k: _shape(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_shape(x) for x in v]
    return type(v).__name__


@pytest.mark.parametrize("fixture,model", [("ref_wow_sr_metadata_x4.json", "realesrgan_x4"),
                                           ("ref_wow_sr_metadata_anime.json", "realesrgan_anime")])
def test_process_wow_sr_reproduces_the_references_recorded_result(tmp_path, monkeypatch, fixture, model):
    import app.wow_sr as wow
    from s2sr import rasterio_lite as rio
    want = json.loads((GOLDEN / fixture).read_text())
    g8 = np.load(GOLDEN / "g8_real_image.npz")
    rgb = np.ascontiguousarray(g8["img_bgr"][:, :, ::-1])                     # the reference's own upload, decoded
    assert list(rgb.shape[:2]) == want["sr_metadata"]["original_size"]        # [576, 432]: (height, width), wow_sr.py:82
    src = tmp_path / "uploads" / Path(want["input"]).name.replace(".jpg", ".png")   # same stem; PNG (lossless) in place of the JPEG
    src.parent.mkdir(parents=True)
    rio.write_png(src, rgb)

    class FakeESRGAN:                                                          # stands in for the GPU operator: schema test only
        def __init__(self, scale=4, device=None, tile_size=256, model_name=None):
            self.scale, self.model_name = scale, model_name

        def enhance(self, img):
            return np.repeat(np.repeat(img, 4, axis=0), 4, axis=1)
    monkeypatch.setattr(wow, "RealESRGAN", FakeESRGAN)
    monkeypatch.setattr(wow, "_enhance_for_crops", lambda img: img)
    got = wow.process_wow_sr(src, tmp_path / "wow", enhance_crops=True, model=model)

    assert list(got) == list(want) and list(got["outputs"]) == list(want["outputs"])          # keys AND their order
    assert list(got["sr_metadata"]) == list(want["sr_metadata"])
    assert _shape(got) == _shape(want)                                                          # every value's JSON type
    assert got["outputs"]["sr_tif"] is None and want["outputs"]["sr_tif"] is None               # non-GeoTIFF input -> PNG only
    assert re.fullmatch(r"\d{8}_\d{6}", got["timestamp"]) and re.fullmatch(r"\d{8}_\d{6}", want["timestamp"])
    stem = Path(want["input"]).stem
    assert Path(got["outputs"]["sr_png"]).name == Path(want["outputs"]["sr_png"]).name == f"{stem}_wow_sr.png"
    assert Path(got["sr_metadata"]["output_file"]).name == Path(want["sr_metadata"]["output_file"]).name
    for k, v in want["sr_metadata"].items():                                                    # everything that is not a path: equal
        if k not in ("input_file", "output_file"):
            assert got["sr_metadata"][k] == v, k
    assert got["sr_metadata"]["stages"][0]["model"] == model
    on_disk = json.loads((tmp_path / "wow" / f"{stem}_wow_sr_metadata.json").read_text())       # the file the job writes
    assert on_disk == got
    out, geo = rio.read_rgb_u8(got["outputs"]["sr_png"])
    assert geo is None and list(out.shape[:2]) == want["sr_metadata"]["output_size"]


def test_tileset_metadata_reproduces_the_references_recorded_file(tmp_path):
    from app.tiling import create_tileset_metadata
    want = json.loads((GOLDEN / "ref_tileset.json").read_text())
    got = create_tileset_metadata(tmp_path / "tiles_wow", want["bounds"], want["minzoom"], want["maxzoom"])
    assert list(got) == list(want) and _shape(got) == _shape(want) and got == want
    assert json.loads((tmp_path / "tiles_wow" / "tileset.json").read_text()) == want


def test_oracle_on_the_references_real_image(golden_dir):
    """The oracle pinned on natural-image statistics too: the reference's upload (decoded pixels in g8_real_image.npz)
    through the imported reference net, 23 and 6 blocks on a 64x96 crop, and the whole 576x432 image (whole-image branch of
    enhance(): 248,832 px <= 4 * 256^2) through the 6-block net at eight 64x64 output windows."""
    import torch
    from oracle import rrdbnet_ref as ref
    from s2sr.weights import synthetic_state_dict
    torch.set_num_threads(8)
    g = np.load(golden_dir / "g8_real_image.npz")
    for nb in (23, 6):
        sd = ref.to_torch_sd(synthetic_state_dict(nb, seed=0))
        q, f = ref.enhance(g["crop_bgr"], sd, nb, return_float=True)
        assert np.abs(f - g[f"crop_out_f32_b{nb}"]).max() <= 1e-5
        assert np.abs(q.astype(np.int16) - g[f"crop_out_u8_b{nb}"].astype(np.int16)).max() <= 1
    sd6 = ref.to_torch_sd(synthetic_state_dict(6, seed=0))
    q, f = ref.enhance(g["img_bgr"], sd6, 6, return_float=True)
    assert f.shape == (2304, 1728, 3)
    for (y, x), wf, wq in zip(g["full_win_yx"], g["full_win_f32_b6"], g["full_win_u8_b6"]):
        assert np.abs(f[y:y + 64, x:x + 64] - wf).max() <= 1e-5
        assert np.abs(q[y:y + 64, x:x + 64].astype(np.int16) - wq.astype(np.int16)).max() <= 1
    assert abs(f.mean(dtype=np.float64) - g["full_mean_std_b6"][0]) <= 1e-6
