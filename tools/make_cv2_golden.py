#!/usr/bin/env python3
"""Pin the post-process oracle to OpenCV -- for whoever has a cv2 wheel (this build's container has none: SURVEY.md 8c).

Runs the cv2 calls of the reference's `_enhance_for_crops` (server/app/wow_sr.py:187-209) and of the farm chain
(server/app/farm_sr.py:61-108, constants of apply_farm_sr :170-178) on the repo's deterministic test images and on the decoded
pixels of the reference's own upload (tests/golden/g8_real_image.npz), keeps every intermediate stage, and writes
tests/golden/g9_cv2_postprocess.npz.  With that file present, tests/test_postprocess_oracle.py::test_oracle_against_cv2_golden
holds oracle/postprocess_ref.py to it stage by stage (<= 2 LSB, exact-match rate printed) and
tests/test_gpu_postprocess.py::test_gpu_against_cv2_golden does the same for the HIP kernels; without it both skip.

    pip install "opencv-contrib-python>=4.8.0"      # the reference's own requirement (server/requirements.txt:29)
    python tools/make_cv2_golden.py

Nothing of the reference is imported: the calls are restated here with the constants cited."""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / "tests" / "golden"


def images():
    """The images the oracle tests use: noise, a smooth green-dominant field, ragged sizes (CLAHE pads), a gradient, a flat
    plane, and a crop of the reference's real upload."""
    rng = np.random.default_rng(4)
    out = {}
    base = rng.integers(0, 256, (96, 128, 3)).astype(np.float32)
    sm = np.stack([np.convolve(base[..., c].ravel(), np.ones(25) / 25, "same").reshape(96, 128) for c in range(3)], -1)
    out["green"] = np.clip(sm * np.array([0.5, 1.0, 0.45]) + np.array([20, 60, 10]), 0, 255).astype(np.uint8)
    out["noise"] = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    out["ragged"] = rng.integers(0, 256, (67, 101, 3), dtype=np.uint8)
    out["half_ragged"] = rng.integers(0, 256, (64, 100, 3), dtype=np.uint8)
    out["flat"] = np.full((40, 48, 3), 100, np.uint8)
    grad = np.zeros((128, 256, 3), np.uint8)
    grad[..., 0] = np.arange(256)[None, :]
    grad[..., 1] = (np.arange(128) * 2)[:, None]
    grad[..., 2] = 255 - np.arange(256)[None, :]
    out["gradient"] = grad
    g8 = GOLDEN / "g8_real_image.npz"
    if g8.exists():
        bgr = np.load(g8)["img_bgr"]
        out["real"] = np.ascontiguousarray(bgr[100:356, 80:400, ::-1])       # RGB crop of the reference's upload
    return out


def stages(cv2, img, clip, grid, sigma, w_img, w_blur, gain):
    """Every intermediate of CLAHE-on-L -> unsharp -> vegetation boost, as the reference computes them."""
    s = {}
    lab = cv2.cvtColor(img, cv2.COLOR_RGB2LAB)                                   # wow_sr.py:190 / farm_sr.py:79
    s["lab"] = lab.copy()
    lab[:, :, 0] = cv2.createCLAHE(clipLimit=clip, tileGridSize=(grid, grid)).apply(lab[:, :, 0])   # :191-192 / :82-83
    s["clahe_l"] = lab[:, :, 0].copy()
    enhanced = cv2.cvtColor(lab, cv2.COLOR_LAB2RGB)                              # :193 / :86
    s["contrast"] = enhanced
    blurred = cv2.GaussianBlur(enhanced, (0, 0), sigma)                          # :196 / :66
    s["blur"] = blurred
    sharpened = cv2.addWeighted(enhanced, w_img, blurred, w_blur, 0)             # :197 / :69
    s["sharp"] = sharpened
    hsv = cv2.cvtColor(sharpened, cv2.COLOR_RGB2HSV)                             # :200 / :94
    s["hsv"] = hsv.copy()
    h = hsv.astype(np.float32)
    mask = ((h[:, :, 0] > 35) & (h[:, :, 0] < 85)).astype(np.float32)
    h[:, :, 1] = np.where(mask > 0, np.clip(h[:, :, 1] * gain, 0, 255), h[:, :, 1])   # :202-206 / :97-103
    s["final"] = np.clip(cv2.cvtColor(h.astype(np.uint8), cv2.COLOR_HSV2RGB), 0, 255).astype(np.uint8)   # :207-209 / :106
    return s


def main():
    try:
        import cv2
    except ImportError:
        sys.exit("cv2 is not installed here: pip install 'opencv-contrib-python>=4.8.0' and run again")
    out = {"cv2_version": np.array(cv2.__version__)}
    for name, img in images().items():
        out[f"{name}.img"] = img
        for tag, prm in (("wow", (2.5, 8, 1.2, 1.4, -0.4, 1.2)), ("farm", (2.5, 8, 1.5, 2.2, -1.2, 1.3))):
            for k, v in stages(cv2, img, *prm).items():
                out[f"{name}.{tag}.{k}"] = v
    GOLDEN.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(GOLDEN / "g9_cv2_postprocess.npz", **out)
    print(f"wrote {GOLDEN / 'g9_cv2_postprocess.npz'} with OpenCV {cv2.__version__}: {len(out) - 1} arrays")


if __name__ == "__main__":
    main()
