"""Synthetic inputs for tests and bench: image-like uint8 tiles.

uint8[B,S,S,3] from numpy PCG64(seed): white noise, 5x5 box filter, per-tile random gain and
offset so that mean ~110 / sigma ~45; `green=True` biases toward the fallback-image statistics of
the reference's fetcher (G in [80,180), R,B in [40,120), server/app/up42_client.py:684-690) so
the hue 36..84 branch of the vegetation boost is exercised.
"""
from __future__ import annotations

import numpy as np


def _box5(a: np.ndarray) -> np.ndarray:
    """5x5 box filter with edge replication over axes (1, 2) of [B,H,W,C] float32."""
    p = np.pad(a, ((0, 0), (2, 2), (2, 2), (0, 0)), mode="edge")
    c = np.cumsum(p, axis=1)
    c = np.concatenate([c[:, 4:5], c[:, 5:] - c[:, :-5]], axis=1)
    d = np.cumsum(c, axis=2)
    d = np.concatenate([d[:, :, 4:5], d[:, :, 5:] - d[:, :, :-5]], axis=2)
    return d / 25.0


def synthetic_tiles(batch: int, size: int = 256, seed: int = 1234, green: bool = False) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    noise = rng.integers(0, 256, size=(batch, size, size, 3)).astype(np.float32)
    sm = _box5(noise)                                  # sigma ~ 14.8 around 127.5
    gain = rng.uniform(2.0, 4.0, size=(batch, 1, 1, 3)).astype(np.float32)
    off = rng.uniform(80.0, 140.0, size=(batch, 1, 1, 3)).astype(np.float32)
    if green:
        off = off * np.array([0.6, 1.2, 0.6], np.float32)
    out = (sm - 127.5) * gain + off
    return np.clip(out, 0, 255).astype(np.uint8)
