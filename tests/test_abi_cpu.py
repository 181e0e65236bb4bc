"""CPU-side checks of the native boundary: the library loads, exports every symbol that
include/s2sr.h declares, the pure host function s2sr_plan_tiles matches the reference's
window plans, and the product path fails loudly without a GPU (no fallback)."""
import re
from pathlib import Path

import numpy as np
import pytest

from s2sr import native

REPO = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    lib = native.load_library()
    header = (REPO / "include" / "s2sr.h").read_text()
    declared = set(re.findall(r"\b(s2sr_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/s2sr.h but not exported"
    assert declared == set(native.EXPORTED_SYMBOLS)
    assert b"gfx950" in lib.s2sr_version()


def test_expected_blob_size():
    lib = native.load_library()
    assert lib.s2sr_expected_blob_floats(23) == 16_697_987
    from s2sr.weights import num_params
    assert lib.s2sr_expected_blob_floats(6) == num_params(6)


def test_plan_tiles_matches_reference_golden(golden_dir):
    g = np.load(golden_dir / "g6_tile_plans.npz")
    for key in g.files:
        h, w = map(int, key.split("x"))
        ts, tp = (16, 2) if (h, w) == (37, 45) else (256, 10)
        if h * w <= ts * ts * 4:
            continue   # whole-image branch: no plan
        wins = native.plan_tiles(h, w, ts, tp, 4)
        rects = np.array([(q.y1, q.y2, q.x1, q.x2) for q in wins], dtype=np.int32)
        assert np.array_equal(rects, g[key]), key


def test_plan_tiles_matches_oracle_everywhere():
    from oracle import rrdbnet_ref as ref
    rng = np.random.default_rng(5)
    cases = [(513, 512), (530, 600), (1024, 1024), (200, 1400), (276, 1000), (4096, 4096), (257, 1025)]
    cases += [(int(a), int(b)) for a, b in rng.integers(1, 1500, size=(40, 2))]
    for (h, w) in cases:
        for ts, tp in ((256, 10), (16, 2), (64, 0), (100, 7)):
            wins = native.plan_tiles(h, w, ts, tp, 4)
            exp = ref.tile_plan(h, w, ts, tp, 4)
            got = [((q.y1, q.y2, q.x1, q.x2), (q.crop_top, q.crop_bottom, q.crop_left, q.crop_right),
                    (q.oy1, q.oy2, q.ox1, q.ox2)) for q in wins]
            assert got == exp, (h, w, ts, tp)


def test_no_gpu_means_loud_failure():
    if native.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(native.S2srError):
        native.Engine(num_block=1)


def test_e4m3_encoder_matches_torch():
    """The weight packer's fp8 encoder (OCP e4m3fn, RNE, saturating) against torch's float8_e4m3fn
    on a sweep that covers subnormals, ties and every binade up to the finite maximum."""
    import torch

    lib = native.load_library()
    rng = np.random.default_rng(0)
    vals = np.concatenate([
        np.array([0.0, -0.0, 2.0 ** -10, 2.0 ** -9, 1.5 * 2.0 ** -9, 2.0 ** -6, 1.0625, 1.1875, 447.9, 448.0], np.float32),
        (rng.standard_normal(4000) * np.exp2(rng.integers(-12, 9, 4000))).astype(np.float32),
        np.ldexp(np.arange(8, 17, dtype=np.float32) + 0.5, -3),           # exact ties between neighbours
    ])
    vals = vals[np.abs(vals) <= 448.0]
    want = torch.from_numpy(vals).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = np.array([lib.s2sr_debug_f32_to_e4m3(float(v)) for v in vals], np.uint8)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, [(float(vals[i]), hex(got[i]), hex(want[i])) for i in bad[:5]]
    assert lib.s2sr_debug_f32_to_e4m3(1e9) == 0x7E and lib.s2sr_debug_f32_to_e4m3(-1e9) == 0xFE   # saturate, never NaN
    assert lib.s2sr_debug_f32_to_e4m3(float("nan")) == 0x7F
