"""Per-layer GPU parity of the TRUNK kernels (conv_trunk.hip: conv_trunk_f16 / conv_trunk_f8, and the row-Winograd form in
conv_wino.hip) through the C ABI test hook s2sr_debug_conv_trunk, against fp64 torch convs of the same operands.

These kernels carry the 345 RDB convs (reference server/app/cnn_super_resolution.py:78-91, 103-107) = 84 % of a step;
tests/test_gpu_conv.py goes through conv3x3.hip and never touches them.  Operands are generated representable in the
kernel's operand format (fp16, or e4m3 at the handle's scales), so what is left is accumulation order and the rounding of
the stored result, both written into the bounds below.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from s2sr import native

pytestmark = pytest.mark.gpu

K14, K5, K5R, F14, F5, F5R = (native.TRUNK_F16_CONV14, native.TRUNK_F16_CONV5, native.TRUNK_F16_CONV5_RRDB,
                              native.TRUNK_F8_CONV14, native.TRUNK_F8_CONV5, native.TRUNK_F8_CONV5_RRDB)


@pytest.fixture(scope="module")
def eng():
    e = native.Engine(num_block=1, precision=native.PREC_F16_HP)
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng8():
    e = native.Engine(num_block=1, precision=native.PREC_FP8)
    yield e
    e.close()


def _h(a):
    return a.astype(np.float16).astype(np.float32)


def _conv64(x, w, b):
    return F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1).numpy()


def _lrelu(v):
    return np.where(v >= 0, v, 0.2 * v)


def _run14(e, x, w, b, form):
    if form in (3, 4) and not native.experimental():
        pytest.skip("row-Winograd / loader-wave forms: experimental library only (make EXP=1, S2SR_LIB=.../libs2sr_exp.so)")
    return e.debug_conv_trunk(K14, x, w, b, form=form)


def _rand(rng, N, Cin, Cout, H, W):
    x = _h(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
    w = _h((rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32))
    b = (rng.standard_normal(Cout) * 0.1).astype(np.float32)
    return x, w, b


# form 1 = 16x32 patches / 5-deep ring (single tiles), form 2 = 32x32 patches / 3-deep ring (batches), form 3 = row-Winograd,
# form 4 = form 2 with a fifth, load-only wave, form 5 = 8x32 patches / 7-deep ring (one tile on 256 CUs)
# form 10 (r04) = 8x32 patches with TWO 16-channel planes per pipeline stage (half the barriers per patch: the single-tile form)
@pytest.mark.parametrize("form", [1, 2, 3, 4, 5, 10])
@pytest.mark.parametrize("Cin,N,H,W", [(64, 1, 16, 32), (96, 2, 33, 45), (128, 1, 65, 31), (160, 1, 7, 100), (160, 2, 40, 64)])
def test_f16_conv14_random(eng, form, Cin, N, H, W):
    rng = np.random.default_rng(Cin * 100 + H + form)
    x, w, b = _rand(rng, N, Cin, 32, H, W)
    y = _run14(eng, x, w, b, form)
    r = _lrelu(_conv64(x, w, b))
    # the plane is stored as fp16 (half an ulp = 2^-11 relative); accumulation order 2e-4 of the scale.  The Winograd form
    # rounds its transformed operands to fp16 once more: its own bound is measured in test_wino_* below.
    tol = (2e-4 if form != 3 else 1.5e-3) * max(1.0, np.abs(r).max()) + np.abs(r) * 2.0 ** -11
    err = np.abs(y - r)
    assert np.all(err <= tol), (float(err.max()), float(np.abs(r).max()))


@pytest.mark.parametrize("form", [1, 2, 3, 4, 5, 10])
def test_f16_conv14_integer_layout(eng, form):
    """Exact data: every cout reads ONE input channel through ONE tap (the channel map is a permutation with a stride, the
    tap varies with the cout), so any swapped lane / tap / channel / row shows up as a wrong integer.  LeakyReLU runs in fp32
    (max(v, 0.2f * v)) and the result is stored as fp16: compared bit for bit."""
    rng = np.random.default_rng(7 + form)
    for Cin, N, H, W in ((64, 1, 18, 35), (160, 1, 34, 66), (128, 2, 9, 33)):
        x = rng.integers(-6, 7, size=(N, Cin, H, W)).astype(np.float32)
        w = np.zeros((32, Cin, 3, 3), np.float32)
        for c in range(32):
            w[c, (c * 5 + 3) % Cin, ((c * 7) % 9) // 3, ((c * 7) % 9) % 3] = 1.0 if c % 3 else -2.0
        b = rng.integers(-5, 6, size=32).astype(np.float32)
        y = _run14(eng, x, w, b, form)
        v = _conv64(x, w, b).astype(np.float32)
        r = np.maximum(v, np.float32(0.2) * v).astype(np.float16).astype(np.float32)
        assert np.array_equal(y, r), (Cin, form, float(np.abs(y - r).max()))
    # every tap on its own, identity channel map: the output is the input shifted by exactly that tap (zero padding)
    Cin, H, W = 96, 20, 40
    x = np.abs(rng.integers(0, 9, size=(1, Cin, H, W))).astype(np.float32)
    for t in range(9):
        w = np.zeros((32, Cin, 3, 3), np.float32)
        for c in range(32):
            w[c, 64 + c, t // 3, t % 3] = 1.0
        y = _run14(eng, x, w, np.zeros(32, np.float32), form)
        assert np.array_equal(y, _conv64(x, w, np.zeros(32, np.float32)).astype(np.float32)), (t, form)


@pytest.mark.parametrize("form,base", [(6, 2), (7, 1), (8, 5), (9, 2), (11, 2)])
def test_f16_conv14_whole_patch_forms(eng, form, base):
    """Form 9 (r04): form 6 with the weights fetched from global memory into AGPRs instead of through the LDS ring (WGL) -- the same
    MFMAs in the same order.  Forms 6 / 7 / 8: the 32x32 / 16x32 / 8x32 conv1-4 kernels compiled for launches without ragged edges or mosaic
    separators (conv_trunk_f16<..., FULL>: no px_live arithmetic, no trash-line selects in the epilogue) -- what 256x256 tile
    batches run.  Same bytes as the generic form on 32-multiple shapes, the exact-integer layout check, and a refusal
    (not a wrong answer) when the shape is ragged."""
    if form in (9, 11) and not native.experimental():
        pytest.skip("weights-from-global / 64x32-patch forms: experimental library only")
    rng = np.random.default_rng(600 + form)
    for Cin, N, H, W in ((64, 1, 32, 32), (96, 2, 64, 96), (160, 1, 96, 64), (128, 3, 32, 160)):
        if form == 11:              # 64x32 patches (16 rows per wave, accumulators split over both register files, double-buffered ring)
            H = 64 * ((H + 63) // 64)
        x, w, b = _rand(rng, N, Cin, 32, H, W)
        y = _run14(eng, x, w, b, form)
        assert np.array_equal(y, _run14(eng, x, w, b, base)), (form, Cin, H, W)
        r = _lrelu(_conv64(x, w, b))
        assert np.all(np.abs(y - r) <= 2e-4 * max(1.0, np.abs(r).max()) + np.abs(r) * 2.0 ** -11)
    Cin, H, W = 160, 64, 64
    x = rng.integers(-6, 7, size=(1, Cin, H, W)).astype(np.float32)
    w = np.zeros((32, Cin, 3, 3), np.float32)
    for c in range(32):
        w[c, (c * 5 + 3) % Cin, ((c * 7) % 9) // 3, ((c * 7) % 9) % 3] = 1.0 if c % 3 else -2.0
    b = rng.integers(-5, 6, size=32).astype(np.float32)
    v = _conv64(x, w, b).astype(np.float32)
    assert np.array_equal(_run14(eng, x, w, b, form), np.maximum(v, np.float32(0.2) * v).astype(np.float16).astype(np.float32))
    x, w, b = _rand(rng, 1, 64, 32, 33, 45)
    with pytest.raises(native.S2srError):
        eng.debug_conv_trunk(K14, x, w, b, form=form)


# conv5 patch forms: 1 = 16x32 patches / 4-deep ring (batches), 5 = 8x32 patches / 5-deep ring (single tiles, r04: one 256x256
# tile gives 256 patches instead of 128; its A fragments are fetched two per step)
@pytest.mark.parametrize("form", [1, 5, 10])
@pytest.mark.parametrize("N,H,W", [(1, 16, 32), (1, 17, 70), (2, 33, 33)])
def test_f16_conv5_random(eng, N, H, W, form):
    """conv5: v = 0.2 * (conv + b) + (x + lo), written as the (fp16 hi, e4m3(lo * 2^lo_exp)) pair; rdb3's conv5 adds
    0.2 * v + skip with the skip read as such a pair."""
    rng = np.random.default_rng(H * 10 + W)
    x, w, b = _rand(rng, N, 192, 64, H, W)
    le = eng.debug_config()["lo_exp"]
    lo = (rng.integers(-7, 8, size=(N, 64, H, W)) * 2.0 ** (-4 - le)).astype(np.float32)       # exact in e4m3 at 2^lo_exp
    conv = _conv64(x, w, b)
    v1 = 0.2 * conv + (x[:, :64].astype(np.float64) + lo)
    y1 = eng.debug_conv_trunk(K5, x, w, b, lo=lo, form=form)

    def bound(v):      # accumulation order + the 4-bit lo half: |v - (hi + lo8)| <= 2^-11 |v| * 2^-4, + e4m3 subnormal step 2^-9-le
        return 2e-4 * max(1.0, np.abs(v).max()) + np.abs(v) * 2.0 ** -15 + 2.0 ** (-9 - le)
    assert np.all(np.abs(y1 - v1) <= bound(v1)), float(np.abs(y1 - v1).max())
    # the pair is BETTER than fp16 alone: rounding the result to fp16 would leave up to 2^-11 |v|
    assert np.abs(y1 - v1).max() < 0.25 * np.abs(v1.astype(np.float32).astype(np.float16).astype(np.float64) - v1).max() + 2e-4
    shi = _h(rng.uniform(0.5, 4.0, size=(N, 64, H, W)).astype(np.float32) * rng.choice([-1.0, 1.0], size=(N, 64, H, W)).astype(np.float32))
    slo = (rng.integers(-7, 8, size=(N, 64, H, W)) * 2.0 ** (-4 - le)).astype(np.float32)
    skip = (shi.astype(np.float64) + slo).astype(np.float32)
    assert np.array_equal(skip.astype(np.float64), shi.astype(np.float64) + slo)                # exactly representable as a pair
    y2 = eng.debug_conv_trunk(K5R, x, w, b, lo=lo, skip=skip, form=form)
    if form in (5, 10):    # every patch form accumulates in the same order: the same bytes
        assert np.array_equal(y1, eng.debug_conv_trunk(K5, x, w, b, lo=lo, form=1))
        assert np.array_equal(y2, eng.debug_conv_trunk(K5R, x, w, b, lo=lo, skip=skip, form=1))
    v2 = 0.2 * v1 + skip.astype(np.float64)
    assert np.all(np.abs(y2 - v2) <= bound(v2)), float(np.abs(y2 - v2).max())


@pytest.mark.parametrize("form", [1, 5, 10])
def test_f16_conv5_integer_layout(eng, form):
    """Single-tap kernels with power-of-two data: 0.2 * acc is rounded once in fp32 (the same for oracle and kernel up to
    the fused form), so compare against the fp32 formula with 1 ulp of slack on the 0.2 product and exact channel maps."""
    rng = np.random.default_rng(11)
    N, H, W = 1, 19, 37
    x = rng.integers(-4, 5, size=(N, 192, H, W)).astype(np.float32)
    for t in (0, 4, 8, 5):
        w = np.zeros((64, 192, 3, 3), np.float32)
        for c in range(64):
            w[c, (c * 11 + 5) % 192, t // 3, t % 3] = 5.0          # 0.2 * 5 * integer: integers up to fp32 rounding of 0.2f
        b = np.zeros(64, np.float32)
        y = eng.debug_conv_trunk(K5, x, w, b, form=form)
        v = 0.2 * _conv64(x, w, b) + x[:, :64]
        assert np.abs(y - v).max() <= 1e-5, (t, float(np.abs(y - v).max()))


# ---- fp8 trunk ------------------------------------------------------------------------------------------------------
def _e4m3_vals(rng, shape, exp):
    """Random values exactly representable as e4m3(v * 2^exp), |v * 2^exp| <= 8 (every e4m3 code up to 8.0 occurs)."""
    b = rng.integers(0, 256, size=shape, dtype=np.uint8)
    b = np.where((b & 0x7F) == 0x7F, 0, b).astype(np.uint8)                 # no NaN codes
    v = torch.from_numpy(b).view(torch.float8_e4m3fn).float().numpy()
    v = np.clip(v, -8.0, 8.0)                                               # 8.0 is a code: still representable
    return (v * 2.0 ** -exp).astype(np.float32)


def _q8(t):
    return t.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def _quant_w(w):
    """The fp8 trunk's weight quantisation (pack_conv_weights_f8 / pack.hip): per-cout power-of-two scale into e4m3's top binade."""
    wt = torch.from_numpy(w)
    m = wt.abs().amax(dim=(1, 2, 3))
    k = torch.floor(torch.log2(448.0 / m))
    k = torch.where(m * 2.0 ** k >= 448.0, k - 1, k)
    k = torch.where(m * 2.0 ** (k + 1) < 448.0, k + 1, k)
    s = (2.0 ** k).view(-1, 1, 1, 1)
    return (_q8(wt * s) / s).numpy()


def _e4m3_ulp(v):
    a = np.maximum(np.abs(v), 2.0 ** -6)
    return np.maximum(2.0 ** (np.floor(np.log2(a)) - 3), 2.0 ** -9)


@pytest.mark.parametrize("form", [0, 1, 3, 5, 8])       # loader wave (default) | four waves | + all streamed | + all resident | two waves per SIMD
@pytest.mark.parametrize("Cin,N,H,W", [(64, 1, 16, 32), (96, 1, 33, 45), (160, 2, 20, 70), (128, 1, 65, 31)])
def test_f8_conv14_random(eng8, form, Cin, N, H, W):
    if form != 0 and not native.experimental():
        pytest.skip("non-default fp8 conv1-4 forms: experimental library only")
    cfg = eng8.debug_config()
    xe, ge = cfg["fp8_x_exp"], cfg["fp8_g_exp"]
    rng = np.random.default_rng(Cin + H + form)
    x = np.concatenate([_e4m3_vals(rng, (N, 64, H, W), xe)] + ([_e4m3_vals(rng, (N, Cin - 64, H, W), ge)] if Cin > 64 else []), axis=1)
    w = (rng.standard_normal((32, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = (rng.standard_normal(32) * 0.1).astype(np.float32)
    y = eng8.debug_conv_trunk(F14, x, w, b, form=form)
    r = _lrelu(_conv64(x, _quant_w(w), b))
    sc = 2.0 ** ge
    # stored as e4m3(r * 2^g_exp): within half an e4m3 ulp of the unrounded value (+ accumulation order), saturating at 448
    rs = np.clip(r * sc, -448, 448)
    assert np.all(np.abs(y * sc - rs) <= 0.5 * _e4m3_ulp(rs) * 1.001 + 2e-4 * max(1.0, np.abs(rs).max())), float(np.abs(y * sc - rs).max())


@pytest.mark.parametrize("N,H,W", [(1, 16, 32), (2, 21, 70)])
def test_f8_conv5_random(eng8, N, H, W):
    cfg = eng8.debug_config()
    xe, ge = cfg["fp8_x_exp"], cfg["fp8_g_exp"]
    rng = np.random.default_rng(H + W)
    x = np.concatenate([_e4m3_vals(rng, (N, 64, H, W), xe), _e4m3_vals(rng, (N, 128, H, W), ge)], axis=1)
    w = (rng.standard_normal((64, 192, 3, 3)) / np.sqrt(9 * 192)).astype(np.float32)
    b = (rng.standard_normal(64) * 0.1).astype(np.float32)
    v1 = 0.2 * _conv64(x, _quant_w(w), b) + x[:, :64]
    y, aux = eng8.debug_conv_trunk(F5, x, w, b)
    tol16 = 2e-4 * max(1.0, np.abs(v1).max()) + np.abs(v1) * 2.0 ** -11
    assert np.all(np.abs(y - v1) <= tol16), float(np.abs(y - v1).max())
    sx = 2.0 ** xe
    vs = np.clip(v1 * sx, -448, 448)
    assert np.all(np.abs(aux * sx - vs) <= 0.5 * _e4m3_ulp(vs) * 1.001 + 2e-4 * max(1.0, np.abs(vs).max()))
    skip = _h(rng.standard_normal((N, 64, H, W)).astype(np.float32))
    y2, _ = eng8.debug_conv_trunk(F5R, x, w, b, skip=skip)
    v2 = 0.2 * v1 + skip
    assert np.all(np.abs(y2 - v2) <= 2e-4 * max(1.0, np.abs(v2).max()) + np.abs(v2) * 2.0 ** -11), float(np.abs(y2 - v2).max())


@pytest.mark.experimental
def test_f8_conv14_forms_agree_bit_for_bit(eng8):
    """The fp8 conv1-4 forms accumulate the same products in the same order: same bytes, whichever form is named."""
    rng = np.random.default_rng(3)
    cfg = eng8.debug_config()
    for Cin in (64, 96, 160):
        x = np.concatenate([_e4m3_vals(rng, (2, 64, 40, 70), cfg["fp8_x_exp"])] +
                           ([_e4m3_vals(rng, (2, Cin - 64, 40, 70), cfg["fp8_g_exp"])] if Cin > 64 else []), axis=1)
        w = (rng.standard_normal((32, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
        b = (rng.standard_normal(32) * 0.1).astype(np.float32)
        y0 = eng8.debug_conv_trunk(F14, x, w, b, form=0)
        for form in (1, 3, 5, 8):
            assert np.array_equal(eng8.debug_conv_trunk(F14, x, w, b, form=form), y0), (Cin, form)


def test_f16_patch_forms_agree_bit_for_bit(eng):
    """16x32 and 32x32 patch forms accumulate in the same order: a tile gives the same bytes alone and inside a batch."""
    rng = np.random.default_rng(4)
    for Cin in (64, 160):
        x, w, b = _rand(rng, 2, Cin, 32, 40, 70)
        y1 = eng.debug_conv_trunk(K14, x, w, b, form=1)
        assert np.array_equal(y1, eng.debug_conv_trunk(K14, x, w, b, form=2)), Cin
        if native.experimental():
            assert np.array_equal(y1, eng.debug_conv_trunk(K14, x, w, b, form=4)), Cin  # the loader-wave form: same stream of MFMAs
        assert np.array_equal(y1, eng.debug_conv_trunk(K14, x, w, b, form=5)), Cin      # 8x32 patches
        assert np.array_equal(y1, eng.debug_conv_trunk(K14, x, w, b, form=10)), Cin     # 8x32 patches, two planes per stage


@pytest.mark.experimental
def test_f16_conv5_lo_encoding_forms_agree(eng):
    """the shipped conv5 epilogue encodes the trunk's lo half in a short form (v_fma_mix_f32 for v - fp16(v), the
    2^lo_exp inside v_cvt_scalef32_pk_fp8_f32); the long form (subtract, multiply, clamp, v_cvt_pk_fp8_f32) is form 2 of the
    experimental library.  Same bytes out -- hi blocks and lo planes -- on ordinary operands and on stress operands whose lo
    values reach the clamp (|lo| up to 448 * 2^-lo_exp) and the e4m3 subnormal range."""
    rng = np.random.default_rng(77)
    le = eng.debug_config()["lo_exp"]
    for scale in (1.0, 40.0, 1e-3):
        N, H, W = 2, 33, 70
        x, w, b = _rand(rng, N, 192, 64, H, W)
        x = _h(x * np.float32(scale))
        lo = (rng.integers(-7, 8, size=(N, 64, H, W)) * 2.0 ** (-4 - le)).astype(np.float32)
        skip = _h(rng.standard_normal((N, 64, H, W)).astype(np.float32) * np.float32(scale))
        for kind, kw in ((K5, {}), (K5R, {"skip": skip})):
            a = eng.debug_conv_trunk(kind, x, w, b, lo=lo, form=1, **kw)
            c = eng.debug_conv_trunk(kind, x, w, b, lo=lo, form=2, **kw)
            assert np.array_equal(a, c), (scale, kind, float(np.abs(a - c).max()))


@pytest.mark.parametrize("form", [5, 10])
def test_f16_small_forms_with_several_patches_per_workgroup(eng, form):
    """The single-tile forms (8x32 patches; form 10: two planes per stage, conv5 on a double-buffered ring) on launches with MORE
    patches than workgroups -- 432 patches on 256 workgroups: the patch boundary inside a workgroup (the first stage of a second
    patch waits behind the previous epilogue's stores) is a path the single-patch shapes above never take.  r04: the double
    buffer's wait allowed the epilogue's stores to stand in for the awaited stage's DMA pieces.  Same bytes as the 16x32 forms."""
    rng = np.random.default_rng(900 + form)
    N, H, W = 6, 96, 192
    x, w, b = _rand(rng, N, 160, 32, H, W)
    assert np.array_equal(eng.debug_conv_trunk(K14, x, w, b, form=form), eng.debug_conv_trunk(K14, x, w, b, form=1))
    x, w, b = _rand(rng, N, 192, 64, H, W)
    le = eng.debug_config()["lo_exp"]
    lo = (rng.integers(-7, 8, size=(N, 64, H, W)) * 2.0 ** (-4 - le)).astype(np.float32)
    skip = _h(rng.standard_normal((N, 64, H, W)).astype(np.float32))
    assert np.array_equal(eng.debug_conv_trunk(K5, x, w, b, lo=lo, form=form), eng.debug_conv_trunk(K5, x, w, b, lo=lo, form=1))
    assert np.array_equal(eng.debug_conv_trunk(K5R, x, w, b, lo=lo, skip=skip, form=form), eng.debug_conv_trunk(K5R, x, w, b, lo=lo, skip=skip, form=1))
