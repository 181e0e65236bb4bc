"""GPU parity of the conv kernel (through the C ABI test hook s2sr_debug_conv) against a
plain torch fp32 conv of the same op.  Operands are pre-rounded to fp16 so that the only
difference left is fp32 accumulation order: tolerance 2e-4 * sum|a*b| scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from s2sr import native

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = native.Engine(num_block=1)
    yield e
    e.close()


def _ref(x, w, b, up, act):
    xt = torch.from_numpy(x)
    if up:
        xt = F.interpolate(xt, scale_factor=2, mode="nearest")
    y = F.conv2d(xt.double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)
    if act:
        y = F.leaky_relu(y, 0.2)
    return y.float().numpy()


def _h(a):
    return a.astype(np.float16).astype(np.float32)


CASES = [
    # N, Cin, Cout, H, W, up, act
    (1, 32, 32, 16, 32, False, False),
    (2, 3, 64, 20, 37, False, False),
    (1, 64, 32, 16, 32, False, True),
    (2, 96, 32, 33, 45, False, True),
    (1, 160, 32, 7, 100, False, True),
    (1, 192, 64, 17, 70, False, False),
    (2, 64, 64, 24, 24, True, True),
    (1, 64, 64, 19, 33, True, True),
    (1, 64, 3, 40, 40, False, False),
    (1, 128, 32, 65, 31, False, True),
]


@pytest.mark.parametrize("N,Cin,Cout,H,W,up,act", CASES)
def test_conv_random(eng, N, Cin, Cout, H, W, up, act):
    rng = np.random.default_rng(Cin * 1000 + Cout + H)
    x = _h(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
    w = _h((rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32) * 0.1
    y = eng.debug_conv(x, w, b, upsample=up, act=act)
    r = _ref(x, w, b, up, act)
    assert y.shape == r.shape
    err = np.abs(y - r).max()
    assert err <= 2e-4 * max(1.0, np.abs(r).max()), err


def test_conv_integer_layout(eng):
    """Exact-integer data with an asymmetric kernel: catches any swapped lane / tap / channel map."""
    rng = np.random.default_rng(3)
    N, Cin, Cout, H, W = 1, 64, 64, 18, 35
    x = rng.integers(-3, 4, size=(N, Cin, H, W)).astype(np.float32)
    w = rng.integers(-2, 3, size=(Cout, Cin, 3, 3)).astype(np.float32)
    b = rng.integers(-5, 6, size=Cout).astype(np.float32)
    y = eng.debug_conv(x, w, b)
    r = _ref(x, w, b, False, False)
    assert np.array_equal(y, r)
    # single-tap kernels: output must be the input shifted by exactly that tap
    for t in range(9):
        w1 = np.zeros((Cout, Cin, 3, 3), np.float32)
        for c in range(Cout):
            w1[c, c, t // 3, t % 3] = 1.0
        y1 = eng.debug_conv(x, w1, np.zeros(Cout, np.float32))
        assert np.array_equal(y1, _ref(x, w1, np.zeros(Cout, np.float32), False, False)), t
