import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, 'sentinel2-super-resolution-poc_amd')
from s2sr import native
e = native.Engine(num_block=1, precision=native.PREC_F16_HP)
rng = np.random.default_rng(0)
def h(a): return a.astype(np.float16).astype(np.float32)
def ref(x, w, b):
    v = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1).numpy()
    return np.where(v >= 0, v, 0.2 * v)
for (N, Cin, H, W) in [(1, 64, 16, 32), (3, 64, 256, 256)]:
    x = h(rng.standard_normal((N, Cin, H, W)).astype(np.float32))
    w = h((rng.standard_normal((32, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32))
    b = (rng.standard_normal(32) * 0.1).astype(np.float32)
    z = np.zeros_like(w)
    y = e.debug_conv_trunk(0, x, z, b, form=3)
    print(N, Cin, H, W, "zero weights: y[0,:4,0,0]", y[0, :4, 0, 0], "lrelu(b)", np.where(b >= 0, b, 0.2 * b)[:4], "max dev", np.abs(y - ref(x, z, b)).max())
    y = e.debug_conv_trunk(0, x, w, np.zeros(32, np.float32), form=3)
    r = ref(x, w, np.zeros(32, np.float32))
    d = np.abs(y - r)
    print("  zero bias: max err", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "|r|max", np.abs(r).max())
    for t in range(9):
        w1 = np.zeros_like(w)
        for c in range(32): w1[c, c, t // 3, t % 3] = 1.0
        xi = np.abs(rng.integers(0, 9, size=x.shape)).astype(np.float32)
        y = e.debug_conv_trunk(0, xi, w1, np.zeros(32, np.float32), form=3)
        r = ref(xi, w1, np.zeros(32, np.float32))
        print("  tap", t, "exact" if np.array_equal(y, r) else f"WRONG max {np.abs(y-r).max()} first bad {np.argwhere(y != r)[:3].tolist()}")
