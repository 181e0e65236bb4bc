#!/usr/bin/env python3
"""Would Winograd F(2x2, 3x3) on fp16 MFMA operands keep the RDB trunk inside the HP tolerance?  CPU emulation: the 345 RDB
convs either direct (fp16-rounded operands, fp32 accumulation -- what conv_trunk_f16 does) or as Winograd with the
transformed weights U = G g G^T and transformed input tiles V = B^T d B rounded to fp16 before the (fp32-accumulated)
channel sums, output transform in fp32.  Trunk in fp32 and head/tail convs exact in both, so the difference is the trunk
arithmetic alone.  Run from the repo root:  python tools/emulate_winograd.py
"""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, 'sentinel2-super-resolution-poc_amd'); sys.path.insert(0, '.')
from s2sr.weights import synthetic_state_dict
from oracle import rrdbnet_ref as ref
torch.set_num_threads(8)
h = lambda t: t.half().float()
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def conv_direct(x, w, b):
    return F.conv2d(h(x), h(w), b, padding=1)


def conv_winograd(x, w, b, round_v=True):
    N, C, H, W = x.shape
    K = w.shape[0]
    assert H % 2 == 0 and W % 2 == 0
    if N > 1:
        return torch.cat([conv_winograd(x[i:i + 1], w, b, round_v) for i in range(N)], 0)
    U = torch.einsum('ij,kcjl,ml->kcim', G, w, G)                      # [K, C, 4, 4]
    U = h(U)
    xp = F.pad(h(x), (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)[0]                            # [C, H/2, W/2, 4, 4]
    V = torch.einsum('ij,cyxjl,ml->cyxim', Bt, d, Bt)                   # [C, ty, tx, 4, 4], fp32 of fp16 inputs (exact sums)
    if round_v:
        V = h(V)
    M = torch.einsum('kcim,cyxim->kyxim', U, V)                          # fp32 accumulation over channels
    Y = torch.einsum('ij,kyxjl,ml->kyxim', At, M, At)                   # [K, ty, tx, 2, 2]
    y = Y.permute(0, 1, 3, 2, 4).reshape(1, K, H, W)
    return y + b.view(1, -1, 1, 1)


def run(x, sd, nb, conv):
    feat = F.conv2d(x * 255, sd['conv_first.weight'], None, padding=1) / 255 + sd['conv_first.bias'].view(1, -1, 1, 1)
    T = feat.clone(); R = feat.clone()
    cat = lambda *a: torch.cat(a, 1)
    for b in range(nb):
        for r in (1, 2, 3):
            p = f'body.{b}.rdb{r}'
            cv = lambda t, n: conv(t, sd[p + n + '.weight'], sd[p + n + '.bias'])
            xin = T
            x1 = F.leaky_relu(cv(xin, '.conv1'), 0.2); x2 = F.leaky_relu(cv(cat(xin, x1), '.conv2'), 0.2)
            x3 = F.leaky_relu(cv(cat(xin, x1, x2), '.conv3'), 0.2); x4 = F.leaky_relu(cv(cat(xin, x1, x2, x3), '.conv4'), 0.2)
            T = cv(cat(xin, x1, x2, x3, x4), '.conv5') * 0.2 + T
            if r == 3:
                T = T * 0.2 + R; R = T
    ex = lambda t, n: F.conv2d(t, sd[n + '.weight'], sd[n + '.bias'], padding=1)
    lr = lambda t: F.leaky_relu(t, 0.2)
    feat = feat + ex(T, 'conv_body')
    feat = lr(ex(F.interpolate(feat, scale_factor=2, mode='nearest'), 'conv_up1'))
    feat = lr(ex(F.interpolate(feat, scale_factor=2, mode='nearest'), 'conv_up2'))
    return ex(lr(ex(feat, 'conv_hr')), 'conv_last')


g = np.load('tests/golden/g4_full_nets.npz'); x = torch.from_numpy(g['x'])
for gain, key in ((0.3, 'y_b23'), (1.0, 'y_b23_gain1')):
    yref = torch.from_numpy(g[key]); sd = ref.to_torch_sd(synthetic_state_dict(23, seed=0, body_gain=gain))
    with torch.no_grad():
        yd = run(x, sd, 23, conv_direct)
        yw = run(x, sd, 23, conv_winograd)
        yw32 = run(x, sd, 23, lambda a, w, b: conv_winograd(a, w, b, round_v=False))
    print(f"gain {gain}: direct fp16 operands max-abs err {(yd - yref).abs().max():.3e};  Winograd F(2,3), U and V in fp16: {(yw - yref).abs().max():.3e};"
          f"  U fp16, V unrounded: {(yw32 - yref).abs().max():.3e}", flush=True)
