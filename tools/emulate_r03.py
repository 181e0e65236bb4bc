#!/usr/bin/env python3
"""Round-3 numerics questions, answered on the CPU before any kernel is written.

(a) 1-D Winograd F(2,3) / F(4,3) along image ROWS (direct in x) for the 345 RDB convs: transformed weights
    U = G g (over dy) and transformed inputs V = B^T d (over 4 / 6 consecutive rows) rounded to fp16, fp32 accumulation
    over channels and the 3 kernel columns, output transform in fp32.  1-D because its input transform is reused by the
    3 kernel columns' MFMAs and needs no LDS re-layout.
(b) "x stays fp16, growth planes x1..x4 go e4m3" (per-output-channel power-of-two weight scales on the growth-plane
    weights, activation scale 2^g_exp): is max-abs <= 1e-3?

Trunk in fp32 and head/tail convs exact in every variant, so the differences are the RDB arithmetic alone.
Run from the repo root:  python tools/emulate_r03.py
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


def q8(t):
    return t.clamp(-448, 448).to(torch.float8_e4m3fn).float()


# F(2,3): points 0, 1, -1, inf
Bt2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
At2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
# F(4,3): points 0, +-1, +-2, inf (Lavin & Gray)
Bt4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=torch.float32)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                   [0, 0, 1]], dtype=torch.float32)
At4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32)


def conv_direct(x, w, b):
    return F.conv2d(h(x), h(w), b, padding=1)


def make_wino_rows(Bt, G, At, round_v=True):
    m = At.shape[0]          # outputs per tile
    a = Bt.shape[0]          # inputs per tile

    def conv(x, w, b):
        N, C, H, W = x.shape
        K = w.shape[0]
        assert H % m == 0
        U = h(torch.einsum('ij,kcjl->kcil', G, w))                        # [K, C, a, 3]: transformed over dy, direct in dx
        xp = F.pad(h(x), (1, 1, 1, 1 + (a - 3 - (m - 1))))                 # rows: 1 above, enough below for the last tile
        d = xp.unfold(2, a, m)                                            # [N, C, H/m, W+2, a]
        V = torch.einsum('ij,ncyxj->ncyxi', Bt, d)                        # [N, C, ty, W+2, a]
        if round_v:
            V = h(V)
        M = torch.zeros(N, K, d.shape[2], W, a)
        for dx in range(3):
            M += torch.einsum('kci,ncyxi->nkyxi', U[:, :, :, dx], V[:, :, :, dx:dx + W, :])
        Y = torch.einsum('ij,nkyxj->nkyix', At, M)                        # [N, K, ty, m, W]
        return Y.reshape(N, K, H, W) + b.view(1, -1, 1, 1)
    return conv


def run(x, sd, nb, conv, mixed=None):
    feat = F.conv2d(x * 255, sd['conv_first.weight'], None, padding=1) / 255 + sd['conv_first.bias'].view(1, -1, 1, 1)
    T = feat.clone(); R = feat.clone()
    cat = lambda *a: torch.cat(a, 1)
    for b in range(nb):
        for r in (1, 2, 3):
            p = f'body.{b}.rdb{r}'
            if mixed is None:
                cv = lambda t, n: conv(t, sd[p + n + '.weight'], sd[p + n + '.bias'])
            else:
                cv = lambda t, n: mixed(t, sd[p + n + '.weight'], sd[p + n + '.bias'])
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


def make_mixed(g_exp=5):
    """x (first 64 input channels) and its weights in fp16; growth channels and their weights in e4m3."""
    def conv(t, w, b):
        xs, gs = t[:, :64], t[:, 64:]
        y = F.conv2d(h(xs), h(w[:, :64]), b, padding=1)
        if gs.shape[1]:
            wg = w[:, 64:]
            m = wg.abs().amax(dim=(1, 2, 3)).clamp_min(1e-30)
            k = torch.floor(torch.log2(448.0 / m))
            wq = q8(wg * (2.0 ** k).view(-1, 1, 1, 1)) / (2.0 ** k).view(-1, 1, 1, 1)
            gq = q8(gs * 2.0 ** g_exp) / 2.0 ** g_exp
            y = y + F.conv2d(gq, wq, None, padding=1)
        return y
    return conv


if __name__ == '__main__':
    g = np.load('tests/golden/g4_full_nets.npz'); x = torch.from_numpy(g['x'])
    which = sys.argv[1:] or ['wino', 'mixed']
    for gain, key in ((0.3, 'y_b23'), (1.0, 'y_b23_gain1')):
        yref = torch.from_numpy(g[key]); sd = ref.to_torch_sd(synthetic_state_dict(23, seed=0, body_gain=gain))
        with torch.no_grad():
            yd = run(x, sd, 23, conv_direct)
            print(f"gain {gain}: direct fp16 operands: max-abs {(yd - yref).abs().max():.3e}", flush=True)
            if 'wino' in which:
                y2 = run(x, sd, 23, make_wino_rows(Bt2, G2, At2))
                print(f"gain {gain}: row-Winograd F(2,3), U and V fp16: {(y2 - yref).abs().max():.3e}", flush=True)
                y4 = run(x, sd, 23, make_wino_rows(Bt4, G4, At4))
                print(f"gain {gain}: row-Winograd F(4,3), U and V fp16: {(y4 - yref).abs().max():.3e}", flush=True)
            if 'mixed' in which:
                for ge in (4, 5):
                    ym = run(x, sd, 23, None, mixed=make_mixed(ge))
                    d = (ym - yref).abs()
                    print(f"gain {gain}: x fp16 + growth planes e4m3 (g_exp {ge}): max-abs {d.max():.3e} rms {d.pow(2).mean().sqrt():.3e}", flush=True)
