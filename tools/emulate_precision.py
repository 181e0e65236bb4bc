#!/usr/bin/env python3
"""CPU emulation of the kernel arithmetic (fp16-rounded conv operands, fp32 accumulation, fp32
trunk) against the golden fp32 reference outputs: shows which convs the fp16 error comes from.
Selecting convs as "exact" stands for the split-operand mode of S2SR_PREC_F16_HP.
Run from the repo root:  python tools/emulate_precision.py
"""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0,'sentinel2-super-resolution-poc_amd'); sys.path.insert(0,'.')
from s2sr.weights import synthetic_state_dict
from oracle import rrdbnet_ref as ref
torch.set_num_threads(8)
h = lambda t: t.half().float()
I = lambda t: t
def run(x, sd, nb, exact):   # exact: set of conv names computed with unrounded operands
    def conv(t, n):
        r = I if n in exact else h
        return F.conv2d(r(t), r(sd[n+'.weight']), sd[n+'.bias'], padding=1)
    r0 = I if 'conv_first' in exact else h
    feat = F.conv2d(r0(x*255), r0(sd['conv_first.weight']), None, padding=1)/255 + sd['conv_first.bias'].view(1,-1,1,1)
    T = feat.clone(); R = feat.clone()
    for b in range(nb):
        for r in (1,2,3):
            p = f'body.{b}.rdb{r}'; xin = h(T); cat = lambda *a: torch.cat(a,1)
            cv = lambda t,n: F.conv2d(h(t), h(sd[p+n+'.weight']), sd[p+n+'.bias'], padding=1)
            x1 = F.leaky_relu(cv(xin,'.conv1'),0.2); x2 = F.leaky_relu(cv(cat(xin,x1),'.conv2'),0.2)
            x3 = F.leaky_relu(cv(cat(xin,x1,x2),'.conv3'),0.2); x4 = F.leaky_relu(cv(cat(xin,x1,x2,x3),'.conv4'),0.2)
            T = cv(cat(xin,x1,x2,x3,x4),'.conv5')*0.2 + T
            if r==3: T = T*0.2 + R; R = T
    lr = lambda t: F.leaky_relu(t,0.2)
    feat = feat + conv(T, 'conv_body')
    feat = lr(conv(F.interpolate(feat, scale_factor=2, mode='nearest'), 'conv_up1'))
    feat = lr(conv(F.interpolate(feat, scale_factor=2, mode='nearest'), 'conv_up2'))
    feat = lr(conv(feat, 'conv_hr'))
    return conv(feat, 'conv_last')
g = np.load('tests/golden/g4_full_nets.npz'); x = torch.from_numpy(g['x'])
for gain,key in ((0.3,'y_b23'),(1.0,'y_b23_gain1')):
    yref = torch.from_numpy(g[key]); sd = ref.to_torch_sd(synthetic_state_dict(23, seed=0, body_gain=gain))
    with torch.no_grad():
        for ex in ([], ['conv_last'], ['conv_hr','conv_last'], ['conv_up2','conv_hr','conv_last'], ['conv_up1','conv_up2','conv_hr','conv_last'],
                   ['conv_first','conv_body','conv_up1','conv_up2','conv_hr','conv_last']):
            y = run(x, sd, 23, set(ex)); print(f"gain {gain} exact={ex}: max-abs err {(y-yref).abs().max():.3e}")
