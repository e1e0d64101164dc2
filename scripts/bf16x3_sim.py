"""Numerical experiment: error of split-bf16 (bf16x3 / bf16x2 / plain bf16) emulation of the conv3 (128->1024) layers."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from catgrasp_amd import synth
from oracle import pointnet_ref as oref

def bf(x): return x.to(torch.bfloat16).to(torch.float32)
def split(x):
    hi = bf(x); lo = bf(x - hi); return hi, lo
def mm_split(w, x, terms):
    # w:(O,K) x:(B,K,N)
    wh, wl = split(w); xh, xl = split(x)
    y = torch.matmul(wh, xh)
    if terms >= 3:
        y = y + torch.matmul(wh, xl) + torch.matmul(wl, xh)
    if terms >= 4:
        y = y + torch.matmul(wl, xl)
    return y

orig = oref._conv_bn
def make_patched(terms, layers):
    def conv_bn(x, sd, conv, bn, relu):
        if any(conv.endswith(l) for l in layers) and terms > 0:
            w = sd[conv + '.weight'][:, :, 0]
            # fold BN scale into w like the device path
            s = sd[bn + '.weight'] / torch.sqrt(sd[bn + '.running_var'] + 1e-5)
            wf = w * s[:, None]
            bfold = (sd[conv + '.bias'] - sd[bn + '.running_mean']) * s + sd[bn + '.bias']
            y = mm_split(wf, x, terms) + bfold.view(1, -1, 1)
            return torch.relu(y) if relu else y
        return orig(x, sd, conv, bn, relu)
    return conv_bn

rng = np.random.default_rng(5)
for gain in (1.0, 1.6):
    sd = synth.make_state_dict('cls', 6, 10, seed=11, gain=gain)
    x = torch.from_numpy(rng.normal(0, 0.5, (8, 2048, 6)).astype(np.float32))
    oref._conv_bn = orig
    y32, _ = oref.pointnet_cls_forward(sd, x); y64, _ = oref.pointnet_cls_forward(sd, x, torch.float64)
    p32 = torch.softmax(y32, 1)
    print(f'gain {gain}: |logit| max {y32.abs().max():.3f}; fp32 vs fp64 {float((y32-y64).abs().max()):.2e}')
    for terms, layers, name in [(3, ['conv3'], 'bf16x3 conv3'), (3, ['conv3', 'conv2'], 'bf16x3 conv3+conv2'),
                                (1, ['conv3'], 'bf16 conv3'), (4, ['conv3'], 'bf16x4 conv3')]:
        oref._conv_bn = make_patched(terms, layers)
        y, _ = oref.pointnet_cls_forward(sd, x)
        e = (y - y64).abs().max().item(); pe = (torch.softmax(y, 1) - p32).abs().max().item()
        print(f'   {name:22s} logits err vs f64 {e:.2e}   probs err vs f32 {pe:.2e}')
oref._conv_bn = orig
