import sys, time
sys.path.insert(0, '.')
import torch
from oracle import pointnet_ref as oref
from catgrasp_amd import synth
sd = synth.make_state_dict('cls', 6, 10, seed=0)
x = torch.randn(200, 2048, 6) * 0.01
for nt in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    with torch.no_grad():
        oref.pointnet_cls_forward(sd, x[:20])
        t = time.time(); oref.pointnet_cls_forward(sd, x); dt = time.time() - t
    print(nt, f'{200/dt:.1f} cand/s', flush=True)
