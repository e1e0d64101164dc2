"""Golden vectors for the building blocks of the reference's pointnet2 module used on their own, with the constructor arguments the
live pipeline does NOT use: the REAL /root/reference/pointnet2.py classes in eval mode on CPU --
STN3d(channel=3|5), STNkd(k=64|20), PointNetEncoder(global_feat x feature_transform x channel=3|4|6) incl. the reference DEFAULTS
(feature_transform=False, channel=3: pointnet2.py:227), PointNetCls(3,10), PointNetSeg(4,30), square_distance with C = 5 and 1.
Build container only.  Weights are not stored: catgrasp_amd.synth.seeded_like(module.state_dict(), seed) regenerates them.

    python tests/golden/make_golden_blocks.py        ->  tests/golden/pointnet2_blocks_golden.npz
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
for m in ('cv2', 'torchvision'):
    sys.modules.setdefault(m, types.ModuleType(m))
sys.path.insert(0, '/root/reference')
import pointnet2 as ref  # noqa: E402

from catgrasp_amd import synth  # noqa: E402

torch.set_num_threads(1)
rng = np.random.default_rng(4242)
out = {}
B, N = 3, 150


def run(tag, model, x, seed):
    model.load_state_dict(synth.seeded_like(model.state_dict(), seed))
    model.eval()
    with torch.no_grad():
        y = model(torch.from_numpy(x))
    out[tag + '_x'] = x
    out[tag + '_seed'] = np.array([seed])
    ys = y if isinstance(y, tuple) else (y,)
    for i, t in enumerate(ys):
        v = np.zeros((0,), np.float32) if t is None else t.numpy()
        if v.ndim == 3 and v.shape[1] == 1088:      # PointNetEncoder(global_feat=False): (B,1088,N) -- keep a strided sample (rows of both parts)
            v = v[:, ::9, ::4].copy()
        if v.ndim == 3 and v.shape[1:] == (64, 64):  # 64 x 64 transforms: strided sample
            v = v[:, ::3, ::3].copy()
        out[f'{tag}_y{i}'] = v


for ch in (3, 5):
    run(f'stn3d_c{ch}', ref.STN3d(ch), rng.normal(0, 0.5, (B, ch, N)).astype(np.float32), 200 + ch)
for k in (64, 20):
    run(f'stnkd_k{k}', ref.STNkd(k=k), rng.normal(0, 0.5, (B, k, N)).astype(np.float32), 300 + k)
for gf in (True, False):
    for ft in (False, True):
        for ch in (3, 4, 6):
            run(f'enc_g{int(gf)}_f{int(ft)}_c{ch}', ref.PointNetEncoder(global_feat=gf, feature_transform=ft, channel=ch),
                rng.normal(0, 0.5, (B, ch, N)).astype(np.float32), 400 + 10 * ch + 2 * gf + ft)
run('cls_c3', ref.PointNetCls(3, 10), rng.normal(0, 0.5, (B, N, 3)).astype(np.float32), 501)
run('seg_c4', ref.PointNetSeg(4, 30), rng.normal(0, 0.5, (B, N, 4)).astype(np.float32), 502)
for C in (5, 1, 3):
    a = rng.normal(0, 1, (2, 17, C)).astype(np.float32); b = rng.normal(0, 1, (2, 300, C)).astype(np.float32)
    out[f'sqd_c{C}_a'] = a; out[f'sqd_c{C}_b'] = b
    out[f'sqd_c{C}'] = ref.square_distance(torch.from_numpy(a), torch.from_numpy(b)).numpy()
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pointnet2_blocks_golden.npz')
np.savez_compressed(path, **out)
print('wrote', path, os.path.getsize(path), 'bytes;', len(out), 'arrays')
