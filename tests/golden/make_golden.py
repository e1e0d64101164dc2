"""Generate tests/golden/pointnet2_golden.npz by running the REAL reference (/root/reference/pointnet2.py,
imported read-only with empty cv2/torchvision stubs) on seeded inputs.  Run in the build container only
(`python tests/golden/make_golden.py`); the GPU box has no /root/reference and uses the committed file.

Weights are not stored (PointNetCls alone is 13.9 MB): they are regenerated from
catgrasp_amd.synth.make_state_dict(kind, seed, gain) which is deterministic (numpy default_rng)."""
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
import pointnet2 as ref  # noqa: E402  (the reference implementation itself)

from catgrasp_amd import synth  # noqa: E402

out = {}
rng = np.random.default_rng(2024)
torch.set_num_threads(1)

# --- models -------------------------------------------------------------------------------------
for kind, n_out, seed, gain in (('cls', 10, 101, 1.6), ('cls', 10, 102, 1.0), ('seg', 300, 103, 1.6)):
    sd = synth.make_state_dict(kind, 6, n_out, seed=seed, gain=gain)
    model = (ref.PointNetCls if kind == 'cls' else ref.PointNetSeg)(6, n_out)
    model.load_state_dict(sd)
    model.eval()
    x = rng.normal(0, 0.5, (3, 160, 6)).astype(np.float32)
    with torch.no_grad():
        y, tf = model(torch.from_numpy(x))
    tag = f'{kind}_{seed}'
    out[tag + '_x'] = x
    out[tag + '_y'] = y.numpy() if kind == 'cls' else y.numpy()[:, ::3, ::4].copy()   # strided sample of (B,N,300)
    out[tag + '_tf'] = tf.numpy()[:, ::7, ::5].copy()      # strided sample of the (B,64,64) feature transform
    out[tag + '_meta'] = np.array([seed, gain, n_out], dtype=np.float64)

# --- PointNet++ primitives ----------------------------------------------------------------------
xyz = (rng.normal(0, 0.05, (2, 700, 3)) + np.array([0, 0, 0.6])).astype(np.float32)
feats = rng.normal(size=(2, 700, 4)).astype(np.float32)
txyz = torch.from_numpy(xyz)
out['prim_xyz'] = xyz
out['prim_feats'] = feats
out['sqdist'] = ref.square_distance(txyz[:, :16], txyz).numpy()
idx = torch.from_numpy(rng.integers(0, 700, (2, 9, 3)))
out['index_idx'] = idx.numpy()
out['index_out'] = ref.index_points(torch.from_numpy(feats), idx).numpy()
torch.manual_seed(77)
out['fps'] = ref.farthest_point_sample(txyz, 48).numpy()           # start drawn by torch.randint under seed 77
new_xyz = ref.index_points(txyz, torch.from_numpy(out['fps']))
out['ball'] = ref.query_ball_point(0.03, 16, txyz, new_xyz).numpy()
torch.manual_seed(78)
nx, npnts, gx, fi = ref.sample_and_group(32, 0.04, 8, txyz, torch.from_numpy(feats), returnfps=True)
out['sg_new_xyz'] = nx.numpy(); out['sg_new_points'] = npnts.numpy(); out['sg_grouped_xyz'] = gx.numpy(); out['sg_fps'] = fi.numpy()
ax, ap = ref.sample_and_group_all(txyz, torch.from_numpy(feats))
out['sga_new_xyz'] = ax.numpy(); out['sga_new_points_sum'] = np.array([ap.numpy().astype(np.float64).sum()])

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pointnet2_golden.npz')
np.savez_compressed(path, **out)
print('wrote', path, os.path.getsize(path), 'bytes')
