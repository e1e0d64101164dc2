"""Generate tests/golden/fps_large_golden.npz: the REAL reference's farthest_point_sample (/root/reference/pointnet2.py:54-75, imported
read-only with empty cv2/torchvision stubs) on the seeded clouds of tests/fps_clouds.py -- 3,000 to 24,000 points (every geometry of the
blob-skipping kernel), filled volume / surface / duplicated points / lattice -- so that the HIP kernels are compared with the reference's
own samples at their working sizes, not only with the oracle.  Only the start indices and the samples are stored (the clouds are
regenerated from their seeds).  Run in the build container only (`python tests/golden/make_golden_fps_large.py`)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
for m in ('cv2', 'torchvision'):
    sys.modules.setdefault(m, types.ModuleType(m))
sys.path.insert(0, '/root/reference')
import pointnet2 as ref  # noqa: E402  (the reference implementation itself)

import fps_clouds  # noqa: E402

torch.set_num_threads(1)
out = {}
for kind, n, seed in fps_clouds.CASES:
    xyz = torch.from_numpy(fps_clouds.cloud(kind, n, seed)[None])
    torch.manual_seed(seed)
    start = torch.randint(0, n, (1,), dtype=torch.long)            # what pointnet2.py:66 draws under this seed
    torch.manual_seed(seed)
    idx = ref.farthest_point_sample(xyz, fps_clouds.NPOINT)
    assert int(idx[0, 0]) == int(start[0])
    out[f'{kind}_{n}_start'] = start.numpy()
    out[f'{kind}_{n}_fps'] = idx.numpy().astype(np.int32)
    print(kind, n, idx[0, :6].tolist(), flush=True)
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'fps_large_golden.npz'), **out)
