import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from catgrasp_amd import aligning, synth
from oracle import aligning_ref
rng = np.random.default_rng(0)
n = 8192
nocs = rng.uniform(-0.5, 0.5, (n, 3)); R = synth.random_rotation(rng); s = np.array([0.016, 0.02, 0.007]); t = np.array([0.02, -0.03, 0.62])
obs = nocs @ (R @ np.diag(s)).T + t + rng.normal(0, 1e-4, (n, 3))
bad = rng.random(n) < 0.2; nocs[bad] = rng.uniform(-0.5, 0.5, (bad.sum(), 3))
ids = np.stack([rng.choice(n, 4, replace=False) for _ in range(10000)]).astype(np.int32)
kw = dict(max_scale=[0.05] * 3, min_scale=[0.005, 0.005, 0.001], max_dimensions=np.array([1.2] * 3))
aligning.estimate9DTransform(nocs, obs, 0.003, ids=ids[:100], **kw); torch.cuda.synchronize()
t0 = time.time(); T, inl = aligning.estimate9DTransform(nocs, obs, 0.003, ids=ids, **kw); torch.cuda.synchronize(); dt = time.time() - t0
print(f'device RANSAC: 10000 hypotheses x {n} pts: {dt*1e3:.1f} ms, inliers {len(inl)}')
t0 = time.time(); Tr, ir, _ = aligning_ref.estimate9DTransform(nocs, obs, 0.003, ids[:300], kw['max_scale'], kw['min_scale'], kw['max_dimensions']); dt = time.time() - t0
print(f'numpy oracle: 300 hypotheses: {dt*1e3:.1f} ms -> {dt/300*10000:.2f} s per 10000')
