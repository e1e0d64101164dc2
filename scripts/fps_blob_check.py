#!/usr/bin/env python
"""The blob-skipping farthest-point sampling (fps_blob_kernel, the default for 2,049 .. 24,576 points) against the round that updates
every point (fps_kernel, CATGRASP_AMD_FPS=plain; tests/test_primitives_gpu.py pins both to the oracle): identical samples on clouds of
every kind -- uniform volume, surface, duplicated points, lattice, one repeated point -- at every size class, two clouds per call with
different starts, and the time per round of each at 1 and 8 clouds per launch.  -> profiles/r4_fps_blob.json (argv[1])."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from catgrasp_amd import primitives   # noqa: E402
from fps_blob_sim import clouds        # noqa: E402

dev = torch.device('cuda:0')
VARIANTS = ('plain', 'blob')


def fps(pts, S, start, variant):
    if variant == 'plain':
        os.environ['CATGRASP_AMD_FPS'] = 'plain'
    else:
        os.environ.pop('CATGRASP_AMD_FPS', None)
    return primitives.farthest_point_sample(pts, S, start=start)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {'parity': [], 'time': []}
rng = np.random.default_rng(7)
bad = 0
for N, S in ((20000, 1024), (600, 200), (1500, 200), (2048, 200), (2049, 200), (3000, 200), (4096, 300), (4097, 300), (8192, 300), (8193, 300), (12000, 300), (12288, 300), (12289, 300), (16384, 300), (16385, 300), (20480, 300), (20481, 300), (22000, 300), (24576, 300)):
    for name, xyz in clouds(N, rng):
        pts = torch.from_numpy(np.stack([xyz, xyz[rng.permutation(N)]])).to(dev)
        start = torch.tensor([int(rng.integers(0, N)), N - 1], device=dev)
        ref = fps(pts, S, start, 'plain').cpu().numpy()
        row = {'N': N, 'S': S, 'cloud': name}
        for v in VARIANTS[1:]:
            t0 = time.time()
            got = fps(pts, S, start, v).cpu().numpy()
            same = bool(np.array_equal(got, ref))
            row[v] = same
            if not same:
                bad += 1
                w = np.argwhere(got != ref)
                row[v + '_first_mismatch'] = [int(w[0][0]), int(w[0][1]), int(got[tuple(w[0])]), int(ref[tuple(w[0])]), int(len(w))]
                row[v + '_range_ok'] = bool(got.min() >= 0 and got.max() < N)
            row[v + '_s'] = round(time.time() - t0, 3)
        res['parity'].append(row); print(row, flush=True)
for xyz in (np.full((9000, 3), np.float32(0.25)), np.concatenate([np.full((8999, 3), np.float32(0.25)), np.array([[1, 2, 3]], np.float32)])):
    pts = torch.from_numpy(xyz[None]).to(dev); start = torch.tensor([5], device=dev)
    ref = fps(pts, 20, start, 'plain').cpu().numpy()
    row = {'cloud': 'degenerate', 'ref': ref[0, :4].tolist()}
    for v in VARIANTS[1:]:
        row[v] = bool(np.array_equal(fps(pts, 20, start, v).cpu().numpy(), ref)); bad += not row[v]
    res['parity'].append(row); print(row, flush=True)
g = torch.Generator(device=dev); g.manual_seed(0)
for N, S in ((20000, 1024), (1024, 512), (2048, 1024), (2049, 1024), (4096, 1024), (8192, 1024), (12288, 1024), (16384, 1024), (24576, 1024)):
    sets = {'uniform cube': (torch.rand(8, N, 3, device=dev, generator=g) * 0.1).contiguous()}
    surf = [c for n, c in clouds(N, np.random.default_rng(3)) if n == 'surface'][0]
    sets['surface'] = torch.from_numpy(np.stack([surf[np.random.default_rng(i).permutation(N)] for i in range(8)])).to(dev)
    for name, pts8 in sets.items():
        for B in (1, 8):
            pts = pts8[:B].contiguous(); start = torch.zeros(B, dtype=torch.long, device=dev)
            row = {'N': N, 'S': S, 'cloud': name, 'clouds': B}
            for v in VARIANTS:
                ms = timed(lambda: fps(pts, S, start, v))
                row[v + '_ms'] = round(ms, 4); row[v + '_us_per_round'] = round(ms / S * 1e3, 3)
            res['time'].append(row); print(row, flush=True)
res['mismatching_cases'] = bad
print('MISMATCHES:', bad)
if len(sys.argv) > 1:
    with open(sys.argv[1], 'w') as f:
        json.dump(res, f, indent=1)
