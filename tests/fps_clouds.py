"""Seeded clouds for the farthest-point-sampling goldens (tests/golden/make_golden_fps_large.py writes the reference's samples for them,
tests/test_oracle_golden.py and tests/test_primitives_gpu.py regenerate the same clouds): numpy Generator streams, stable across versions."""
import numpy as np


def cloud(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == 'uniform':                                   # a filled volume
        return (rng.random((n, 3)) * 0.1).astype(np.float32)
    if kind == 'surface':                                   # a corrugated tube, what a depth camera sees of an object
        t = rng.random((n, 2)); r = 0.03 + 0.01 * np.sin(12 * t[:, 0] * np.pi)
        return np.stack([r * np.cos(2 * np.pi * t[:, 0]), r * np.sin(2 * np.pi * t[:, 0]), 0.08 * t[:, 1]], -1).astype(np.float32)
    if kind == 'duplicates':                                # every point ~7 times, scattered over the indices: equal running distances
        base = rng.normal(0, 0.05, (n // 7, 3)).astype(np.float32)
        return base[rng.integers(0, len(base), n)]
    if kind == 'lattice':                                   # an integer lattice in random order: ties between distinct points
        lat = np.stack(np.meshgrid(np.arange(32), np.arange(32), np.arange(32), indexing='ij'), -1).reshape(-1, 3).astype(np.float32)
        lat = lat[rng.permutation(len(lat))[:n]] if n <= len(lat) else np.concatenate([lat, lat[rng.integers(0, len(lat), n - len(lat))]])
        return (lat * np.float32(0.01)).astype(np.float32)
    raise ValueError(kind)


CASES = [(kind, n, 1000 + 7 * i + j) for i, n in enumerate((3000, 9000, 20000, 24000)) for j, kind in enumerate(('uniform', 'surface', 'duplicates', 'lattice'))]
NPOINT = 96
