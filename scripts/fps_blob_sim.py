#!/usr/bin/env python
"""dev (host only, numpy): the blob-skipping farthest-point sampling of primitives.hip emulated at blob granularity --
  * exactness: the samples must equal the plain O(N * npoint) loop of pointnet2.py:54-75 (float32, (dx*dx + dy*dy) + dz*dz, first index
    on ties), for random clouds, surfaces, duplicated points and lattices;
  * cost model: per round, how many (wavefront, group) blobs are updated, and the maximum over the four SIMDs of the work their two
    wavefronts do (a round lasts as long as its slowest SIMD), in VALU instructions.
usage: fps_blob_sim.py [N] [npoint] [cells_per_axis] [group_slots]"""
import sys

import numpy as np

f32 = np.float32


def plain_fps(xyz, npoint, start):
    n = len(xyz)
    run = np.full(n, 1e10, f32)
    out = np.zeros(npoint, np.int64)
    far = start
    for it in range(npoint):
        out[it] = far
        d = xyz - xyz[far]
        d = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        run = np.minimum(run, d)
        far = int(np.argmax(run))
    return out


def part1by2(v):
    v = v.astype(np.uint32) & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v


def blob_fps(xyz, npoint, start, cells=16, gs=8, nt=512, stats=None):
    n = len(xyz)
    nw = nt // 64
    ppt = -(-n // nt); ppt += (-ppt) % gs
    ng = ppt // gs
    blob_pts = 64 * gs
    lo, hi = xyz.min(0), xyz.max(0)
    inv = f32(cells) / np.maximum(hi - lo, f32(1e-30))
    c = np.clip(((xyz - lo) * inv).astype(np.int32), 0, cells - 1)
    key = part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)
    perm = np.argsort(key, kind='stable')                       # the device's order inside a cell is arbitrary; any order is valid
    nblob = ng * nw
    cap = nblob * blob_pts
    P = np.zeros((cap, 3), f32); P[:n] = xyz[perm]
    idx = np.full(cap, 0xffff, np.int64); idx[:n] = perm
    real = np.arange(cap) < n
    run = np.where(real, f32(1e10), f32(0)).astype(f32)
    blo = np.full((nblob, 3), np.inf, f32); bhi = np.full((nblob, 3), -np.inf, f32)
    for j in range(nblob):
        s = slice(j * blob_pts, (j + 1) * blob_pts)
        if real[s].any():
            blo[j] = P[s][real[s]].min(0); bhi[j] = P[s][real[s]].max(0)
    bmax = np.array([run[j * blob_pts:(j + 1) * blob_pts].max() for j in range(nblob)], f32)
    out = np.zeros(npoint, np.int64)
    far = start
    cen = xyz[start]
    upd_total = 0; simd_max_total = 0
    for it in range(npoint):
        out[it] = far
        g = np.maximum(np.maximum(blo - cen, cen - bhi), f32(0))
        lb = (g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]) + g[:, 2] * g[:, 2]
        need = np.nonzero(lb < bmax)[0]
        for j in need:
            s = slice(j * blob_pts, (j + 1) * blob_pts)
            d = P[s] - cen
            d = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            run[s] = np.minimum(run[s], d)
            bmax[j] = run[s].max()
        # blob j = group (j // nw) of wavefront (j % nw); wavefronts w and w + 4 share a SIMD
        wave_upd = np.bincount(need % nw, minlength=nw)
        simd = wave_upd.reshape(-1, 4).sum(0) if nw % 4 == 0 else wave_upd
        upd_total += len(need); simd_max_total += simd.max()
        if stats is not None:
            # instruction-count model of a round with the winner search skipped in wavefronts that updated nothing:
            # idle wavefront 80, busy one 300 + 59 per updated group; a SIMD issues for its two wavefronts in turn
            for name, wave_of in (('round_robin', need % nw), ('contiguous', need // ng)):
                wu = np.bincount(wave_of, minlength=nw)
                cost = np.where(wu > 0, 300 + 59 * wu, 80)
                stats.setdefault('cost_' + name, []).append(cost.reshape(-1, 4).sum(0).max())
                stats.setdefault('busy_' + name, []).append((wu > 0).sum())
            wu = np.bincount(need % nw, minlength=nw)
            stats.setdefault('cost_now', []).append((300 + 59 * wu).reshape(-1, 4).sum(0).max())
        m = bmax.max()
        cands = []
        for j in np.nonzero(bmax == m)[0]:
            s = np.arange(j * blob_pts, (j + 1) * blob_pts)
            cands.append(s[run[s] == m])
        cands = np.concatenate(cands)
        w = cands[np.argmin(idx[cands])]
        far = int(idx[w]); cen = P[w]
    if stats is not None:
        stats.update(nblob=nblob, mean_updates=upd_total / npoint, frac=upd_total / npoint / nblob, mean_simd_max=simd_max_total / npoint,
                     ng=ng, gs=gs, nw=nw)
    return out


def clouds(n, rng):
    yield 'uniform cube', (rng.random((n, 3)) * 0.1).astype(f32)
    t = rng.random((n, 2)); r = 0.03 + 0.01 * np.sin(12 * t[:, 0] * np.pi)
    yield 'surface', np.stack([r * np.cos(2 * np.pi * t[:, 0]), r * np.sin(2 * np.pi * t[:, 0]), 0.08 * t[:, 1]], -1).astype(f32)
    base = rng.normal(0, 0.05, (n // 7, 3)).astype(f32)
    yield 'duplicates', base[rng.integers(0, len(base), n)]
    lat = np.stack(np.meshgrid(np.arange(32), np.arange(32), np.arange(32), indexing='ij'), -1).reshape(-1, 3).astype(f32)
    lat = lat[rng.permutation(len(lat))[:n]] if n <= len(lat) else np.concatenate([lat, lat[rng.integers(0, len(lat), n - len(lat))]])
    yield 'lattice', (lat * f32(0.01)).astype(f32)


if __name__ == '__main__':
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    cells = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    gs = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    rng = np.random.default_rng(0)
    for name, xyz in clouds(N, rng):
        st = {}
        s = S if name in ('uniform cube', 'surface') else min(S, 200)
        a = plain_fps(xyz, s, 3); b = blob_fps(xyz, s, 3, cells, gs, stats=st)
        # VALU instructions per SIMD and round: two skip tests + per updated group (5.5 per slot + wave max 11 + bookkeeping 4)
        per_group = 5.5 * gs + 15
        model = 2 * 17 + st['mean_simd_max'] * per_group
        full = 2 * (5.5 * st['ng'] * gs + 12)
        print('   model (instructions per SIMD and round): now %.0f, skip-idle round-robin %.0f (%.1f busy waves), skip-idle contiguous %.0f (%.1f busy waves)' % (
            np.mean(st['cost_now']), np.mean(st['cost_round_robin']), np.mean(st['busy_round_robin']), np.mean(st['cost_contiguous']), np.mean(st['busy_contiguous'])))
        print(f"{name:13s} equal={np.array_equal(a, b)}  blobs={st['nblob']} x {64 * gs}  updated/round={st['mean_updates']:.2f} ({st['frac']:.1%})  "
              f"busiest SIMD: {st['mean_simd_max']:.2f} groups  VALU/SIMD/round: {model:.0f} vs {full:.0f} now ({model / full:.2f})")
