#!/usr/bin/env python
"""Host only, numpy: fps_blob_kernel of csrc/fps.hip emulated lane by lane -- the sorted-position layout, the per-lane group maxima, the
box test (radius=True: against the cloud's largest running distance, what the kernel ships; radius=False: against each blob's own
maximum, the first design), the tie detection (two lanes / two groups / two slots at the maximum), the slow path through `perm`, the
wave records and the second-stage tie break -- against the plain loop of pointnet2.py:54-75.  The scatter order inside a cell is
randomised (the device's atomics give an arbitrary one).  tests/test_fps_blob_model_cpu.py runs it against the oracle.
usage: fps_blob_lane_sim.py [GS] [radius]"""
import sys

import numpy as np

from fps_blob_sim import plain_fps, clouds, part1by2

f32 = np.float32
NONE = 0x7fffffff


def kernel(xyz, npoint, start, NT=512, PPT=40, GS=8, rng=None, count=None, radius=False):
    N = len(xyz)
    NW, NG, CAP = NT // 64, PPT // GS, NT * PPT
    assert N <= CAP
    # prologue: counting sort by Morton cell, arbitrary order inside a cell
    lo, hi = xyz.min(0), xyz.max(0)
    inv = f32(16) / np.maximum(hi - lo, f32(1e-30))
    c = np.minimum(np.maximum((xyz - lo) * inv, f32(0)), f32(15)).astype(np.uint32)
    key = part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)
    order = rng.permutation(N)
    order = order[np.argsort(key[order], kind='stable')]
    perm = np.full(CAP, 0xffff, np.int64); perm[:N] = order
    w_, l_, k_ = np.meshgrid(np.arange(NW), np.arange(64), np.arange(PPT), indexing='ij')
    pos = ((((k_ // GS) * NW + w_) * GS + (k_ % GS)) << 6) + l_          # fps_slot_pos
    p = perm[pos]
    real = p != 0xffff
    P = np.where(real[..., None], xyz[np.where(real, p, 0)], f32(0)).astype(f32)      # (NW,64,PPT,3)
    dist = np.where(real, f32(1e10), f32(0)).astype(f32)
    gm = dist.reshape(NW, 64, NG, GS).max(3)                               # gmax[g] per lane
    bm = gm.max(1)                                                          # (NW,NG) wave-uniform
    Pg = P.reshape(NW, 64 * 1, NG, GS, 3)
    rg = real.reshape(NW, 64, NG, GS)
    blo = np.where(rg[..., None], Pg, np.inf).min((1, 3)).astype(f32)      # (NW,NG,3)
    bhi = np.where(rg[..., None], Pg, -np.inf).max((1, 3)).astype(f32)
    out = np.zeros(npoint, np.int64)
    out[0] = start
    cen = xyz[start]
    slow = 0
    rad = f32(1e10)
    for it in range(1, npoint):
        rec_v = np.zeros(16, f32); rec_p = np.zeros((16, 3), f32); rec_i = np.full(16, NONE, np.int64)
        with np.errstate(invalid='ignore'):
            q = np.maximum(np.maximum(blo - cen, cen - bhi), f32(0))
        lb = (q[..., 0] * q[..., 0] + q[..., 1] * q[..., 1]) + q[..., 2] * q[..., 2]
        need = (lb < rad) if radius else (lb < bm)
        for w in range(NW):
            for g in range(NG):
                if need[w, g]:
                    s = slice(g * GS, (g + 1) * GS)
                    d = P[w, :, s] - cen
                    d = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
                    dist[w, :, s] = np.minimum(dist[w, :, s], d)
                    gm[w, :, g] = dist[w, :, s].max(1)
                    bm[w, g] = gm[w, :, g].max()
            if radius:
                bv = gm[w].max(1)                                            # per lane
                wmax = bv.max()
                cand = bv == wmax
                wl = int(np.argmax(cand))
                eqg = gm[w] == bv[:, None]                                   # (64,NG)
                gi = np.argmax(eqg, 1); cg = eqg.sum(1)
                gw = int(gi[wl])
                eq = dist[w, :, gw * GS:(gw + 1) * GS] == wmax
                k = np.where(eq.any(1), gw * GS + np.argmax(eq, 1), gw * GS + GS - 1)
                cnt = eq.sum(1)
                kw = int(k[wl])
                tie = (int(cand.sum()) - 1) | (int(cg[wl]) - 1) | (int(cnt[wl]) - 1)
                cntg = 1
                slow_groups = (gm[w] == wmax).any(0)
            else:
                wmax = bm[w].max()
                eqg = bm[w] == wmax
                gw = int(np.argmax(eqg)); cntg = int(eqg.sum())
                cand = gm[w, :, gw] == wmax
                wl = int(np.argmax(cand))
                eq = dist[w, :, gw * GS:(gw + 1) * GS] == wmax                 # (64,GS)
                k = np.where(eq.any(1), gw * GS + np.argmax(eq, 1), gw * GS + GS - 1)
                cnt = eq.sum(1)
                kw = int(k[wl])
                tie = (int(cand.sum()) - 1) | (int(cnt[wl]) - 1)
                slow_groups = bm[w] == wmax
            if tie | (cntg - 1):
                slow += 1
                bi = np.full(64, NONE, np.int64); bk = np.zeros(64, np.int64)
                for g in range(NG):
                    if slow_groups[g]:
                        for j in range(GS):
                            oi = perm[pos[w, :, g * GS + j]]
                            better = (dist[w, :, g * GS + j] == wmax) & (oi < bi)
                            bi = np.where(better, oi, bi); bk = np.where(better, g * GS + j, bk)
                mi = bi.min()
                wl = int(np.argmax(bi == mi)); kw = int(bk[wl])
            rec_v[w] = wmax; rec_p[w] = P[w, wl, kw]; rec_i[w] = pos[w, wl, kw]
        best = rec_v.max()
        c16 = rec_v == best
        if c16.sum() != 1:
            oi = np.where(rec_i < CAP, perm[np.minimum(rec_i, CAP - 1)], NONE)
            mi = np.where(c16, oi, NONE).min()
            c16 = c16 & (oi == mi)
        win = int(np.argmax(c16))
        out[it] = rec_i[win]; cen = rec_p[win].copy(); rad = best
    out[1:] = perm[out[1:]]
    if count is not None:
        count['slow'] = slow
    return out


if __name__ == '__main__':
    GS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    RADIUS = len(sys.argv) > 2 and sys.argv[2] == 'radius'
    rng = np.random.default_rng(1)
    ok = True
    for N, PPT, S in ((20000, 40, 96), (8193, 40, 64), (22000, 48, 64), (24576, 48, 40), (20480, 40, 40)):
        for name, xyz in clouds(N, rng):
            cnt = {}
            st = int(rng.integers(0, N))
            a = plain_fps(xyz, S, st); b = kernel(xyz, S, st, PPT=PPT, GS=GS, rng=rng, count=cnt, radius=RADIUS)
            same = np.array_equal(a, b); ok &= same
            print(f'N={N:6d} PPT={PPT} GS={GS} {name:13s} equal={same}  slow-path wave-rounds={cnt["slow"]} of {(S - 1) * 8}', flush=True)
    # a cloud of identical points, and one point repeated with a single outlier
    for xyz in (np.full((9000, 3), f32(0.25)), np.concatenate([np.full((8999, 3), f32(0.25)), np.array([[1, 2, 3]], f32)])):
        a = plain_fps(xyz, 20, 5); b = kernel(xyz, 20, 5, PPT=40, GS=GS, rng=rng, radius=RADIUS)
        same = np.array_equal(a, b); ok &= same
        print('degenerate cloud equal=', same, a[:6], b[:6])
    print('ALL EQUAL' if ok else 'MISMATCH')
