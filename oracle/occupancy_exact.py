"""TEST INFRASTRUCTURE ONLY -- an INDEPENDENT decision procedure for makeOccupancyGridFromCloudScan (my_cpp/common.cpp:324-431).

octomap is absent (PARITY UNPINNED against the library itself), so the ray cast of oracle/collision_ref.c and csrc/occupancy.hip -- a
restatement of octomap's incremental voxel walk (castRay: tMax / tDelta stepping) -- is checked here against geometry that shares
nothing with it: for every lattice query the occupied leaves the ray from the sensor origin passes through are found by the slab
method (ray vs axis-aligned box, float64, every occupied leaf tested on its own -- no stepping, no keys carried along), the first
of them by entry parameter is the leaf a correct walk must report, and the lattice point is "occupied" iff that leaf's centre is
within max_range and not farther than the query (common.cpp:393-404).  Queries whose decision hangs on a quantity inside a stated
epsilon band (a leaf the ray only grazes, |centre| ~ |query|, |centre| ~ max_range) are reported as undecidable rather than guessed.
"""
import numpy as np


def lattice(pts, resolution, pad=np.float32(0.005)):
    """The query lattice of common.cpp:366-383 in the reference's float32 arithmetic: (Q,3) float32 points, max_range (float32)."""
    pts = np.asarray(pts, dtype=np.float32)
    res = np.float32(resolution)
    mx, mn = pts.max(0), pts.min(0)
    n = [int((mx[a] + pad - (mn[a] - pad)) / res) for a in range(3)]
    ax = [(mn[a] - pad + np.arange(n[a], dtype=np.float32) * res).astype(np.float32) for a in range(3)]
    q = np.stack(np.meshgrid(ax[0], ax[1], ax[2], indexing='ij'), -1).reshape(-1, 3)
    max_range = np.float32(np.sqrt(float(mx[0] + pad) ** 2 + float(mx[1] + pad) ** 2 + float(mx[2] + pad) ** 2))
    return q, max_range


def occupied_leaves(pts, resolution):
    """Occupied depth-16 leaves after insertPointCloud: the leaves that contain a scan point (integer coordinates, floor(x/res))."""
    k = np.floor(np.asarray(pts, dtype=np.float32).astype(np.float64) * (1.0 / float(np.float32(resolution)))).astype(np.int64)
    return np.unique(k[(np.abs(k) < 32768).all(1)], axis=0)


def decide(pts, resolution, queries, max_range, graze_eps=1e-4, dist_eps=2e-6, chunk=2048):
    """-> (occupied (Q,) bool, decidable (Q,) bool).  graze_eps is relative to the leaf edge (a chord shorter than that is a graze)."""
    res = float(np.float32(resolution))
    leaves = occupied_leaves(pts, resolution)
    lo, hi = leaves * res, (leaves + 1) * res                     # (n,3) leaf boxes
    centre = (leaves + 0.5) * res
    cdist = np.sqrt((centre ** 2).sum(1))
    origin_occupied = bool((leaves == 0).all(1).any())
    q64 = np.asarray(queries, dtype=np.float64)
    qdist = np.sqrt((q64 ** 2).sum(1))
    occ = np.zeros(len(q64), dtype=bool)
    ok = np.ones(len(q64), dtype=bool)
    for s in range(0, len(q64), chunk):
        q = q64[s:s + chunk]
        d = q / np.maximum(qdist[s:s + chunk, None], 1e-300)      # (c,3) unit directions
        with np.errstate(divide='ignore', invalid='ignore'):
            inv = 1.0 / d[:, None, :]                              # (c,1,3)
            t1, t2 = lo[None] * inv, hi[None] * inv                # (c,n,3)
        tn, tf = np.minimum(t1, t2), np.maximum(t1, t2)
        zero = (d == 0.0)[:, None, :] & np.ones((1, len(leaves), 1), dtype=bool)
        inside0 = (lo[None] <= 0.0) & (hi[None] >= 0.0)            # a zero direction component: the slab holds the whole ray or none of it
        tn = np.where(zero, np.where(inside0, -np.inf, np.inf), tn)
        tf = np.where(zero, np.where(inside0, np.inf, -np.inf), tf)
        t_in, t_out = tn.max(2), tf.min(2)                         # (c,n)
        t_in0 = np.maximum(t_in, 0.0)
        chord = t_out - t_in0
        crossed = chord > graze_eps * res                          # definitely walked through
        grazed = (np.abs(chord) <= graze_eps * res)                # may or may not be visited
        first = np.where(crossed, t_in0, np.inf)
        j = first.argmin(1)
        hit = np.isfinite(first[np.arange(len(q)), j])
        t_hit = first[np.arange(len(q)), j]
        # an earlier (or equally early) grazed leaf could be reported instead: undecidable
        early_graze = (grazed & (t_in0 <= t_hit[:, None] + graze_eps * res)).any(1)
        # two crossed leaves entered at (numerically) the same parameter: the walk's order decides, not geometry
        second = np.partition(first, 1, axis=1)[:, 1] if first.shape[1] > 1 else np.full(len(q), np.inf)
        tie = hit & np.isfinite(second) & (second - np.where(hit, t_hit, 0.0) <= graze_eps * res)
        c = cdist[j]
        qd = qdist[s:s + chunk]
        o = hit & (c <= float(max_range)) & (c <= qd)
        band = hit & ((np.abs(c - qd) <= dist_eps) | (np.abs(c - float(max_range)) <= res))
        if origin_occupied:
            o[:] = True
        occ[s:s + chunk] = o
        ok[s:s + chunk] = ~(early_graze | tie | band) if not origin_occupied else True
    return occ, ok
