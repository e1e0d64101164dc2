"""TEST INFRASTRUCTURE ONLY (oracle) -- how much does makeOccupancyGridFromCloudScan depend on what could NOT be pinned to octomap?

my_cpp/common.cpp:384-402 calls octomap's OcTree::castRay for every point of a 1 mm lattice; octomap is not available here (PARITY
UNPINNED, oracle/collision_ref.c), so the ray walk of the oracle and of csrc/occupancy.hip is a restatement.  This study (the twin of
oracle/collision_sensitivity.py) re-runs the SAME lattice scan on the background clouds of all 8 objects of the C3
scene (BASELINE.json configs[2]; run_grasp_simulation.py:127-139: scene points within gripper_diameter / 2 of the object, minus the
object, one per 1 mm voxel) under the points where octomap builds / readings of castRay can differ (cr_set_occupancy_variant) and counts
the lattice points whose verdict flips against the parity oracle:

  tie_le       the voxel walk steps along the dimension with the smallest tMax; ties broken by `<=` instead of `<`
  range_after  the maxRange test on a new leaf comes after its occupancy test instead of before
  float_coords leaf centres in float arithmetic ((float)k + 0.5f) * res instead of double, rounded once (identical by construction: the
               double product of two floats is exact -- kept as a control)
  float_dir    the direction normalised with a float square root
  strict_dist  `dist < dist_query` instead of `<=` (counts the points that sit exactly on the boundary)
  all          tie_le + range_after + float_coords + float_dir together

    python -m oracle.occupancy_sensitivity [--out profiles/r5_occupancy_sensitivity.json]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from catgrasp_amd import synth                              # noqa: E402  (host-side scene generator only)
from oracle import collision_oracle as co                   # noqa: E402

VARIANTS = {'tie_le': (1, 0, 0, 0, 0), 'range_after': (0, 1, 0, 0, 0), 'float_coords': (0, 0, 1, 0, 0), 'float_dir': (0, 0, 0, 1, 0),
            'strict_dist': (0, 0, 0, 0, 1), 'all': (1, 1, 1, 1, 0)}


def set_variant(*v):
    co.lib().cr_set_occupancy_variant(*[ctypes.c_int(int(x)) for x in v])


def rows_set(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return set(map(bytes, a.view(np.dtype((np.void, 12))).reshape(-1))) if len(a) else set()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r5_occupancy_sensitivity.json'))
    args = ap.parse_args()
    objs = synth.make_scene(8, 2500, seed=0, kind='nut')
    g = synth.make_gripper()
    per_object, totals = [], {k: {'flipped': 0, 'gained': 0, 'lost': 0} for k in VARIANTS}
    base_total = 0
    t0 = time.time()
    try:
        for k in range(len(objs)):
            bg = synth.background_points(objs, k, g['diameter'])
            set_variant(0, 0, 0, 0, 0)
            base = rows_set(co.make_occupancy_grid(bg, 0.001))
            row = {'object': k, 'background_points': int(len(bg)), 'lattice_points_occupied': len(base)}
            base_total += len(base)
            for name, v in VARIANTS.items():
                set_variant(*v)
                alt = rows_set(co.make_occupancy_grid(bg, 0.001))
                gained, lost = len(alt - base), len(base - alt)
                row[name] = {'gained': gained, 'lost': lost}
                totals[name]['gained'] += gained; totals[name]['lost'] += lost; totals[name]['flipped'] += gained + lost
            per_object.append(row)
    finally:
        set_variant(0, 0, 0, 0, 0)
    out = {'what': 'lattice points of makeOccupancyGridFromCloudScan (1 mm) whose verdict differs from the parity oracle under alternative '
                   'readings of octomap::castRay, C3 scene (8 objects x 2,500 points), background clouds as run_grasp_simulation.py:127-139 forms them',
           'reference': 'my_cpp/common.cpp:384-402', 'occupied_lattice_points_baseline': base_total,
           'totals': {k: dict(v, fraction_of_occupied=round(v['flipped'] / max(base_total, 1), 8)) for k, v in totals.items()},
           'per_object': per_object, 'seconds': round(time.time() - t0, 1)}
    with open(args.out, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out['totals'], indent=1)); print('occupied baseline', base_total, 'written', args.out)


if __name__ == '__main__':
    main()
