#!/usr/bin/env python
"""The step's collision filter alone: the C3 batch's 16 segments (8 objects x {canonical grasps x 12 symmetries with nudging, cone poses})
as the ONE launch sequence score_slice issues (my_cpp.FilterPlan / cg_filter_grasp_pose_multi), HIP-event time per sequence, against the
round-5 form (one filter_on_device call per segment on 8 side streams).  CATGRASP_AMD_LIB=<other build> times an ablation build.
usage: python scripts/time_filter_plan.py [candidates=50000] [--pmc: 4 rounds of the sequence only, for a rocprofv3 --pmc pass]"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from catgrasp_amd import workload  # noqa: E402

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50000
dummy = types.SimpleNamespace(cfg={'n_pts': 2048})
b = workload.SceneBatch(dev, dummy, dummy, kind='nut', n_objects=8, pts_per_object=2500, per_replica=n, gripper_subdivisions=4)
rects = [(s, *r) for s, a, c in workload.intersect(b.segs, 0, b.n_total) for r in workload.split_eval_range(s.n_sym, a, c)]
key = ('time', 0, b.n_total)
codes, poses = b.run_filter_many(key, rects)
plan = b._plans[key]
g = b.gripper
if '--pmc' in sys.argv:
    for _ in range(4):
        plan.run(g['gripper_in_grasp'], True, keep_rejected_pose=True)
    torch.cuda.synchronize()
    print('pmc rounds 4', plan.E)
    sys.exit(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timed(lambda: plan.run(g['gripper_in_grasp'], True, keep_rejected_pose=True))


def per_segment():
    main = torch.cuda.current_stream()
    fork = torch.cuda.Event(); fork.record(main)
    for s, i0, i1, j0, j1 in rects:
        st = b._streams[s.obj]
        with torch.cuda.stream(st):
            st.wait_event(fork)
            b.run_filter(s, i0, i1, j0, j1)
    for st in b._streams:
        main.wait_stream(st)


ms_old = timed(per_segment)
c2 = torch.cat([b.run_filter(s, i0, i1, j0, j1)[0] for s, i0, i1, j0, j1 in rects])
print(f'lib={os.environ.get("CATGRASP_AMD_LIB", "in-tree")} evaluations={plan.E} segments={len(rects)}: one sequence {ms:.4f} ms '
      f'({plan.E / ms * 1e3 / 1e6:.1f} M evaluations/s); one call per segment on 8 streams {ms_old:.4f} ms; codes equal: {bool(torch.equal(codes, c2))} '
      f'histogram {torch.bincount(codes.long(), minlength=5).tolist()}')
