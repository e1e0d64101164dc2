"""The iiwa14 closed-form IK (csrc/iiwa_ik.hip on the device; oracle/iiwa_ik_ref.py is its host restatement) against golden vectors produced
by the REFERENCE's own generated IKFast solver (tests/golden/make_golden_iiwa_ik.py; my_cpp/common.cpp:9-72)."""
import os

import numpy as np
import pytest

from oracle import iiwa_ik_ref as K

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'iiwa_ik_golden.npz'))


def _angdiff(a, b):
    d = a - b
    return np.abs(np.arctan2(np.sin(d), np.cos(d)))


def test_forward_kinematics_matches_the_reference_solver():
    assert np.abs(K.forward_kinematics(GOLD['fk_joints']) - GOLD['fk_poses']).max() < 1e-12


def test_solutions_and_limit_test_match_ikfast():
    poses = GOLD['poses32'].astype(np.float64)
    sols, valid = K.solve(poses)
    # every solution we call valid reproduces the pose (to the float32 rounding of the input pose)
    err = np.abs(K.forward_kinematics(sols) - poses[:, None]).max((-1, -2))
    assert np.median(err[valid]) < 1e-7 and err[valid].max() < 1e-4      # float32 pose rounding / sin q5 near the wrist band
    # where the reference returns the full set of 8, ours is the same set
    full = GOLD['nsol'] == 8
    assert full.sum() > 5000
    ref = GOLD['sols'][full]                                              # (n,8,7)
    d = _angdiff(ref[:, :, None, :], sols[full][:, None, :, :]).max(-1)   # (n, 8 ref, 8 ours)
    assert d.min(-1).max() < 1e-6 and valid[full].all()
    # no solution for the reference -> none for us
    none = GOLD['nsol'] == 0
    assert not valid[none].any()
    # the answer the filter consumes: any solution inside the joint limits
    ours = K.ik_within_limits(poses, GOLD['upper'], GOLD['lower'])
    agree = (ours == GOLD['within']).mean()
    assert agree >= 0.9995, agree              # the only known source of disagreement is the wrist-singularity band (see iiwa_ik.py)
    print('agreement with IKFast on', len(ours), 'poses:', agree)


def test_limits_and_degenerate_inputs():
    poses = GOLD['poses32'][:64].astype(np.float64)
    assert not K.ik_within_limits(poses, np.zeros(7) - 1.0, np.zeros(7) - 2.0).any()      # empty interval
    up = GOLD['upper'].copy(); lo = GOLD['lower'].copy()
    lo[2] = 0.1                                                                          # the fixed redundancy joint (0) violates its limit
    assert not K.ik_within_limits(poses, up, lo).any()
    assert K.ik_within_limits(np.zeros((0, 4, 4)), GOLD['upper'], GOLD['lower']).shape == (0,)
    far = np.eye(4)[None].repeat(3, 0); far[:, :3, 3] = [[3, 0, 0], [0, 0, 5], [0, 0, 0.36]]   # out of reach / on the base axis
    assert not K.ik_within_limits(far, GOLD['upper'], GOLD['lower']).any()


@pytest.mark.gpu
def test_device_kernel_matches_host_and_ikfast(cuda_device):
    import torch
    from catgrasp_amd import my_cpp
    ee = torch.from_numpy(GOLD['poses32']).to(cuda_device)
    ok = my_cpp.ik_within_limits_device(ee, GOLD['upper'], GOLD['lower']).cpu().numpy().astype(bool)
    host = K.ik_within_limits(GOLD['poses32'].astype(np.float64), GOLD['upper'], GOLD['lower'])
    assert np.array_equal(ok, host)
    assert (ok == GOLD['within']).mean() >= 0.9995
    lo = GOLD['lower'].copy(); lo[2] = 0.1
    assert not my_cpp.ik_within_limits_device(ee[:100], GOLD['upper'], lo).any().item()
    assert my_cpp.ik_within_limits_device(ee[:0], GOLD['upper'], GOLD['lower']).shape == (0,)
