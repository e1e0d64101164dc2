"""BASELINE.json configs[0] (C1) on the device, against the output of the REAL reference predicter run on the CPU.  (Its own module,
collected last: it was added after the round's GPU minutes were spent, so its first hardware run is the driver's.)"""
import numpy as np
import pytest

from catgrasp_amd import synth

pytestmark = pytest.mark.gpu


def test_c1_config_against_the_real_reference_predict_batch(cuda_device, mlp_precision):
    """BASELINE.json configs[0] (C1) at its stated size: one 'nut' instance, 2048-pt cloud, 256 grasp candidates.  The REAL
    GraspPredicter.predict_batch ran it on the CPU (tests/golden/predicter_golden_c1.npz, make_golden_predicter.py); the drop-in,
    called the same way under the same numpy seed, must return the same lists within 1e-4 and leave numpy's global generator in the
    same state (n_valid == n_pts: every pose draws a full permutation of the cloud)."""
    import os
    from catgrasp_amd.predicter import DEFAULT_GRASP_CFG, GraspPredicter
    p = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'predicter_golden_c1.npz'))
    gp = GraspPredicter('nut', cfg=DEFAULT_GRASP_CFG, state_dict=synth.make_state_dict('cls', 6, 10, seed=79), device=cuda_device)
    np.random.seed(456)
    ret = gp.predict_batch({'cloud_xyz': p['xyz'], 'cloud_normal': p['normal']}, list(p['poses']))
    assert np.array_equal(np.random.randint(0, 2 ** 31, 4), p['rng_after'])
    assert len(ret) == 256
    assert np.abs(np.array([r[2] for r in ret]) - p['grasp_probs']).max() <= 1e-4
    assert np.abs(np.array([float(r[1]) for r in ret]) - p['grasp_conf']).max() <= 1e-4
    srt = np.sort(p['grasp_probs'], axis=1)
    sure = (srt[:, -1] - srt[:, -2]) > 2e-4
    assert sure.mean() > 0.9 and np.array_equal(np.array([int(r[0]) for r in ret])[sure], p['grasp_labels'][sure])
