"""TEST INFRASTRUCTURE ONLY (oracle): host restatement (numpy, float64, vectorised over poses) of the algorithm of
catgrasp_amd/csrc/iiwa_ik.hip -- the product runs the HIP kernel, never this file.

Closed-form inverse kinematics of the KUKA LBR iiwa14 with the redundancy joint (index 2) fixed at 0 -- the solver
`get_ik_within_limits` (my_cpp/common.cpp:9-72) obtains from its generated IKFast file (Transform6D, joints 0,1,3,4,5,6
solved, free joint 2 value-initialised to 0 at :15).  Written from the arm's DH description

      T_i = Rz(q_i) Tz(d_i) Rx(alpha_i),   alpha = (-,+,+,-,-,+,0) pi/2,   d = (0.36, 0, 0.42, 0, 0.4, 0, 0.081)

(checked against the reference's own forward kinematics in tests/test_iiwa_ik.py):
  * wrist centre   W = p - 0.081 R[:,2];  with q2 = 0 shoulder, elbow and wrist lie in the vertical plane through the base axis:
        W = (c0 r, s0 r, 0.36 + h),  r = 0.42 sin q1 + 0.4 sin(q1 - q3),  h = 0.42 cos q1 + 0.4 cos(q1 - q3)
    -> two base branches (q0, q0 + pi with r -> -r) x two elbow branches (q3 = +-acos((r^2 + h^2 - a^2 - b^2) / 2ab));
  * wrist          R_4^7 = Rz(q4) Ry(q5) Rz(q6)  -> two branches (sign of sin q5);
  up to 8 solutions, angles in (-pi, pi].
Degeneracy windows of the generated solver that change the ANSWER are reproduced (measured against the real solver through
oracle/_ref, tests/test_iiwa_ik.py): no solution when the wrist centre is within 1 mm of the base axis (rho^2 < 1e-6: the
solver's shoulder-singularity branch returns nothing), the elbow equation accepts |cos q3| <= 1 + 1e-7, and wrist solutions with
|sin q5| < 2e-3 are dropped (the generated code drops most of them between 5e-4 and 3e-3 depending on the other wrist
angles -- that band is the only place the two can disagree: 1e-4 of random poses).
Pinned to the reference: tests/golden/iiwa_ik_golden.npz holds the answers of the reference's own generated solver."""
import numpy as np

D_BS, D_SE, D_EW, D_WF = 0.36, 0.42, 0.4, 0.081
_H = np.pi / 2
ALPHA = (-_H, _H, _H, -_H, -_H, _H, 0.0)
DLINK = (D_BS, 0.0, D_SE, 0.0, D_EW, 0.0, D_WF)
RHO2_MIN = 1e-6              # wrist centre closer than 1 mm to the base axis: the reference solver returns no solution
C3_TOL = 1e-7                # IKFAST_SINCOS_THRESH on the elbow cosine
SINGULAR_EPS = 2e-3          # |sin q5| below this: wrist solutions dropped (see above)


def forward_kinematics(q):
    """q (...,7) -> (...,4,4) flange pose in the base frame."""
    q = np.asarray(q, dtype=np.float64)
    T = np.broadcast_to(np.eye(4), q.shape[:-1] + (4, 4)).copy()
    for i in range(7):
        c, s = np.cos(q[..., i]), np.sin(q[..., i])
        ca, sa = np.cos(ALPHA[i]), np.sin(ALPHA[i])
        A = np.zeros(q.shape[:-1] + (4, 4))
        A[..., 0, 0] = c; A[..., 0, 1] = -s * ca; A[..., 0, 2] = s * sa
        A[..., 1, 0] = s; A[..., 1, 1] = c * ca; A[..., 1, 2] = -c * sa
        A[..., 2, 1] = sa; A[..., 2, 2] = ca; A[..., 2, 3] = DLINK[i]
        A[..., 3, 3] = 1.0
        T = T @ A
    return T


def _r04(q0, q1, q3):
    """rotation of frame 4 (after the elbow joint) for q2 = 0: the product of the first four link rotations."""
    q = np.stack([q0, q1, np.zeros_like(q0), q3], -1)
    R = np.broadcast_to(np.eye(3), q.shape[:-1] + (3, 3)).copy()
    for i in range(4):
        c, s = np.cos(q[..., i]), np.sin(q[..., i])
        ca, sa = np.cos(ALPHA[i]), np.sin(ALPHA[i])
        A = np.zeros(q.shape[:-1] + (3, 3))
        A[..., 0, 0] = c; A[..., 0, 1] = -s * ca; A[..., 0, 2] = s * sa
        A[..., 1, 0] = s; A[..., 1, 1] = c * ca; A[..., 1, 2] = -c * sa
        A[..., 2, 1] = sa; A[..., 2, 2] = ca
        R = R @ A
    return R


def solve(ee_in_base):
    """ee_in_base (E,4,4) -> (solutions (E,8,7) float64, valid (E,8) bool)."""
    T = np.asarray(ee_in_base, dtype=np.float64).reshape(-1, 4, 4)
    E = T.shape[0]
    R, p = T[:, :3, :3], T[:, :3, 3]
    W = p - D_WF * R[:, :, 2]
    rho0 = np.hypot(W[:, 0], W[:, 1])
    hh = W[:, 2] - D_BS
    L2 = rho0 * rho0 + hh * hh
    c3 = (L2 - D_SE * D_SE - D_EW * D_EW) / (2 * D_SE * D_EW)
    reach = (np.abs(c3) <= 1.0 + C3_TOL) & (rho0 * rho0 >= RHO2_MIN)
    a3 = np.arccos(np.clip(c3, -1.0, 1.0))
    sols = np.zeros((E, 8, 7))
    valid = np.zeros((E, 8), dtype=bool)
    k = 0
    for sb in (1.0, -1.0):                                   # base branch
        q0 = np.arctan2(sb * W[:, 1], sb * W[:, 0])
        rho = sb * rho0
        for se in (1.0, -1.0):                               # elbow branch
            q3 = se * a3
            beta = np.arctan2(D_EW * np.sin(q3), D_SE + D_EW * np.cos(q3))
            q1 = np.arctan2(rho, hh) + beta
            q1 = np.arctan2(np.sin(q1), np.cos(q1))           # wrap to (-pi, pi]
            M = np.swapaxes(_r04(q0, q1, q3), 1, 2) @ R
            c5 = np.clip(M[:, 2, 2], -1.0, 1.0)
            s5a = np.sqrt(np.maximum(0.0, 1.0 - c5 * c5))
            for sw in (1.0, -1.0):                           # wrist branch
                q5 = np.arctan2(sw * s5a, c5)
                q4 = np.arctan2(sw * M[:, 1, 2], sw * M[:, 0, 2])
                q6 = np.arctan2(sw * M[:, 2, 1], -sw * M[:, 2, 0])
                sols[:, k] = np.stack([q0, q1, np.zeros(E), q3, q4, q5, q6], 1)
                valid[:, k] = reach & (s5a >= SINGULAR_EPS)
                k += 1
    return sols, valid


def ik_within_limits(ee_in_base, upper, lower):
    """(E,4,4) poses, 7 upper / lower joint limits -> (E,) bool: some IK solution lies inside the limits (common.cpp:44-67)."""
    sols, valid = solve(ee_in_base)
    up = np.asarray(upper, dtype=np.float64).reshape(1, 1, 7)
    lo = np.asarray(lower, dtype=np.float64).reshape(1, 1, 7)
    inside = ((sols <= up) & (sols >= lo)).all(-1)
    return (inside & valid).any(-1)
