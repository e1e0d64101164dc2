// Cone grasp-candidate generation (row N3b): the device form of PointConeGraspSampler.sample_one_surface_point
// (dexnet/grasping/grasp_sampler.py:225-298) and the centring step of sample_grasps (:189-198), float64 like numpy.
//   cone_frames_kernel  one wave per sampled surface point: accumulate M = sum n n^T over the neighbours inside r_ball,
//                       smallest-eigenvalue direction (Jacobi), Gram-Schmidt against the approach axis -> R0
//   cone_poses_kernel   one thread per pose: R0 . R_sphere(dir) . Rx(angle), column-normalised, stepped along the approach
//   center_grasps_kernel one wave per pose: shift along the closing axis to the middle of the object's extent
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

__device__ void jacobi3d(double S[3][3], double V[3][3], double e[3]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = (i == j);
  for (int sweep = 0; sweep < 16; ++sweep) {
    const double off = fabs(S[0][1]) + fabs(S[0][2]) + fabs(S[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(S[p][q]) < 1e-300) continue;
        const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { const double a = S[k][p], b = S[k][q]; S[k][p] = c * a - s * b; S[k][q] = s * a + c * b; }
        for (int k = 0; k < 3; ++k) { const double a = S[p][k], b = S[q][k]; S[p][k] = c * a - s * b; S[q][k] = s * a + c * b; }
        for (int k = 0; k < 3; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
      }
  }
  for (int i = 0; i < 3; ++i) e[i] = S[i][i];
}

__device__ __forceinline__ double wsum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// mode 0: out_doublings[k] = number of r_ball doublings sample k needs before sum(M) != 0 (grasp_sampler.py:243-247)
// mode 1: frames[k] = R0 (9 doubles, row-major) using radius r_ball[k]
__global__ __launch_bounds__(256) void cone_frames_kernel(const double* __restrict__ pts, const double* __restrict__ nrm, int P,
                                                          const int* __restrict__ sample_ids, int K, const double* __restrict__ r_ball, double r0,
                                                          int mode, int* __restrict__ out_doublings, double* __restrict__ frames) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  const int sid = sample_ids[k];
  const double sx = pts[sid * 3], sy = pts[sid * 3 + 1], sz = pts[sid * 3 + 2];
  double r = mode == 0 ? r0 : r_ball[k];
  double M[3][3];
  int doublings = 0;
  for (;;) {
    double m[6] = {0, 0, 0, 0, 0, 0};
    for (int p = lane; p < P; p += 64) {
      const double dx = sx - pts[p * 3], dy = sy - pts[p * 3 + 1], dz = sz - pts[p * 3 + 2];
      const double d2 = dx * dx + dy * dy + dz * dz;
      if (sqrt(d2) <= r && d2 != 0.0) {                    // query_ball_point(r) and `sqr_distances != 0`
        double nx = nrm[p * 3], ny = nrm[p * 3 + 1], nz = nrm[p * 3 + 2];
        const double nn = sqrt(nx * nx + ny * ny + nz * nz);
        if (nn != 0.0) {
          nx /= nn; ny /= nn; nz /= nn;
          m[0] += nx * nx; m[1] += nx * ny; m[2] += nx * nz; m[3] += ny * ny; m[4] += ny * nz; m[5] += nz * nz;
        }
      }
    }
    for (int i = 0; i < 6; ++i) m[i] = wsum(m[i]);
    M[0][0] = m[0]; M[0][1] = M[1][0] = m[1]; M[0][2] = M[2][0] = m[2]; M[1][1] = m[3]; M[1][2] = M[2][1] = m[4]; M[2][2] = m[5];
    const double total = m[0] + 2 * m[1] + 2 * m[2] + m[3] + 2 * m[4] + m[5];      // sum(sum(M))
    if (mode == 1 || total != 0.0 || doublings >= 60) break;
    r *= 2.0; ++doublings;
  }
  if (mode == 0) { if (lane == 0) out_doublings[k] = doublings; return; }
  if (lane != 0) return;
  double a[3] = {-nrm[sid * 3], -nrm[sid * 3 + 1], -nrm[sid * 3 + 2]};
  const double an = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  for (int i = 0; i < 3; ++i) a[i] /= an;
  double V[3][3], e[3];
  jacobi3d(M, V, e);
  int im = 0; if (e[1] < e[im]) im = 1; if (e[2] < e[im]) im = 2;
  double mn[3] = {V[0][im], V[1][im], V[2][im]};
  const double pr = (a[0] * mn[0] + a[1] * mn[1] + a[2] * mn[2]) / (a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  for (int i = 0; i < 3; ++i) mn[i] -= pr * a[i];
  const double mnn = sqrt(mn[0] * mn[0] + mn[1] * mn[1] + mn[2] * mn[2]);
  for (int i = 0; i < 3; ++i) mn[i] /= mnn;
  double mj[3] = {mn[1] * a[2] - mn[2] * a[1], mn[2] * a[0] - mn[0] * a[2], mn[0] * a[1] - mn[1] * a[0]};
  const double mjn = sqrt(mj[0] * mj[0] + mj[1] * mj[1] + mj[2] * mj[2]);
  for (int i = 0; i < 3; ++i) mj[i] /= mjn;
  double* o = frames + (size_t)k * 9;
  for (int i = 0; i < 3; ++i) { o[i * 3] = a[i]; o[i * 3 + 1] = mj[i]; o[i * 3 + 2] = mn[i]; }
}

__device__ void mul3d(const double* A, const double* B, double* C) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
__device__ void colnorm3d(double* R) {                     // Utils.normalizeRotation (Utils.py:172-178)
  for (int c = 0; c < 3; ++c) { const double n = sqrt(R[c] * R[c] + R[3 + c] * R[3 + c] + R[6 + c] * R[6 + c]); R[c] /= n; R[3 + c] /= n; R[6 + c] /= n; }
}

// python directionVecToRotation(direction, ref=(1,0,0)) (Utils.py:262-290)
__device__ void dir_to_rot_x(const double* d_in, double* R) {
  double d[3]; const double n = sqrt(d_in[0] * d_in[0] + d_in[1] * d_in[1] + d_in[2] * d_in[2]);
  for (int i = 0; i < 3; ++i) d[i] = d_in[i] / n;
  const double v[3] = {0.0, d[2], -d[1]};                   // cross(direction, (1,0,0))
  if (v[1] == 0.0 && v[2] == 0.0) { for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0); return; }
  const double s = sqrt(v[1] * v[1] + v[2] * v[2]), c = d[0];
  const double K[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
  double K2[9]; mul3d(K, K, K2);
  double Rt[9];
  for (int i = 0; i < 9; ++i) Rt[i] = (i % 4 == 0) + K[i] + K2[i] * (1 - c) / (s * s);
  for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) R[r * 3 + cc] = Rt[cc * 3 + r];      // R = R.T
  colnorm3d(R);
}

__global__ __launch_bounds__(256) void cone_poses_kernel(const double* __restrict__ pts, const int* __restrict__ sample_ids, const double* __restrict__ frames,
                                                         int K, const double* __restrict__ sphere_pts, int S, int n_rot, double rot_step_deg,
                                                         int n_depth, double approach_step, double init_bite, double* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per = (1 + (long)S * n_rot) * n_depth;
  if (i >= (long)K * per) return;
  const int k = (int)(i / per);
  const long j = i - (long)k * per;
  const long ri = j / n_depth;
  const int di = (int)(j - ri * n_depth);
  double R[9];
  const double* R0 = frames + (size_t)k * 9;
  if (ri == 0) { for (int q = 0; q < 9; ++q) R[q] = R0[q]; }
  else {
    const long m = ri - 1;
    const int si = (int)(m / n_rot), qi = (int)(m - (long)si * n_rot);
    double Rs[9], T[9];
    dir_to_rot_x(sphere_pts + si * 3, Rs);
    const double ang = ((double)qi * rot_step_deg) * 3.14159265358979323846 / 180.0;      // np.arange(0,180,step)[qi] * pi/180
    const double ca = cos(ang), sa = sin(ang);
    const double Rx[9] = {1, 0, 0, 0, ca, -sa, 0, sa, ca};
    mul3d(R0, Rs, T); mul3d(T, Rx, R);
  }
  colnorm3d(R);
  const double d = (double)di * approach_step;                                            // np.arange(0, hand_depth, step)[di]
  const int sid = sample_ids[k];
  double* o = out + i * 16;
  for (int r = 0; r < 3; ++r) {
    o[r * 4] = R[r * 3]; o[r * 4 + 1] = R[r * 3 + 1]; o[r * 4 + 2] = R[r * 3 + 2];
    o[r * 4 + 3] = pts[sid * 3 + r] + init_bite * R[r * 3] + R[r * 3] * d;
  }
  o[12] = 0; o[13] = 0; o[14] = 0; o[15] = 1;
}

// grasp_pose <- grasp_pose @ translate(0, cy, 0), cy = mid y-extent of inv(grasp_pose).points (grasp_sampler.py:191-197)
__global__ __launch_bounds__(256) void center_grasps_kernel(double* __restrict__ poses, long G, const double* __restrict__ pts, int P) {
  const int lane = threadIdx.x & 63;
  for (long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6); g < G; g += (long)gridDim.x * 4) {
    double* T = poses + g * 16;
    // y row of the inverse of a rigid pose: R^T row 1 = column 1 of R;  ty = -col1 . t
    const double r0 = T[1], r1 = T[5], r2 = T[9];
    const double ty = -(r0 * T[3] + r1 * T[7] + r2 * T[11]);
    double lo = 1e300, hi = -1e300;
    for (int p = lane; p < P; p += 64) { const double y = r0 * pts[p * 3] + r1 * pts[p * 3 + 1] + r2 * pts[p * 3 + 2] + ty; lo = fmin(lo, y); hi = fmax(hi, y); }
    for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
    const double cy = (hi + lo) / 2;
    __builtin_amdgcn_wave_barrier();
    if (lane < 3) T[lane * 4 + 3] += T[lane * 4 + 1] * cy;
  }
}

}  // namespace

extern "C" int cg_cone_frames(const double* pts, const double* normals, int P, const int* sample_ids, int K, const double* r_ball, double r0,
                              int mode, int* out_doublings, double* frames, void* stream) {
  if (P < 0 || K < 0 || (mode != 0 && mode != 1)) return CG_ERR_ARG;
  if (K == 0) return CG_OK;
  if (!pts || !normals || !sample_ids || (mode == 0 && !out_doublings) || (mode == 1 && (!frames || !r_ball))) return CG_ERR_ARG;
  hipLaunchKernelGGL(cone_frames_kernel, dim3((unsigned)((K + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pts, normals, P, sample_ids, K, r_ball,
                     r0, mode, out_doublings, frames);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_cone_poses(const double* pts, const int* sample_ids, const double* frames, int K, const double* sphere_pts, int S, int n_rot,
                             double rot_step_deg, int n_depth, double approach_step, double init_bite, double* out, void* stream) {
  if (K < 0 || S < 0 || n_rot < 0 || n_depth < 0) return CG_ERR_ARG;
  const long total = (long)K * (1 + (long)S * n_rot) * n_depth;
  if (total == 0) return CG_OK;
  if (!pts || !sample_ids || !frames || !out || (S > 0 && !sphere_pts)) return CG_ERR_ARG;
  hipLaunchKernelGGL(cone_poses_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts, sample_ids, frames, K,
                     sphere_pts, S, n_rot, rot_step_deg, n_depth, approach_step, init_bite, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_center_grasps(double* poses, long G, const double* pts, int P, void* stream) {
  if (G < 0 || P < 0) return CG_ERR_ARG;
  if (G == 0 || P == 0) return CG_OK;
  if (!poses || !pts) return CG_ERR_ARG;
  long blocks = (G + 3) / 4; if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(center_grasps_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, poses, G, pts, P);
  return cg_hip_status(hipGetLastError());
}
