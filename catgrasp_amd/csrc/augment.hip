// my_cpp.augmentGraspPoses / directionVecToRotation (my_cpp/common.cpp:75-153): fan a surface point's grasp frame
// out over approach directions (sphere points), in-plane rotations and approach depths.  One thread per output pose.
// The reference orthonormalises with Eigen::JacobiSVD (R = U V^T); here the same nearest rotation is obtained
// with the scaled Newton iteration for the polar factor, R <- (R + R^-T)/2, which converges quadratically to U V^T
// for any non-singular R (the inputs are rotations up to float rounding).
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

struct M3 { float m[9]; };

__device__ __forceinline__ M3 mul3(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C.m[r * 3 + c] = (A.m[r * 3] * B.m[c] + A.m[r * 3 + 1] * B.m[3 + c]) + A.m[r * 3 + 2] * B.m[6 + c];
  return C;
}

__device__ __forceinline__ M3 transpose3(const M3& A) {
  return M3{{A.m[0], A.m[3], A.m[6], A.m[1], A.m[4], A.m[7], A.m[2], A.m[5], A.m[8]}};
}

__device__ M3 polar_rotation(M3 R) {
  for (int it = 0; it < 6; ++it) {
    const float* a = R.m;
    const float c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const float det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    if (!(fabsf(det) > 1e-20f)) break;
    const float id = 1.0f / det;
    // inverse transpose = cofactor matrix / det
    M3 T;
    T.m[0] = c00 * id; T.m[1] = c01 * id; T.m[2] = c02 * id;
    T.m[3] = (a[2] * a[7] - a[1] * a[8]) * id; T.m[4] = (a[0] * a[8] - a[2] * a[6]) * id; T.m[5] = (a[1] * a[6] - a[0] * a[7]) * id;
    T.m[6] = (a[1] * a[5] - a[2] * a[4]) * id; T.m[7] = (a[2] * a[3] - a[0] * a[5]) * id; T.m[8] = (a[0] * a[4] - a[1] * a[3]) * id;
#pragma unroll
    for (int k = 0; k < 9; ++k) R.m[k] = 0.5f * (R.m[k] + T.m[k]);
  }
  return R;
}

// directionVecToRotation(direction, ref=(1,0,0))  (common.cpp:75-115)
__device__ M3 direction_to_rotation_x(float dx, float dy, float dz) {
  const M3 I{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  const float n = sqrtf((dx * dx + dy * dy) + dz * dz);
  dx /= n; dy /= n; dz /= n;
  // v = direction x ref, ref = (1,0,0)
  const float vx = 0.f, vy = dz, vz = -dy;
  const float s = sqrtf((vx * vx + vy * vy) + vz * vz);
  if (s < 1e-5f) return I;
  const float c = dx;
  const M3 K{{0, -vz, vy, vz, 0, -vx, -vy, vx, 0}};
  const M3 K2 = mul3(K, K);
  const float f = (1.f - c) / (s * s);
  M3 R;
#pragma unroll
  for (int k = 0; k < 9; ++k) R.m[k] = I.m[k] + K.m[k] + K2.m[k] * f;
  return polar_rotation(transpose3(R));
}

__global__ __launch_bounds__(256) void augment_grasp_poses_kernel(M3 R0, float px, float py, float pz, const float* __restrict__ sphere_pts,
                                                                  int n_sphere, int n_rot, float rot_step, int n_depth, float approach_step,
                                                                  float init_bite, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n_R = 1 + (long)n_sphere * n_rot;
  if (i >= n_R * n_depth) return;
  const long ri = i / n_depth;
  const int di = (int)(i - ri * n_depth);
  M3 R = R0;
  if (ri > 0) {
    const long k = ri - 1;
    const int si = (int)(k / n_rot), qi = (int)(k - (long)si * n_rot);
    const M3 Rs = direction_to_rotation_x(sphere_pts[si * 3], sphere_pts[si * 3 + 1], sphere_pts[si * 3 + 2]);
    float x_rot = 0.f;
    for (int q = 0; q < qi; ++q) x_rot += rot_step;                    // the reference's float accumulation
    const float ang = (float)((double)x_rot / 180.0 * 3.14159265358979323846);
    const float ca = cosf(ang), sa = sinf(ang);
    const M3 Rx{{1, 0, 0, 0, ca, -sa, 0, sa, ca}};
    R = mul3(mul3(R0, Rs), Rx);
  }
  R = polar_rotation(R);
  float d = 0.f;
  for (int q = 0; q < di; ++q) d += approach_step;
  const float ax = R.m[0], ay = R.m[3], az = R.m[6];
  float* o = out + i * 16;
  o[0] = R.m[0]; o[1] = R.m[1]; o[2] = R.m[2]; o[3] = px + init_bite * ax + ax * d;
  o[4] = R.m[3]; o[5] = R.m[4]; o[6] = R.m[5]; o[7] = py + init_bite * ay + ay * d;
  o[8] = R.m[6]; o[9] = R.m[7]; o[10] = R.m[8]; o[11] = pz + init_bite * az + az * d;
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
}

}  // namespace

extern "C" int cg_augment_grasp_poses(const float* h_R0, const float* h_selected_point, const float* sphere_pts, int n_sphere,
                                      int n_rot, float inplane_rot_step, int n_depth, float approach_step, float init_bite,
                                      float* out, void* stream) {
  if (!h_R0 || !h_selected_point || n_sphere < 0 || n_rot < 0 || n_depth < 0) return CG_ERR_ARG;
  const long total = (1 + (long)n_sphere * n_rot) * n_depth;
  if (total == 0) return CG_OK;
  if (!out || (n_sphere > 0 && !sphere_pts)) return CG_ERR_ARG;
  M3 R0; for (int k = 0; k < 9; ++k) R0.m[k] = h_R0[k];
  hipLaunchKernelGGL(augment_grasp_poses_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, R0,
                     h_selected_point[0], h_selected_point[1], h_selected_point[2], sphere_pts, n_sphere, n_rot, inplane_rot_step,
                     n_depth, approach_step, init_bite, out);
  return cg_hip_status(hipGetLastError());
}
