// get_ik_within_limits (my_cpp/common.cpp:9-72) on the device: closed-form IK of the KUKA LBR iiwa14 with the redundancy
// joint (index 2) fixed at 0, one thread per end-effector pose, float64.  The algorithm, its derivation from the arm's DH
// table and the degeneracy windows of the reference's generated solver that it reproduces are stated in
// oracle/iiwa_ik_ref.py (the host restatement of the same arithmetic used by the tests, itself pinned to the real solver's
// answers in tests/golden/iiwa_ik_golden.npz).
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

struct Limits { double up[7]; double lo[7]; };

constexpr double D_BS = 0.36, D_SE = 0.42, D_EW = 0.4, D_WF = 0.081;
constexpr double RHO2_MIN = 1e-6, C3_TOL = 1e-7, SINGULAR_EPS = 2e-3;

// R <- R . (Rz(q) Rx(alpha)),  alpha = sgn * pi/2
__device__ __forceinline__ void mul_link(double* R, double q, double sgn) {
  const double c = cos(q), s = sin(q);
  // Rz(q) Rx(alpha) = [[c, 0, s*sa], [s, 0, -c*sa], [0, sa, 0]] for cos(alpha) = 0, sa = sgn
  double out[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double a = R[r * 3 + 0], b = R[r * 3 + 1], d = R[r * 3 + 2];
    out[r * 3 + 0] = a * c + b * s;
    out[r * 3 + 1] = d * sgn;
    out[r * 3 + 2] = (a * s - b * c) * sgn;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) R[k] = out[k];
}

__device__ __forceinline__ bool inside(double q, int j, const Limits& lim) { return q <= lim.up[j] && q >= lim.lo[j]; }

__global__ __launch_bounds__(256) void iiwa_ik_kernel(const float* __restrict__ ee, long E, Limits lim, unsigned char* __restrict__ ok) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float* T = ee + e * 16;
  double R[9], p[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) R[r * 3 + c] = (double)T[r * 4 + c];
    p[r] = (double)T[r * 4 + 3];
  }
  const double wx = p[0] - D_WF * R[2], wy = p[1] - D_WF * R[5], wz = p[2] - D_WF * R[8];
  const double rho0 = hypot(wx, wy), hh = wz - D_BS;
  const double c3 = (rho0 * rho0 + hh * hh - D_SE * D_SE - D_EW * D_EW) / (2 * D_SE * D_EW);
  bool found = false;
  if (fabs(c3) <= 1.0 + C3_TOL && rho0 * rho0 >= RHO2_MIN && inside(0.0, 2, lim)) {
    const double a3 = acos(fmin(1.0, fmax(-1.0, c3)));
    for (int ib = 0; ib < 2 && !found; ++ib) {
      const double sb = ib ? -1.0 : 1.0;
      const double q0 = atan2(sb * wy, sb * wx);
      if (!inside(q0, 0, lim)) continue;
      for (int ie = 0; ie < 2 && !found; ++ie) {
        const double q3 = (ie ? -1.0 : 1.0) * a3;
        if (!inside(q3, 3, lim)) continue;
        double q1 = atan2(sb * rho0, hh) + atan2(D_EW * sin(q3), D_SE + D_EW * cos(q3));
        q1 = atan2(sin(q1), cos(q1));
        if (!inside(q1, 1, lim)) continue;
        double A[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        mul_link(A, q0, -1.0); mul_link(A, q1, 1.0); mul_link(A, 0.0, 1.0); mul_link(A, q3, -1.0);
        // M = A^T R (only the entries the wrist needs)
        const double m02 = A[0] * R[2] + A[3] * R[5] + A[6] * R[8];
        const double m12 = A[1] * R[2] + A[4] * R[5] + A[7] * R[8];
        const double m22 = A[2] * R[2] + A[5] * R[5] + A[8] * R[8];
        const double m20 = A[2] * R[0] + A[5] * R[3] + A[8] * R[6];
        const double m21 = A[2] * R[1] + A[5] * R[4] + A[8] * R[7];
        const double c5 = fmin(1.0, fmax(-1.0, m22));
        const double s5a = sqrt(fmax(0.0, 1.0 - c5 * c5));
        if (s5a < SINGULAR_EPS) continue;
        for (int iw = 0; iw < 2; ++iw) {
          const double sw = iw ? -1.0 : 1.0;
          const double q5 = atan2(sw * s5a, c5);
          const double q4 = atan2(sw * m12, sw * m02);
          const double q6 = atan2(sw * m21, -sw * m20);
          if (inside(q4, 4, lim) && inside(q5, 5, lim) && inside(q6, 6, lim)) { found = true; break; }
        }
      }
    }
  }
  ok[e] = found ? 1 : 0;
}

}  // namespace

extern "C" int cg_iiwa_ik_within_limits(const float* ee_in_base, long E, const double* h_upper7, const double* h_lower7,
                                        unsigned char* ok, void* stream) {
  if (E < 0 || !h_upper7 || !h_lower7) return CG_ERR_ARG;
  if (E == 0) return CG_OK;
  if (!ee_in_base || !ok) return CG_ERR_ARG;
  Limits lim;
  for (int j = 0; j < 7; ++j) { lim.up[j] = h_upper7[j]; lim.lo[j] = h_lower7[j]; }
  hipLaunchKernelGGL(iiwa_ik_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ee_in_base, E, lim, ok);
  return cg_hip_status(hipGetLastError());
}
