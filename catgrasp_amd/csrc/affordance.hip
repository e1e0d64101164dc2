// Task-relevant grasp affordance P(T|G) (run_grasp_simulation.py:50-75 compute_grasp_affordance_worker +
// pybullet_env/env_grasp.py:243-283 get_finger_contact_area), one wavefront per (grasp, finger), float64 like the
// reference's numpy/open3d arithmetic:
//   q = cam_in_finger . p for every canonical point; keep points inside the finger's x/z extent; the contact patch
//   is everything within `surface_tol` (in y, the closing direction) of the first point the finger touches
//   (min y for grip_dir +y, max y for -y); reject the finger if that first point's normal faces along grip_dir;
//   P(T|G)_finger = mean affordance over the patch (the reference looks each patch point up in a kd-tree of the
//   full canonical cloud; that nearest-neighbour affordance is grasp-independent and precomputed per point by
//   cg_nearest_neighbor); P(T|G) = mean over the fingers that have a patch, NaN if none (the reference drops it).
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

struct Finger { double xmin, xmax, zmin, zmax; int grip_sign; };   // grip_dir = (0, grip_sign, 0)

struct AffArgs {
  const double* cam_in_finger; long G;      // (G,12) rows [R|t] of inv(finger_in_grasp).inv(grasp_in_cam)
  const double* pts; const double* nrm; const double* aff; int P;
  Finger f[2]; int n_fingers; double tol;
  double* p_t_given_g;                      // (G)
  int* contact_counts;                      // optional (G, n_fingers)
};

__device__ __forceinline__ double wmin(double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ double wmax(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ double wsum(double v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

__global__ __launch_bounds__(256) void grasp_affordance_kernel(AffArgs a) {
  const int lane = threadIdx.x & 63;
  for (long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6); g < a.G; g += (long)gridDim.x * 4) {
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = a.cam_in_finger[g * 12 + k];
    double acc = 0.0; int nvalid = 0;
    for (int fi = 0; fi < a.n_fingers; ++fi) {
      const Finger f = a.f[fi];
      // pass 1: the first-touched y among the points inside the finger's x/z extent
      double yref = f.grip_sign > 0 ? 1e300 : -1e300;
      int within = 0;
      for (int p = lane; p < a.P; p += 64) {
        const double x = a.pts[p * 3], y = a.pts[p * 3 + 1], z = a.pts[p * 3 + 2];
        const double qx = T[0] * x + T[1] * y + T[2] * z + T[3];
        const double qy = T[4] * x + T[5] * y + T[6] * z + T[7];
        const double qz = T[8] * x + T[9] * y + T[10] * z + T[11];
        if (qx >= f.xmin && qx <= f.xmax && qz >= f.zmin && qz <= f.zmax) { ++within; yref = f.grip_sign > 0 ? fmin(yref, qy) : fmax(yref, qy); }
      }
      yref = f.grip_sign > 0 ? wmin(yref) : wmax(yref);
      within = (int)wsum((double)within);
      int cnt = 0;
      if (within > 0) {
        // pass 2: contact patch statistics + the index of the first point realising the minimum distance
        double s = 0.0, dbest = 1e300; int ibest = 0x7fffffff;
        for (int p = lane; p < a.P; p += 64) {
          const double x = a.pts[p * 3], y = a.pts[p * 3 + 1], z = a.pts[p * 3 + 2];
          const double qx = T[0] * x + T[1] * y + T[2] * z + T[3];
          const double qy = T[4] * x + T[5] * y + T[6] * z + T[7];
          const double qz = T[8] * x + T[9] * y + T[10] * z + T[11];
          if (qx >= f.xmin && qx <= f.xmax && qz >= f.zmin && qz <= f.zmax) {
            const double d = fabs(qy - yref);
            if (d <= a.tol) { ++cnt; s += a.aff[p]; if (d < dbest) { dbest = d; ibest = p; } }
          }
        }
        cnt = (int)wsum((double)cnt); s = wsum(s);
        for (int o = 32; o > 0; o >>= 1) {
          const double od = __shfl_xor(dbest, o); const int oi = __shfl_xor(ibest, o);
          if (od < dbest || (od == dbest && oi < ibest)) { dbest = od; ibest = oi; }
        }
        if (cnt > 0) {
          // closest_normal . grip_dir > 0 -> this finger has no valid contact (env_grasp.py:273-277)
          const double nx = a.nrm[ibest * 3], ny = a.nrm[ibest * 3 + 1], nz = a.nrm[ibest * 3 + 2];
          const double fy = T[4] * nx + T[5] * ny + T[6] * nz;
          if (fy * (double)f.grip_sign > 0.0) cnt = 0;
          else { acc += s / (double)cnt; ++nvalid; }
        }
      }
      if (a.contact_counts && lane == 0) a.contact_counts[g * a.n_fingers + fi] = cnt;
    }
    if (lane == 0) a.p_t_given_g[g] = nvalid > 0 ? acc / (double)nvalid : __longlong_as_double(0x7ff8000000000000LL);
  }
}

// brute-force nearest neighbour in float64, first minimum on ties: idx[q] = argmin_r |query_q - ref_r|
__global__ __launch_bounds__(256) void nearest_neighbor_kernel(const double* __restrict__ query, long Q, const double* __restrict__ ref, int R,
                                                               int* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  for (long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6); q < Q; q += (long)gridDim.x * 4) {
    const double x = query[q * 3], y = query[q * 3 + 1], z = query[q * 3 + 2];
    double dbest = 1e300; int ibest = 0x7fffffff;
    for (int r = lane; r < R; r += 64) {
      const double dx = ref[r * 3] - x, dy = ref[r * 3 + 1] - y, dz = ref[r * 3 + 2] - z;
      const double d = dx * dx + dy * dy + dz * dz;
      if (d < dbest) { dbest = d; ibest = r; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double od = __shfl_xor(dbest, o); const int oi = __shfl_xor(ibest, o);
      if (od < dbest || (od == dbest && oi < ibest)) { dbest = od; ibest = oi; }
    }
    if (lane == 0) idx[q] = ibest;
  }
}

}  // namespace

extern "C" int cg_grasp_affordance(const double* cam_in_finger, long G, const double* pts, const double* normals, const double* point_affordance,
                                   int P, int n_fingers, const double* h_finger_extents, const int* h_grip_signs, double surface_tol,
                                   double* p_t_given_g, int* contact_counts, void* stream) {
  if (G < 0 || P < 0 || n_fingers < 1 || n_fingers > 2 || !h_finger_extents || !h_grip_signs) return CG_ERR_ARG;
  if (G == 0) return CG_OK;
  if (!cam_in_finger || !p_t_given_g || (P > 0 && (!pts || !normals || !point_affordance))) return CG_ERR_ARG;
  AffArgs a;
  a.cam_in_finger = cam_in_finger; a.G = G; a.pts = pts; a.nrm = normals; a.aff = point_affordance; a.P = P; a.n_fingers = n_fingers;
  for (int i = 0; i < n_fingers; ++i) {
    a.f[i] = Finger{h_finger_extents[i * 4], h_finger_extents[i * 4 + 1], h_finger_extents[i * 4 + 2], h_finger_extents[i * 4 + 3], h_grip_signs[i]};
    if (h_grip_signs[i] != 1 && h_grip_signs[i] != -1) return CG_ERR_ARG;      // the reference raises for any other grip_dir
  }
  a.tol = surface_tol; a.p_t_given_g = p_t_given_g; a.contact_counts = contact_counts;
  long blocks = (G + 3) / 4; if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(grasp_affordance_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_nearest_neighbor(const double* query, long Q, const double* ref, int R, int* idx, void* stream) {
  if (Q < 0 || R < 0) return CG_ERR_ARG;
  if (Q == 0) return CG_OK;
  if (!query || !ref || !idx || R == 0) return CG_ERR_ARG;
  long blocks = (Q + 3) / 4; if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(nearest_neighbor_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, query, Q, ref, R, idx);
  return cg_hip_status(hipGetLastError());
}
