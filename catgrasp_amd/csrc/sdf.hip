// Signed-distance-field lookups of meshpy's Sdf3D (meshpy/meshpy/sdf.py:216-389) on the device.
//   mode 0  trilinear  `_signed_distance(coords, fast=False)`  sdf.py:312-343: clip to [0,dim-1], 8 corners in the
//           reference's corner order, weight = prod(1-|corner-x|), corners outside the grid contribute 0
//   mode 1  nearest    `_signed_distance(coords, fast=True)` / `_signed_distance_batch` sdf.py:318-321,351-357:
//           round-half-even, clamp, gather
//   any-inside         `is_any_points_inside` sdf.py:377-389: round, drop out-of-grid points, any(sd < 0)
// and the batched per-candidate form the reference prepares with transform_pt_obj_to_grid_batch (sdf.py:362-373):
// one wavefront per candidate transforms the scene points into the gripper's grid frame and tests any(sd<0).
// Grid layout: data[i][j][k] row-major (x slowest, z fastest), exactly `self.data_` (sdf_file.py:59-87).
// Pure gather kernels: 12 B of coordinates in + 4 B out per point, the grid (<= ~20 MB) is L2/MALL resident.
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

struct Grid { const float* data; int nx, ny, nz; };

__device__ __forceinline__ float grid_at(const Grid& g, int i, int j, int k) {
  return g.data[((size_t)i * g.ny + j) * g.nz + k];
}

__device__ __forceinline__ float clipf(float v, float hi) { return fminf(fmaxf(v, 0.f), hi); }

__device__ float sdf_trilinear(const Grid& g, float x, float y, float z) {
  x = clipf(x, (float)(g.nx - 1)); y = clipf(y, (float)(g.ny - 1)); z = clipf(z, (float)(g.nz - 1));
  const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
  // corner order of Sdf3D.min_coords_* / max_coords_* (sdf.py:219-224): bit patterns (x,y,z) with 1 = max
  const int cx[8] = {0, 1, 0, 0, 1, 0, 1, 1};
  const int cy[8] = {0, 0, 1, 0, 1, 1, 0, 1};
  const int cz[8] = {0, 0, 0, 1, 0, 1, 1, 1};
  float sd = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float px = fx + (float)cx[c], py = fy + (float)cy[c], pz = fz + (float)cz[c];
    const int i = (int)px, j = (int)py, k = (int)pz;
    float v = 0.f;
    if (i >= 0 && j >= 0 && k >= 0 && i < g.nx && j < g.ny && k < g.nz) v = grid_at(g, i, j, k);
    const float w = ((1.f - fabsf(px - x)) * (1.f - fabsf(py - y))) * (1.f - fabsf(pz - z));
    sd = sd + w * v;
  }
  return sd;
}

__device__ __forceinline__ float sdf_nearest(const Grid& g, float x, float y, float z) {
  int i = (int)rintf(x), j = (int)rintf(y), k = (int)rintf(z);      // round-half-even like np.round / torch.round
  i = min(max(i, 0), g.nx - 1); j = min(max(j, 0), g.ny - 1); k = min(max(k, 0), g.nz - 1);
  return grid_at(g, i, j, k);
}

// coords: (B,3,N) component-major like the reference's (3,N)/(B,3,N) arrays
__global__ __launch_bounds__(256) void sdf_lookup_kernel(Grid g, const float* __restrict__ coords, long B, long N, int mode,
                                                         float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  const long b = i / N, p = i - b * N;
  const float* c = coords + b * 3 * N;
  const float x = c[p], y = c[N + p], z = c[2 * N + p];
  out[i] = mode == 0 ? sdf_trilinear(g, x, y, z) : sdf_nearest(g, x, y, z);
}

__device__ __forceinline__ bool inside_neg(const Grid& g, float x, float y, float z) {
  const float rx = rintf(x), ry = rintf(y), rz = rintf(z);
  if (!(rx >= 0.f && ry >= 0.f && rz >= 0.f && rx < (float)g.nx && ry < (float)g.ny && rz < (float)g.nz)) return false;
  return grid_at(g, (int)rx, (int)ry, (int)rz) < 0.f;
}

__global__ __launch_bounds__(256) void sdf_any_inside_kernel(Grid g, const float* __restrict__ coords, long N, int* __restrict__ flag) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  bool in = false;
  if (p < N) in = inside_neg(g, coords[p], coords[N + p], coords[2 * N + p]);
  if (__ballot(in) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// One wave per candidate: x_grid = A_e x + t_e for every scene point (A|t rows (E,12), the composition
// T_world_grid . inv(gripper_in_cam_e)), any(sd[round(x_grid)] < 0) over in-grid points.
__global__ __launch_bounds__(256) void sdf_points_inside_batch_kernel(Grid g, const float* __restrict__ xf, long E,
                                                                      const float* __restrict__ pts, int P, unsigned char* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (long e = (long)blockIdx.x * 4 + (threadIdx.x >> 6); e < E; e += (long)gridDim.x * 4) {
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = xf[e * 12 + k];
    bool hit = false;
    for (int p0 = 0; p0 < P; p0 += 64) {
      const int p = p0 + lane;
      bool in = false;
      if (p < P) {
        const float x = pts[p * 3 + 0], y = pts[p * 3 + 1], z = pts[p * 3 + 2];
        const float gx = fmaf(T[0], x, fmaf(T[1], y, fmaf(T[2], z, T[3])));
        const float gy = fmaf(T[4], x, fmaf(T[5], y, fmaf(T[6], z, T[7])));
        const float gz = fmaf(T[8], x, fmaf(T[9], y, fmaf(T[10], z, T[11])));
        in = inside_neg(g, gx, gy, gz);
      }
      if (__ballot(in) != 0ull) { hit = true; break; }
    }
    if (lane == 0) out[e] = hit ? 1 : 0;
  }
}

}  // namespace

extern "C" int cg_sdf_lookup(const float* grid, int nx, int ny, int nz, const float* coords, long B, long N, int mode, float* out,
                             void* stream) {
  if (nx <= 0 || ny <= 0 || nz <= 0 || B < 0 || N < 0 || (mode != 0 && mode != 1)) return CG_ERR_ARG;
  if (B * N == 0) return CG_OK;
  if (!grid || !coords || !out) return CG_ERR_ARG;
  hipLaunchKernelGGL(sdf_lookup_kernel, dim3((unsigned)((B * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     Grid{grid, nx, ny, nz}, coords, B, N, mode, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_sdf_any_inside(const float* grid, int nx, int ny, int nz, const float* coords, long N, int* flag, void* stream) {
  if (nx <= 0 || ny <= 0 || nz <= 0 || N < 0) return CG_ERR_ARG;
  if (N == 0) return CG_OK;
  if (!grid || !coords || !flag) return CG_ERR_ARG;
  hipLaunchKernelGGL(sdf_any_inside_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     Grid{grid, nx, ny, nz}, coords, N, flag);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_sdf_points_inside_batch(const float* grid, int nx, int ny, int nz, const float* cam_to_grid, long E,
                                          const float* pts, int n_pts, unsigned char* out, void* stream) {
  if (nx <= 0 || ny <= 0 || nz <= 0 || E < 0 || n_pts < 0) return CG_ERR_ARG;
  if (E == 0) return CG_OK;
  if (!grid || !cam_to_grid || !out || (n_pts > 0 && !pts)) return CG_ERR_ARG;
  long blocks = (E + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(sdf_points_inside_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     Grid{grid, nx, ny, nz}, cam_to_grid, E, pts, n_pts, out);
  return cg_hip_status(hipGetLastError());
}
