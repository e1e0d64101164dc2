// Small HBM-bound kernels around the two networks:
//  * cg_build_grasp_input  : per-candidate GraspDataset.transform on the device (dataset_grasp.py:63-91):
//                            gather resampled points, cloud -> grasp frame, optional (x-mean)/std.
//  * cg_build_nunocs_input : NunocsIsolatedDataset.transform + NormalizeCloud
//                            (dataset_nunocs.py:38-65, augmentations.py:66-75).
//  * cg_softmax_pg         : softmax / argmax / confidence of predicter.py:86-91 and the
//                            p_G reduction of run_grasp_simulation.py:313.
//  * cg_nunocs_decode      : bin argmax decode + z-confidence of predicter.py:144-150.
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

// one thread per output point; consecutive threads write consecutive 24-byte rows (fully coalesced
// stores), gathers hit the L2-resident object cloud.
__global__ __launch_bounds__(256) void build_grasp_input_kernel(
    const float* __restrict__ xyz, const float* __restrict__ nrm, const int* __restrict__ ids,
    const float* __restrict__ pose_inv, const float* __restrict__ mean, const float* __restrict__ inv_std,
    int G, int n_pts, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)G * n_pts) return;
  const int g = (int)(i / n_pts);
  const float* T = pose_inv + (size_t)g * 12;   // rows of [R | t]: x_g = R x + t, n_g = R n
  const int id = ids[i];
  const float px = xyz[(size_t)id * 3 + 0], py = xyz[(size_t)id * 3 + 1], pz = xyz[(size_t)id * 3 + 2];
  const float nx = nrm[(size_t)id * 3 + 0], ny = nrm[(size_t)id * 3 + 1], nz = nrm[(size_t)id * 3 + 2];
  float v[6];
  v[0] = fmaf(T[0], px, fmaf(T[1], py, fmaf(T[2], pz, T[3])));
  v[1] = fmaf(T[4], px, fmaf(T[5], py, fmaf(T[6], pz, T[7])));
  v[2] = fmaf(T[8], px, fmaf(T[9], py, fmaf(T[10], pz, T[11])));
  v[3] = fmaf(T[0], nx, fmaf(T[1], ny, T[2] * nz));
  v[4] = fmaf(T[4], nx, fmaf(T[5], ny, T[6] * nz));
  v[5] = fmaf(T[8], nx, fmaf(T[9], ny, T[10] * nz));
  if (mean) {
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = (v[j] - mean[j]) * inv_std[j];
  }
  f32x2* o = (f32x2*)(out + i * 6);
  o[0] = f32x2{v[0], v[1]}; o[1] = f32x2{v[2], v[3]}; o[2] = f32x2{v[4], v[5]};
}

// one workgroup per object cloud: gather -> min/max reduce -> normalise.
__global__ __launch_bounds__(1024) void build_nunocs_input_kernel(
    const float* __restrict__ xyz, const float* __restrict__ nrm, const int* __restrict__ ids,
    const float* __restrict__ mean, const float* __restrict__ inv_std, int n_pts, float* __restrict__ out) {
  __shared__ float red[6][16];
  __shared__ float mn[3], inv_scale;
  const int b = blockIdx.x;
  const int* idb = ids + (size_t)b * n_pts;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n_pts; i += blockDim.x) {
    const int id = idb[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) { float v = xyz[(size_t)id * 3 + j]; lo[j] = fminf(lo[j], v); hi[j] = fmaxf(hi[j], v); }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    for (int o = 32; o > 0; o >>= 1) { lo[j] = fminf(lo[j], __shfl_xor(lo[j], o)); hi[j] = fmaxf(hi[j], __shfl_xor(hi[j], o)); }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { for (int j = 0; j < 3; ++j) { red[j][w] = lo[j]; red[3 + j][w] = hi[j]; } }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    float l[3], h[3];
    for (int j = 0; j < 3; ++j) { l[j] = red[j][0]; h[j] = red[3 + j][0]; for (int k = 1; k < nw; ++k) { l[j] = fminf(l[j], red[j][k]); h[j] = fmaxf(h[j], red[3 + j][k]); } }
    const float scale = fmaxf(fmaxf(h[0] - l[0], h[1] - l[1]), h[2] - l[2]);
    mn[0] = l[0]; mn[1] = l[1]; mn[2] = l[2];
    inv_scale = scale + 1e-15f;   // divisor, as the reference: (xyz - min) / (scale + 1e-15)
  }
  __syncthreads();
  const float div = inv_scale;
  for (int i = threadIdx.x; i < n_pts; i += blockDim.x) {
    const int id = idb[i];
    float v[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) { v[j] = (xyz[(size_t)id * 3 + j] - mn[j]) / div; v[3 + j] = nrm[(size_t)id * 3 + j]; }
    if (mean) {
#pragma unroll
      for (int j = 0; j < 6; ++j) v[j] = (v[j] - mean[j]) * inv_std[j];
    }
    float* o = out + ((size_t)b * n_pts + i) * 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) o[j] = v[j];
  }
}

__global__ __launch_bounds__(256) void softmax_pg_kernel(const float* __restrict__ logits, int B, int C,
                                                         float* __restrict__ probs, int* __restrict__ label,
                                                         float* __restrict__ conf, float* __restrict__ p_g) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* l = logits + (size_t)b * C;
  float m = l[0]; int am = 0;
  for (int k = 1; k < C; ++k) { if (l[k] > m) { m = l[k]; am = k; } }
  float s = 0.f;
  for (int k = 0; k < C; ++k) s += expf(l[k] - m);
  const float inv = 1.f / s;
  float pg = 0.f, best = -1.f; int bl = 0;
  for (int k = 0; k < C; ++k) {
    const float p = expf(l[k] - m) * inv;
    probs[(size_t)b * C + k] = p;
    pg = fmaf(p, (float)k, pg);
    if (p > best) { best = p; bl = k; }       // argmax of the probabilities, first maximum (predicter.py:89)
  }
  (void)am;
  label[b] = bl; conf[b] = best; p_g[b] = pg / (float)C;
}

__global__ __launch_bounds__(256) void nunocs_decode_kernel(const float* __restrict__ logits, long P, int nbins,
                                                            float* __restrict__ coords, float* __restrict__ conf_z) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (point, axis)
  if (i >= P * 3) return;
  const float* l = logits + i * nbins;
  float m = l[0]; int am = 0;
  for (int k = 1; k < nbins; ++k) { if (l[k] > m) { m = l[k]; am = k; } }
  coords[i] = (float)am * (1.0f / (float)nbins) - 0.5f;
  if ((i % 3) == 2) {
    float s = 0.f;
    for (int k = 0; k < nbins; ++k) s += expf(l[k] - m);
    conf_z[i / 3] = 1.f / s;
  }
}

}  // namespace

extern "C" int cg_build_grasp_input(const float* cloud_xyz, const float* cloud_normal, int n_cloud, const int* ids,
                                    const float* pose_inv, const float* mean, const float* inv_std, int G, int n_pts,
                                    float* out, void* stream) {
  if (!cloud_xyz || !cloud_normal || !ids || !pose_inv || !out || n_cloud <= 0 || G < 0 || n_pts <= 0) return CG_ERR_ARG;
  if ((mean == nullptr) != (inv_std == nullptr)) return CG_ERR_ARG;
  if (G == 0) return CG_OK;
  const long total = (long)G * n_pts;
  hipLaunchKernelGGL(build_grasp_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     cloud_xyz, cloud_normal, ids, pose_inv, mean, inv_std, G, n_pts, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_build_nunocs_input(const float* cloud_xyz, const float* cloud_normal, int n_cloud, const int* ids,
                                     const float* mean, const float* inv_std, int B, int n_pts, float* out, void* stream) {
  if (!cloud_xyz || !cloud_normal || !ids || !out || n_cloud <= 0 || B < 0 || n_pts <= 0) return CG_ERR_ARG;
  if ((mean == nullptr) != (inv_std == nullptr)) return CG_ERR_ARG;
  if (B == 0) return CG_OK;
  hipLaunchKernelGGL(build_nunocs_input_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream,
                     cloud_xyz, cloud_normal, ids, mean, inv_std, n_pts, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_softmax_pg(const float* logits, int B, int C, float* probs, int* label, float* conf, float* p_g,
                             void* stream) {
  if (!logits || !probs || !label || !conf || !p_g || B < 0 || C <= 0) return CG_ERR_ARG;
  if (B == 0) return CG_OK;
  hipLaunchKernelGGL(softmax_pg_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     logits, B, C, probs, label, conf, p_g);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_nunocs_decode(const float* logits, long P, int nbins, float* coords, float* conf_z, void* stream) {
  if (!logits || !coords || !conf_z || P < 0 || nbins <= 0) return CG_ERR_ARG;
  if (P == 0) return CG_OK;
  hipLaunchKernelGGL(nunocs_decode_kernel, dim3((unsigned)((P * 3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     logits, P, nbins, coords, conf_z);
  return cg_hip_status(hipGetLastError());
}
