// PointNet++ grouping primitives of the reference's pointnet2.py:14-149 as CDNA4 HIP kernels.
//   square_distance      pointnet2.py:14-33    HBM-write bound (B*N*M*4 bytes out)
//   index_points         pointnet2.py:35-51    gather
//   farthest_point_sample pointnet2.py:54-75   -> fps.hip
//   query_ball_point     pointnet2.py:78-98    first-nsample-by-index selection via ballot + prefix popcount
//                                              (reproduces the reference's full sort of a (B,S,N) int64 tensor
//                                              without materialising or sorting anything)
//   sample_and_group     pointnet2.py:101-129  fused gather + centre subtraction + feature concat
// Float arithmetic mirrors the reference expressions term by term (no FMA contraction: the library is
// built with -ffp-contract=off), so FPS indices are exact and ball-query membership differs from the
// torch CPU evaluation only where |d^2 - r^2| is within float rounding of the -2ab+a^2+b^2 expansion.
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------- square_distance
// dist = -2 * (s . d);  dist += |s|^2;  dist += |d|^2      (pointnet2.py:30-32)
__device__ __forceinline__ float sqdist_expanded(float sx, float sy, float sz, float s2, float dx, float dy, float dz, float d2) {
  const float dot = (sx * dx + sy * dy) + sz * dz;
  float v = -2.0f * dot;
  v = v + s2;
  v = v + d2;
  return v;
}

constexpr int SQ_ROWS = 16;
__global__ __launch_bounds__(256) void square_distance_kernel(const float* __restrict__ src, const float* __restrict__ dst,
                                                              int N, int M, float* __restrict__ out) {
  __shared__ float srow[SQ_ROWS][4];
  const int b = blockIdx.z;
  const int n0 = blockIdx.y * SQ_ROWS;
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < SQ_ROWS) {
    const int n = n0 + threadIdx.x;
    if (n < N) {
      const float* s = src + ((size_t)b * N + n) * 3;
      const float x = s[0], y = s[1], z = s[2];
      srow[threadIdx.x][0] = x; srow[threadIdx.x][1] = y; srow[threadIdx.x][2] = z;
      srow[threadIdx.x][3] = (x * x + y * y) + z * z;
    }
  }
  __syncthreads();
  if (m >= M) return;
  const float* d = dst + ((size_t)b * M + m) * 3;
  const float dx = d[0], dy = d[1], dz = d[2];
  const float d2 = (dx * dx + dy * dy) + dz * dz;
  const int rows = min(SQ_ROWS, N - n0);
  for (int r = 0; r < rows; ++r)
    out[((size_t)b * N + n0 + r) * M + m] = sqdist_expanded(srow[r][0], srow[r][1], srow[r][2], srow[r][3], dx, dy, dz, d2);
}

// The same expression for C-dimensional points (pointnet2.py:14-33 is generic in C): dot and squared norms accumulate over the
// channels in index order, (((x0 y0 + x1 y1) + x2 y2) + ...), which is the C = 3 kernel's order.  Source rows are read through the
// scalar/vector caches (every lane of a wave reads the same address), destination rows once per thread.
__global__ __launch_bounds__(256) void square_distance_nd_kernel(const float* __restrict__ src, const float* __restrict__ dst,
                                                                 int N, int M, int C, float* __restrict__ out) {
  const int b = blockIdx.z;
  const int n0 = blockIdx.y * SQ_ROWS;
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float* d = dst + ((size_t)b * M + m) * C;
  float d2 = d[0] * d[0];
  for (int c = 1; c < C; ++c) d2 = d2 + d[c] * d[c];
  const int rows = min(SQ_ROWS, N - n0);
  for (int r = 0; r < rows; ++r) {
    const float* s = src + ((size_t)b * N + n0 + r) * C;
    float dot = s[0] * d[0], s2 = s[0] * s[0];
    for (int c = 1; c < C; ++c) { dot = dot + s[c] * d[c]; s2 = s2 + s[c] * s[c]; }
    float v = -2.0f * dot;
    v = v + s2;
    v = v + d2;
    out[((size_t)b * N + n0 + r) * M + m] = v;
  }
}

// ---------------------------------------------------------------- index_points
__global__ __launch_bounds__(256) void index_points_kernel(const float* __restrict__ points, const long long* __restrict__ idx,
                                                           int N, int C, long S, long total, float* __restrict__ out, int* __restrict__ err) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long row = i / C;           // b*S + s
  const int c = (int)(i - row * C);
  const long b = row / S;
  const long long id = idx[row];
  if (id < 0 || id >= N) { *err = 1; return; }
  out[i] = points[((size_t)b * N + id) * C + c];
}

// ---------------------------------------------------------------- ball query
// One wavefront per query point.  Points are scanned in index order 64 at a time; the in-radius lanes of a
// chunk get their output slots from a prefix popcount of the ballot, so the result is exactly "the first
// nsample indices with d <= r^2, ascending", padded with the first hit (all N when the ball is empty).
constexpr int BQ_U = 8;
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int S,
                                                         float r2, int nsample, long long* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);           // query index s within cloud b (wave-uniform)
  const int b = blockIdx.y;
  if (q >= S) return;
  const float* qp = new_xyz + ((size_t)b * S + q) * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const float q2 = (qx * qx + qy * qy) + qz * qz;
  const float* xb = xyz + (size_t)b * N * 3;
  long long* o = out + ((size_t)b * S + q) * nsample;
  int count = 0;
  long long first = N;
  // BQ_U chunks of 64 points per trip, their loads all in flight before the first is looked at (a trip is one L2 round trip whatever its
  // size, and a typical ball is full after a few hundred points); the chunks are then consumed in index order exactly as one at a time
  for (int p0 = 0; p0 < N && count < nsample; p0 += 64 * BQ_U) {
    float x[BQ_U], y[BQ_U], z[BQ_U];
#pragma unroll
    for (int u = 0; u < BQ_U; ++u) {
      const int p = p0 + u * 64 + lane, c = p < N ? p : N - 1;
      x[u] = xb[c * 3 + 0]; y[u] = xb[c * 3 + 1]; z[u] = xb[c * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < BQ_U; ++u) {
      if (count >= nsample) break;
      const int p = p0 + u * 64 + lane;
      const float p2 = (x[u] * x[u] + y[u] * y[u]) + z[u] * z[u];
      const float d = sqdist_expanded(qx, qy, qz, q2, x[u], y[u], z[u], p2);
      const bool in = p < N && !(d > r2);                      // group_idx[sqrdists > radius**2] = N
      const unsigned long long mask = __ballot(in);
      if (mask == 0ull) continue;
      if (first == N) first = p0 + u * 64 + __builtin_ctzll(mask);
      const int pos = count + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
      if (in && pos < nsample) o[pos] = p;
      count += __builtin_popcountll(mask);
    }
  }
  if (count > nsample) count = nsample;
  for (int k = count + lane; k < nsample; k += 64) o[k] = first;
}

// ---------------------------------------------------------------- fused grouping (sample_and_group tail)
// new_points[b,s,k,:] = cat(xyz[idx] - new_xyz[s], points[idx])      (pointnet2.py:118-123)
__global__ __launch_bounds__(256) void group_points_kernel(const float* __restrict__ xyz, const float* __restrict__ points,
                                                           const float* __restrict__ new_xyz, const long long* __restrict__ idx,
                                                           int N, int S, int K, int D, long total, float* __restrict__ new_points,
                                                           float* __restrict__ grouped_xyz, int* __restrict__ err) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int C = 3 + D;
  const long row = i / C;               // (b*S + s)*K + k
  const int c = (int)(i - row * C);
  const long bs = row / K;              // b*S + s
  const long b = bs / S;
  const long long id = idx[row];
  if (id < 0 || id >= N) { *err = 1; return; }
  float v;
  if (c < 3) {
    const float g = xyz[((size_t)b * N + id) * 3 + c];
    if (grouped_xyz) grouped_xyz[row * 3 + c] = g;
    v = g - new_xyz[bs * 3 + c];
  } else {
    v = points[((size_t)b * N + id) * D + (c - 3)];
  }
  new_points[i] = v;
}

}  // namespace

extern "C" int cg_square_distance(const float* src, const float* dst, int B, int N, int M, float* out, void* stream) {
  if (B < 0 || N < 0 || M < 0) return CG_ERR_ARG;
  if ((long)B * N * M == 0) return CG_OK;
  if (!src || !dst || !out) return CG_ERR_ARG;
  dim3 grid((unsigned)((M + 255) / 256), (unsigned)((N + SQ_ROWS - 1) / SQ_ROWS), (unsigned)B);
  hipLaunchKernelGGL(square_distance_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, N, M, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_square_distance_nd(const float* src, const float* dst, int B, int N, int M, int C, float* out, void* stream) {
  if (B < 0 || N < 0 || M < 0 || C <= 0) return CG_ERR_ARG;
  if ((long)B * N * M == 0) return CG_OK;
  if (!src || !dst || !out) return CG_ERR_ARG;
  if (C == 3) return cg_square_distance(src, dst, B, N, M, out, stream);
  dim3 grid((unsigned)((M + 255) / 256), (unsigned)((N + SQ_ROWS - 1) / SQ_ROWS), (unsigned)B);
  hipLaunchKernelGGL(square_distance_nd_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, N, M, C, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_index_points(const float* points, const long long* idx, int B, int N, int C, long S, float* out, int* err_flag,
                               void* stream) {
  if (B < 0 || N < 0 || C <= 0 || S < 0) return CG_ERR_ARG;
  const long total = (long)B * S * C;
  if (total == 0) return CG_OK;
  if (!points || !idx || !out || !err_flag) return CG_ERR_ARG;
  hipLaunchKernelGGL(index_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points, idx, N, C,
                     S, total, out, err_flag);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_query_ball_point(const float* xyz, const float* new_xyz, int B, int N, int S, float radius_sq, int nsample,
                                   long long* out, void* stream) {
  if (B < 0 || N <= 0 || S < 0 || nsample < 0) return CG_ERR_ARG;
  if ((long)B * S * nsample == 0) return CG_OK;
  if (!xyz || !new_xyz || !out) return CG_ERR_ARG;
  dim3 grid((unsigned)((S + 3) / 4), (unsigned)B);
  hipLaunchKernelGGL(ball_query_kernel, grid, dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, N, S, radius_sq, nsample, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_group_points(const float* xyz, const float* points, const float* new_xyz, const long long* idx, int B, int N, int S,
                               int K, int D, float* new_points, float* grouped_xyz, int* err_flag, void* stream) {
  if (B < 0 || N <= 0 || S < 0 || K < 0 || D < 0) return CG_ERR_ARG;
  const long total = (long)B * S * K * (3 + D);
  if (total == 0) return CG_OK;
  if (!xyz || !new_xyz || !idx || !new_points || !err_flag || (D > 0 && !points)) return CG_ERR_ARG;
  hipLaunchKernelGGL(group_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, points, new_xyz,
                     idx, N, S, K, D, total, new_points, grouped_xyz, err_flag);
  return cg_hip_status(hipGetLastError());
}
