// CollisionManager.isAnyCollision for the pairs the grasp filter never forms (my_cpp/collision_manager.cpp:93-111 loops over EVERY
// pair of registered objects): posed mesh / posed mesh and voxelised cloud / voxelised cloud.
//
//   mesh / mesh  : FCL collides two BVHModel<OBBRSSf> (collision_manager.cpp:41-45) down to triangle pairs; the predicate is "some
//                  closed triangle of A meets some closed triangle of B".  Here: both meshes posed into the common frame with the
//                  fma chain of the mesh/cloud path, one thread per triangle of A, B's posed triangles and boxes staged through LDS
//                  a chunk at a time, box reject, then a 17-axis float32 separating-axis test (2 normals, 9 edge x edge, 6 in-plane
//                  edge normals -- the last six decide coplanar pairs, where the cross products degenerate).
//   cloud / cloud: two fcl::OcTree (collision_manager.cpp:63-70): occupied leaf cube against occupied leaf cube, each set in its own
//                  pose: cube of A (axis-aligned in A's frame) against the cube of B carried into A's frame by inv(pose A) pose B --
//                  the 15-axis box/box separating-axis test FCL's box-box narrow phase is built on, float32.
// Both are brute-force pair scans behind a cheap reject (a few 10^7 pairs for the path's sizes: the API has no caller inside the
// reference's pick cycle -- it exists so that the drop-in CollisionManager answers every pair the reference's would).
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

constexpr int PB_CHUNK = 256;     // triangles / voxels of B staged per workgroup pass

__device__ __forceinline__ void pose_vertex(const float* T, const float* v, float* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
    o[r] = fmaf(T[r * 4 + 0], v[0], fmaf(T[r * 4 + 1], v[1], fmaf(T[r * 4 + 2], v[2], T[r * 4 + 3])));
}

__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// closed triangle (p0,p1,p2) vs closed triangle (q0,q1,q2), coordinates relative to p0 (small numbers: the projections do not cancel)
__device__ __forceinline__ bool tri_tri_overlap(const float* P, const float* Q) {
  float p[3][3], q[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int a = 0; a < 3; ++a) { p[i][a] = P[3 * i + a] - P[a]; q[i][a] = Q[3 * i + a] - P[a]; }
  float ep[3][3], eq[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int a = 0; a < 3; ++a) { ep[i][a] = p[(i + 1) % 3][a] - p[i][a]; eq[i][a] = q[(i + 1) % 3][a] - q[i][a]; }
  float np_[3], nq[3];
  cross3(ep[0], ep[1], np_); cross3(eq[0], eq[1], nq);
  int sep = 0;
  auto axis = [&](const float* L) {
    const float a0 = dot3(L, p[0]), a1 = dot3(L, p[1]), a2 = dot3(L, p[2]);
    const float b0 = dot3(L, q[0]), b1 = dot3(L, q[1]), b2 = dot3(L, q[2]);
    const float amin = fminf(fminf(a0, a1), a2), amax = fmaxf(fmaxf(a0, a1), a2);
    const float bmin = fminf(fminf(b0, b1), b2), bmax = fmaxf(fmaxf(b0, b1), b2);
    sep = (amax < bmin || bmax < amin) ? 1 : sep;
  };
  axis(np_); axis(nq);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { float L[3]; cross3(ep[i], eq[j], L); axis(L); }
#pragma unroll
  for (int i = 0; i < 3; ++i) { float L[3]; cross3(np_, ep[i], L); axis(L); cross3(nq, eq[i], L); axis(L); }
  return !sep;
}

// A: (nfa) triangles, B: (nfb); out[0] |= 1 on the first hit.  grid.x over A's triangles (256 per workgroup), grid.y over chunks of B.
__global__ __launch_bounds__(256) void mesh_mesh_collide_kernel(const float* VA, const int* FA, int nfa, const float* VB, const int* FB, int nfb,
                                                               const float* poseA, const float* poseB, int chunks_per_block,
                                                               unsigned char* out) {
  __shared__ float tb[PB_CHUNK][16];          // posed triangle of B (9) + its box (6)
  float TA[12], TB[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) { TA[i] = poseA[i]; TB[i] = poseB[i]; }
  const int ta = blockIdx.x * 256 + threadIdx.x;
  float P[9], lo[3], hi[3];
  const bool live = ta < nfa;
  if (live) {
#pragma unroll
    for (int k = 0; k < 3; ++k) pose_vertex(TA, VA + 3 * (size_t)FA[3 * (size_t)ta + k], P + 3 * k);
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(fminf(P[a], P[3 + a]), P[6 + a]); hi[a] = fmaxf(fmaxf(P[a], P[3 + a]), P[6 + a]); }
  }
  bool hit = false;
  const int c0 = blockIdx.y * chunks_per_block;
  for (int c = c0; c < c0 + chunks_per_block && c * PB_CHUNK < nfb; ++c) {
    if (__syncthreads_or(threadIdx.x == 0 && *(volatile unsigned char*)out)) break;      // some workgroup already found a pair (the
                                                                                        // barrier also orders the LDS reuse below)
    const int tbi = c * PB_CHUNK + threadIdx.x;
    if (threadIdx.x < PB_CHUNK && tbi < nfb) {
      float Q[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) pose_vertex(TB, VB + 3 * (size_t)FB[3 * (size_t)tbi + k], Q + 3 * k);
#pragma unroll
      for (int i = 0; i < 9; ++i) tb[threadIdx.x][i] = Q[i];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        tb[threadIdx.x][9 + a] = fminf(fminf(Q[a], Q[3 + a]), Q[6 + a]);
        tb[threadIdx.x][12 + a] = fmaxf(fmaxf(Q[a], Q[3 + a]), Q[6 + a]);
      }
    }
    __syncthreads();
    const int n = min(PB_CHUNK, nfb - c * PB_CHUNK);
    if (live && !hit) {
      for (int j = 0; j < n; ++j) {
        const float* t = tb[j];
        if (lo[0] > t[12] || hi[0] < t[9] || lo[1] > t[13] || hi[1] < t[10] || lo[2] > t[14] || hi[2] < t[11]) continue;
        if (tri_tri_overlap(P, t)) { hit = true; break; }
      }
    }
  }
  if (hit) *out = 1;
}

// cube (centre ca, half edge ha, axis-aligned) vs cube of half edge hb at centre cb with axes = the columns of R (both in A's frame)
__device__ __forceinline__ bool box_box_overlap(const float* ca, float ha, const float* cb, float hb, const float* R /* 3x3 row-major */) {
  float t[3], Q[9];
#pragma unroll
  for (int a = 0; a < 3; ++a) t[a] = cb[a] - ca[a];
#pragma unroll
  for (int i = 0; i < 9; ++i) Q[i] = fabsf(R[i]);
  int sep = 0;
  // A's axes
#pragma unroll
  for (int i = 0; i < 3; ++i) sep = fabsf(t[i]) > ha + hb * ((Q[3 * i] + Q[3 * i + 1]) + Q[3 * i + 2]) ? 1 : sep;
  // B's axes
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float tj = (t[0] * R[j] + t[1] * R[3 + j]) + t[2] * R[6 + j];
    sep = fabsf(tj) > ha * ((Q[j] + Q[3 + j]) + Q[6 + j]) + hb ? 1 : sep;
  }
  // edge x edge: L = a_i x b_j
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const float tl = t[i2] * R[3 * i1 + j] - t[i1] * R[3 * i2 + j];
      const float ra = ha * Q[3 * i2 + j] + ha * Q[3 * i1 + j];
      const float rb = hb * Q[3 * i + j2] + hb * Q[3 * i + j1];
      sep = fabsf(tl) > ra + rb ? 1 : sep;
    }
  }
  return !sep;
}

// keys (n,4) int16 (key - 32768).  rel: 3x4 row-major, B's frame -> A's frame.  One thread per voxel of A; B staged through LDS.
__global__ __launch_bounds__(256) void cloud_cloud_collide_kernel(const short* keysA, int na, float resA, const short* keysB, int nb, float resB,
                                                                 const float* rel, int chunks_per_block, unsigned char* out) {
  __shared__ float cbs[PB_CHUNK][4];
  float T[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = rel[i];
  const float R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
  const float ha = 0.5f * resA, hb = 0.5f * resB;
  const float reach = 1.7320508f * (ha + hb) * 1.0001f;            // centres farther apart than both half diagonals: no contact
  const int ia = blockIdx.x * 256 + threadIdx.x;
  const bool live = ia < na;
  float ca[3] = {0.f, 0.f, 0.f};
  if (live) {
#pragma unroll
    for (int a = 0; a < 3; ++a) ca[a] = ((float)keysA[4 * (size_t)ia + a] + 0.5f) * resA;
  }
  bool hit = false;
  const int c0 = blockIdx.y * chunks_per_block;
  for (int c = c0; c < c0 + chunks_per_block && c * PB_CHUNK < nb; ++c) {
    if (__syncthreads_or(threadIdx.x == 0 && *(volatile unsigned char*)out)) break;
    const int ib = c * PB_CHUNK + threadIdx.x;
    if (threadIdx.x < PB_CHUNK && ib < nb) {
      float cb[3], o[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) cb[a] = ((float)keysB[4 * (size_t)ib + a] + 0.5f) * resB;
      pose_vertex(T, cb, o);
      cbs[threadIdx.x][0] = o[0]; cbs[threadIdx.x][1] = o[1]; cbs[threadIdx.x][2] = o[2];
    }
    __syncthreads();
    const int n = min(PB_CHUNK, nb - c * PB_CHUNK);
    if (live && !hit) {
      for (int j = 0; j < n; ++j) {
        const float* cb = cbs[j];
        if (fabsf(cb[0] - ca[0]) > reach || fabsf(cb[1] - ca[1]) > reach || fabsf(cb[2] - ca[2]) > reach) continue;
        if (box_box_overlap(ca, ha, cb, hb, R)) { hit = true; break; }
      }
    }
  }
  if (hit) *out = 1;
}

}  // namespace

extern "C" int cg_mesh_mesh_collide(const float* vertices_a, const int* faces_a, int n_faces_a, const float* vertices_b, const int* faces_b,
                                    int n_faces_b, const float* pose_a, const float* pose_b, unsigned char* out, void* stream) {
  if (n_faces_a < 0 || n_faces_b < 0) return CG_ERR_ARG;
  if (!pose_a || !pose_b || !out) return CG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, 1, s);
  if (e != hipSuccess) return (int)e;
  if (n_faces_a == 0 || n_faces_b == 0) return CG_OK;
  if (!vertices_a || !faces_a || !vertices_b || !faces_b) return CG_ERR_ARG;
  const int chunks = (n_faces_b + PB_CHUNK - 1) / PB_CHUNK;
  const int gx = (n_faces_a + 255) / 256;
  int gy = chunks;
  if ((long)gx * gy > 4096) gy = (int)((4096 + gx - 1) / gx);
  if (gy < 1) gy = 1;
  const int per = (chunks + gy - 1) / gy;
  gy = (chunks + per - 1) / per;
  hipLaunchKernelGGL(mesh_mesh_collide_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, vertices_a, faces_a, n_faces_a, vertices_b,
                     faces_b, n_faces_b, pose_a, pose_b, per, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_voxels_voxels_collide(const short* keys_a, int n_keys_a, float resolution_a, const short* keys_b, int n_keys_b,
                                        float resolution_b, const float* b_in_a, unsigned char* out, void* stream) {
  if (n_keys_a < 0 || n_keys_b < 0 || !(resolution_a > 0.f) || !(resolution_b > 0.f)) return CG_ERR_ARG;
  if (!b_in_a || !out) return CG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, 1, s);
  if (e != hipSuccess) return (int)e;
  if (n_keys_a == 0 || n_keys_b == 0) return CG_OK;
  if (!keys_a || !keys_b) return CG_ERR_ARG;
  const int chunks = (n_keys_b + PB_CHUNK - 1) / PB_CHUNK;
  const int gx = (n_keys_a + 255) / 256;
  int gy = chunks;
  if ((long)gx * gy > 4096) gy = (int)((4096 + gx - 1) / gx);
  if (gy < 1) gy = 1;
  const int per = (chunks + gy - 1) / gy;
  gy = (chunks + per - 1) / per;
  hipLaunchKernelGGL(cloud_cloud_collide_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, keys_a, n_keys_a, resolution_a, keys_b,
                     n_keys_b, resolution_b, b_in_a, per, out);
  return cg_hip_status(hipGetLastError());
}
